#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r4l; mkdir -p $O
python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log
grep -E "^FAILED|^ERROR|passed|failed|rc=" $O/gputest.log | tail -8; grep -E "^E " $O/gputest.log | head -12
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python - <<PY
import torch, dsacstar, sys
print("native:", dsacstar.NATIVE, getattr(dsacstar, "NATIVE_ERROR", None))
print([l.split()[-1] for l in open("/proc/self/maps") if "crossloc" in l and ".so" in l][:6])
PY
