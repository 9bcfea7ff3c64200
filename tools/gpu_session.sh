#!/bin/bash
# One gpurun call: the GPU test suite, the smoke entry point and the bench line (scratch script of the round; edit per session).
O=$GRAFT_REPO_ROOT/gpurun_out/sess; mkdir -p $O
python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log
grep -E "^FAILED|^ERROR|passed|failed|rc=" $O/gputest.log | tail -8; grep -E "^E " $O/gputest.log | head -12
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'])"
