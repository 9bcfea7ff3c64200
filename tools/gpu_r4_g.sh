#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r4g; mkdir -p $O
timeout 600 python -m pytest tests/test_cnn_gpu.py -m gpu -q -k "fused_stem" 2>&1 | tail -2
python tools/stem12_bench.py
export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-secondary --no-cpu-baseline"
for v in t4 old; do
  cd /tmp
  unset XL_STEM12_TILE XL_NO_STEM12
  [ $v = old ] && export XL_NO_STEM12=1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$v -- $B --steps 6 --warmup 2 > $O/kt_$v.log 2>&1
  cd $GRAFT_REPO_ROOT
  cp $(ls $O/kt_$v/*/*kernel_stats.csv | head -1) $O/kstats_$v.csv
  cp $(ls $O/kt_$v/*/*kernel_trace.csv | head -1) $O/ktrace_$v.csv
  rm -rf $O/kt_$v
  echo "== $v"; python tools/kstats_show.py $O/kstats_$v.csv stem12 conv1_mfma s2_kernel gn_stats dsac
done
unset XL_STEM12_TILE XL_NO_STEM12
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  $B > $O/new$i.json 2>/dev/null; XL_NO_STEM12=1 $B > $O/old$i.json 2>/dev/null
done
python - <<PY
import json
for n in ("new1", "old1", "new2", "old2"):
    try:
        d = json.load(open("$O/%s.json" % n)); print(n, d["value"], d["ms_per_step"], d["config"]["cnn_ms_per_batch"], d["config"]["median_err_cm"])
    except Exception as e:
        print(n, "failed", e)
PY
