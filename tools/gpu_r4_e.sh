#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r4e; mkdir -p $O
for t in 4 8; do
  XL_STEM12_TILE=$t timeout 600 python -m pytest tests/test_cnn_gpu.py -m gpu -q -k "fused_stem" 2>&1 | tail -2
  XL_STEM12_TILE=$t python tools/stem12_bench.py
  XL_STEM12_TILE=$t XL_STEM12_CLK=1 python tools/stem12_bench.py 2>&1 | tail -4
done
B="python bench.py --no-secondary --no-cpu-baseline"
$B > $O/b4.json 2>$O/b4.err; XL_STEM12_TILE=8 $B > $O/b8.json 2>/dev/null; XL_NO_STEM12=1 $B > $O/bold.json 2>/dev/null
$B > $O/b4b.json 2>/dev/null; XL_NO_STEM12=1 $B > $O/boldb.json 2>/dev/null
python - <<PY
import json
for n in ("b4", "b8", "bold", "b4b", "boldb"):
    try:
        d = json.load(open("$O/%s.json" % n)); print(n, d["value"], d["ms_per_step"], d["config"]["median_err_cm"])
    except Exception as e:
        print(n, "failed", e)
PY
tail -3 $O/b4.err
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- $B --steps 6 --warmup 2 > $O/kt.log 2>&1
cd $GRAFT_REPO_ROOT
cp $(ls $O/kt/*/*kernel_stats.csv | head -1) $O/kstats.csv; rm -rf $O/kt
python tools/kstats_show.py $O/kstats.csv stem12 conv1_mfma s2_kernel gn_stats dsac
