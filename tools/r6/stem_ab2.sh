#!/bin/bash
set -u
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
O="$GRAFT_REPO_ROOT/gpurun_out/r6_stem"; mkdir -p "$O"; : > "$O/ab2.txt"
for rep in 1 2 3; do
  XL_STEM12_WGS=2 timeout 300 python tools/stem12_bench.py 95 2>&1 | grep -E "^stem12" >> "$O/ab2.txt"
done
XL_STEM12_CLK=1 timeout 300 python tools/stem12_bench.py 95 2>&1 | grep "stem12 clk" | tail -4 >> "$O/ab2.txt"
cat "$O/ab2.txt"
timeout 900 python -m pytest tests/test_cnn_gpu.py -m gpu -x -q -k "stem or conv1 or golden or full_size" 2>&1 | tail -4
timeout 600 python bench.py --no-secondary --no-cpu-baseline > "$O/bench.json" 2> /dev/null
python -c "
import json; d=json.load(open('$O/bench.json')); r=d['roofline']; print(d['value'], 'img/s; ms/step', d['ms_per_step'], 'dominant', r.get('avg_launch_ms'), 'frac', r['frac'], d['config'].get('dsac_valu_roofline'))"
