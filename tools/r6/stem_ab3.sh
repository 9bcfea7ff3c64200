#!/bin/bash
set -u
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
O="$GRAFT_REPO_ROOT/gpurun_out/r6_stem"; mkdir -p "$O"; : > "$O/ab3.txt"
for rep in 1 2 3; do
for w in 2 3; do
  echo "== WGS $w rep $rep" >> "$O/ab3.txt"
  XL_STEM12_WGS=$w timeout 300 python tools/stem12_bench.py 95 2>&1 | grep -E "^stem12" >> "$O/ab3.txt"
done
done
cat "$O/ab3.txt"
for w in 2 3; do
XL_STEM12_WGS=$w timeout 600 python bench.py --no-secondary --no-cpu-baseline > "$O/bench_w$w.json" 2> /dev/null
python -c "
import json; d=json.load(open('$O/bench_w$w.json')); r=d['roofline']; print('wgs $w:', d['value'], 'img/s; ms/step', d['ms_per_step'], 'dominant', r.get('avg_launch_ms'))"
done
