#!/bin/bash
set -u
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
O="$GRAFT_REPO_ROOT/gpurun_out/r6_stem"; mkdir -p "$O"; : > "$O/ab.txt"
for rep in 1 2; do
for w in 2 3; do
  echo "== WGS $w rep $rep" >> "$O/ab.txt"
  XL_STEM12_WGS=$w timeout 300 python tools/stem12_bench.py 95 2>&1 | grep -E "^stem12" >> "$O/ab.txt"
done
done
XL_STEM12_CLK=1 timeout 300 python tools/stem12_bench.py 95 2>&1 | grep "stem12 clk" | tail -4 >> "$O/ab.txt"
cat "$O/ab.txt"
timeout 900 python -m pytest tests/test_cnn_gpu.py -m gpu -x -q -k "stem or conv1 or golden or full_size" 2>&1 | tail -4
bash tools/r6/kstats.sh "" base
bash tools/r6/kstats.sh "XL_STEM12_WGS=3" wgs3
