#!/bin/bash
# Round 6, experiment 4: the epilogue with the lane exchange (64-byte pieces; default) against the old pattern (XL_PAIR_VAR=4)
set -u
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
O="$GRAFT_REPO_ROOT/gpurun_out/r6_store"; mkdir -p "$O"; : > "$O/ab3.txt"
for rep in 1 2 3; do
for v in 0 4; do
    echo "== var $v rep $rep" >> "$O/ab3.txt"
    XL_PAIR_ONLY_DMA=1 XL_PAIR_VAR=$v timeout 300 python tools/pair_gemm_bench.py 95 2>&1 | grep -E "^pair_dma" >> "$O/ab3.txt"
done
done
XL_PAIR_CLK=1 timeout 300 python tools/pair_gemm_bench.py 95 2>&1 | grep "pair clk" | tail -3 >> "$O/ab3.txt"
cat "$O/ab3.txt"
timeout 900 python -m pytest tests/test_pair_gpu.py -m gpu -x -q 2>&1 | tail -5
