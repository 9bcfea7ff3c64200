#!/bin/bash
# Round 6, experiment 2: what the epilogue stores of the pair GEMM cost (XL_PAIR_DBG=8: no stores; 7: MFMAs + stores only; 15: MFMAs only)
set -u
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
O="$GRAFT_REPO_ROOT/gpurun_out/r6_store"; mkdir -p "$O"; : > "$O/ab.txt"
for rep in 1 2; do
for d in 0 8 7 15 1 2 4; do
    echo "== dbg $d rep $rep" >> "$O/ab.txt"
    XL_PAIR_ONLY_DMA=1 XL_PAIR_DBG=$d timeout 300 python tools/pair_gemm_bench.py 95 2>&1 | grep -E "^pair_dma  [0-9]" >> "$O/ab.txt"
done
done
echo "== clk" >> "$O/ab.txt"
XL_PAIR_CLK=1 timeout 300 python tools/pair_gemm_bench.py 95 2>&1 | grep "pair clk" | tail -3 >> "$O/ab.txt"
cat "$O/ab.txt"
