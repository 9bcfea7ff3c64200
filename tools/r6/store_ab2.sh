#!/bin/bash
# Round 6, experiment 3: the store PATTERN of the pair GEMM's epilogue (XL_PAIR_DBG=16: 8 rows x 128 B per instruction, 32: 16 rows x 64 B;
# 23 = 7 + 16: MFMAs + full-line stores only); results are garbage, timing only
set -u
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
O="$GRAFT_REPO_ROOT/gpurun_out/r6_store"; mkdir -p "$O"; : > "$O/ab2.txt"
for rep in 1 2; do
for d in 0 16 32 8 23 7; do
    echo "== dbg $d rep $rep" >> "$O/ab2.txt"
    XL_PAIR_ONLY_DMA=1 XL_PAIR_DBG=$d timeout 300 python tools/pair_gemm_bench.py 95 2>&1 | grep -E "^pair_dma  [0-9]" >> "$O/ab2.txt"
done
done
cat "$O/ab2.txt"
