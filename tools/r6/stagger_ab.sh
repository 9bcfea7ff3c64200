#!/bin/bash
# Round 6, experiment 1: the pair GEMM's workgroups started out of phase (XL_PAIR_STAGGER = spread in 10 ns ticks) and
# streaming stores (XL_PAIR_VAR=2); kernel alone, 95 frames, random data.
set -u
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
O="$GRAFT_REPO_ROOT/gpurun_out/r6_stagger"; mkdir -p "$O"
python tools/hbm_bw_probe.py > "$O/hbm_probe.txt" 2>&1
for rep in 1 2; do
for s in 0 1500 3000 4000 5000 6000 8000; do
  for v in 0 2; do
    echo "== stagger $s var $v rep $rep" >> "$O/ab.txt"
    XL_PAIR_ONLY_DMA=1 XL_PAIR_STAGGER=$s XL_PAIR_VAR=$v timeout 300 python tools/pair_gemm_bench.py 95 2>&1 | grep -E "^pair_dma" >> "$O/ab.txt"
  done
done
done
cat "$O/ab.txt"
