#!/bin/bash
# Round 6, experiment 5: two phase-shifted 128 x 256 workgroups per CU (XL_PAIR_PP=1) against the 256 x 256 workgroup
set -u
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
O="$GRAFT_REPO_ROOT/gpurun_out/r6_pp"; mkdir -p "$O"; : > "$O/ab.txt"
for rep in 1 2 3; do
for v in 0 1; do
    echo "== pp $v rep $rep" >> "$O/ab.txt"
    XL_PAIR_ONLY_DMA=1 XL_PAIR_PP=$v timeout 300 python tools/pair_gemm_bench.py 95 2>&1 | grep -E "^pair_dma" >> "$O/ab.txt"
done
done
cat "$O/ab.txt"
XL_PAIR_PP=1 timeout 900 python -m pytest tests/test_pair_gpu.py tests/test_reference_fixtures.py -m gpu -x -q -k "pair or 480x720" 2>&1 | tail -4
for v in 0 1; do
XL_PAIR_PP=$v timeout 600 python bench.py --no-secondary --no-cpu-baseline > "$O/bench_pp$v.json" 2> /dev/null
python -c "
import json; d=json.load(open('$O/bench_pp$v.json')); r=d['roofline']; print('pp $v:', d['value'], 'img/s; ms/step', d['ms_per_step'], 'dominant', r.get('avg_launch_ms'), 'frac', r['frac'])"
done
