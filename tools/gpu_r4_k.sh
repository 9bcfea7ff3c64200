#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r4k; mkdir -p $O
python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log
grep -E "^FAILED|^ERROR|passed|failed|rc=" $O/gputest.log | tail -8
for b in 47 1; do
  python tools/dsac_bench.py $b; XL_DSAC_PAIR_CELLS=0 python tools/dsac_bench.py $b
  python tools/dsac_bench.py $b; XL_DSAC_PAIR_CELLS=0 python tools/dsac_bench.py $b
done
B="python bench.py --no-secondary --no-cpu-baseline"
$B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['config']['dsac_ms_per_batch'])"
XL_DSAC_PAIR_CELLS=0 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench one cell per trip', d['value'], d['ms_per_step'], d['config']['dsac_ms_per_batch'])"
