#!/bin/bash
O=gpurun_out/r3h; mkdir -p $O
python -m pytest tests -m gpu -q -x > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log; grep -E "^FAILED|passed|failed" $O/gputest.log | tail -5
for b in 1 8; do
  XL_BENCH_VERBOSE=1 python bench.py --no-secondary --no-cpu-baseline --batch $b --steps 30 --warmup 5 > $O/benchv_b$b.json 2> $O/benchv_b$b.err
  python bench.py --no-secondary --no-cpu-baseline --batch $b --steps 50 --warmup 5 > $O/bench_b$b.json 2> $O/bench_b$b.err
  python -c "import json; d=json.load(open('$O/bench_b$b.json')); print('B=$b', d['value'], d['ms_per_step'], d['config']['cnn_ms_per_batch'], d['config']['dsac_ms_per_batch'])"
  grep "by type" $O/benchv_b$b.err
done
