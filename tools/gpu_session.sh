#!/bin/bash
B="python bench.py --no-secondary --no-cpu-baseline"
for b in 24 32 47 64 24 47; do
  $B --mlr 3 --batch $b --steps 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mlr3 batch $b', d['value'], d['ms_per_step'], d['config']['median_err_cm'])"
done
for b in 95 96 111 127; do
  $B --batch $b --steps 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch $b', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done
