#!/bin/bash
B="python bench.py --no-secondary --no-cpu-baseline --mlr 3"
for a in "--batch 47" "--batch 64" "--batch 95" "--batch 47" "--batch 95"; do
  $B $a 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$a', d['value'], d['ms_per_step'])"
done
