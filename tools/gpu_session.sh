#!/bin/bash
B="python bench.py --no-secondary --no-cpu-baseline"
for i in 1 2; do
for a in "--batch 95" "--batch 95 --cnn-streams 2" "--batch 190" "--batch 190 --cnn-streams 2"; do
  $B $a 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$a', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done; done
