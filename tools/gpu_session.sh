#!/bin/bash
# scratch script of the current gpurun call (overwritten per call)
O=$GRAFT_REPO_ROOT/gpurun_out/sess; mkdir -p $O
python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log
grep -E "^FAILED|^ERROR|passed|failed|rc=" $O/gputest.log | tail -8; grep -E "^E " $O/gputest.log | head -12
B="python bench.py --no-secondary --no-cpu-baseline"
for i in 1 2; do
  $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('1 stream', d['value'], d['ms_per_step'])"
  $B --cnn-streams 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('2 streams', d['value'], d['ms_per_step'])"
done
