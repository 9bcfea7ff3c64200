#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/sess; mkdir -p $O
python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log
grep -E "^FAILED|^ERROR|passed|failed|rc=" $O/gputest.log | tail -8; grep -E "^E " $O/gputest.log | head -12
B="python bench.py --no-secondary --no-cpu-baseline"
for i in 1 2 3; do
  $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('epilogue stats', d['value'], d['ms_per_step'])"
  XL_NO_STEM_STATS=1 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('stats passes  ', d['value'], d['ms_per_step'])"
done
python tools/b1_trace.py run 1 > /dev/null 2>&1
python bench.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print(d['value'], c['latency_b1_ms'], c.get('b8_images_per_s'), c.get('mlr3_images_per_s'), c.get('train16_full_step_ms'))"
