#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r4i; mkdir -p $O
python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log
grep -E "^FAILED|^ERROR|passed|failed|rc=" $O/gputest.log | tail -12
python tools/train_step_bench.py --full --steps 8 2>&1 | tail -3
XL_CONV1_VALU=1 python tools/train_step_bench.py --full --steps 8 2>&1 | tail -2
python bench.py --no-secondary --no-cpu-baseline --mlr 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mlr3', d['value'], d['ms_per_step'])"
