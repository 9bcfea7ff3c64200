#!/bin/bash
# round 3, call b: GPU suite (all failures) + A/B of the Winograd V form (fp32 V split inside the GEMM vs bf16 planes written by the transform)
O=gpurun_out/r3b; mkdir -p $O
python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log
XL_BENCH_VERBOSE=1 python bench.py --no-secondary --no-cpu-baseline > $O/bench_actV.json 2> $O/bench_actV.err
XL_WINO_V_SPLIT=1 XL_BENCH_VERBOSE=1 python bench.py --no-secondary --no-cpu-baseline > $O/bench_splitV.json 2> $O/bench_splitV.err
python bench.py --no-secondary --no-cpu-baseline > $O/bench_actV_quiet.json 2> /dev/null
XL_WINO_V_SPLIT=1 python bench.py --no-secondary --no-cpu-baseline > $O/bench_splitV_quiet.json 2> /dev/null
tail -4 $O/gputest.log
for f in $O/bench_*.json; do echo $f; cut -c1-200 $f; done
