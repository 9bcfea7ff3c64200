#!/bin/bash
# usage: gpu_ab.sh <outdir> [--tests] [--verbose] variant1 variant2 ...   (a variant = "NAME:ENV1=V1,ENV2=V2" or "NAME:")
# runs the GPU suite (optional) and bench.py once per variant (quiet run; plus one XL_BENCH_VERBOSE run with --verbose)
O=gpurun_out/$1; shift; mkdir -p $O
TESTS=0; VERB=0
while [[ "$1" == --* ]]; do [[ "$1" == --tests ]] && TESTS=1; [[ "$1" == --verbose ]] && VERB=1; shift; done
if [[ $TESTS == 1 ]]; then python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log; grep -E "^FAILED|passed|failed" $O/gputest.log | tail -8; fi
for v in "$@"; do
  name=${v%%:*}; envs=${v#*:}
  envcmd="env"; IFS=',' read -ra E <<< "$envs"; for e in "${E[@]}"; do [[ -n "$e" ]] && envcmd="$envcmd $e"; done
  $envcmd python bench.py --no-secondary --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err
  if [[ $VERB == 1 ]]; then $envcmd XL_BENCH_VERBOSE=1 python bench.py --no-secondary --no-cpu-baseline > $O/benchv_$name.json 2> $O/benchv_$name.err; grep "by type" $O/benchv_$name.err; fi
  echo "$name: $(python -c "import json,sys; d=json.load(open('$O/bench_$name.json')); print(d['value'], d['ms_per_step'], d['config']['cnn_ms_per_batch'], d['roofline']['avg_launch_ms'], d.get('roofline_forward',{}).get('frac'))" 2>&1 | tail -1)"
done
