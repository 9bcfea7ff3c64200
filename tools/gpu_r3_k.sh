#!/bin/bash
O=gpurun_out/r3k; mkdir -p $O
python -m pytest tests/test_cnn_gpu.py tests/test_full_size_gpu.py tests/test_chain_gpu.py -m gpu -q -x > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log; grep -E "^FAILED|passed|failed" $O/gputest.log | tail -5
for v in small: big:XL_NO_SMALL_TILES=1; do
  name=${v%%:*}; envs=${v#*:}
  env $envs XL_BENCH_VERBOSE=1 python bench.py --no-secondary --no-cpu-baseline --batch 1 --steps 30 --warmup 5 > $O/benchv_b1_$name.json 2> $O/benchv_b1_$name.err
  env $envs python bench.py --no-secondary --no-cpu-baseline --batch 1 --steps 50 --warmup 5 > $O/bench_b1_$name.json 2> /dev/null
  python -c "import json; d=json.load(open('$O/bench_b1_$name.json')); print('$name B=1', d['value'], d['ms_per_step'], d['config']['cnn_ms_per_batch'])"
  grep "by type" $O/benchv_b1_$name.err
  grep "k1 s1  512-> 512  60x 90" $O/benchv_b1_$name.err | head -3
done
