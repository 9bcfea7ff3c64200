#!/bin/bash
timeout 1200 python -m pytest tests/test_cnn_bwd_gpu.py tests/test_train_gpu.py tests/test_reference_fixtures.py -m gpu -q -x 2>&1 | grep -E "^E   |passed|failed|Error" | cut -c1-300 | head
for i in 1 2; do python tools/train_step_bench.py --full --steps 10 2>&1 | tail -1; XL_NO_STEM_STATS=1 python tools/train_step_bench.py --full --steps 10 2>&1 | tail -1; done
