#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r4n; mkdir -p $O
timeout 900 python -m pytest tests/test_cnn_bwd_gpu.py tests/test_train_gpu.py tests/test_reference_fixtures.py tests/test_full_size_gpu.py tests/test_semantics_gpu.py -m gpu -q -x > $O/t.log 2>&1; echo "rc=$?"; grep -E "^FAILED|^ERROR|passed|failed" $O/t.log | tail; grep -E "^E " $O/t.log | head -20
python tools/train_step_bench.py --full --steps 8 2>&1 | tail -1
XL_NO_TRAIN_DEFER=1 python tools/train_step_bench.py --full --steps 8 2>&1 | tail -1
python tools/train_step_bench.py --full --steps 8 2>&1 | tail -1
XL_NO_TRAIN_DEFER=1 python tools/train_step_bench.py --full --steps 8 2>&1 | tail -1
