#!/bin/bash
python -m pytest tests/test_semantics_gpu.py tests/test_cnn_bwd_gpu.py tests/test_train_gpu.py -m gpu -q 2>&1 | grep -E "passed|failed"
XL_TRAIN_STEM_STATS=1 python -m pytest tests/test_semantics_gpu.py tests/test_cnn_bwd_gpu.py tests/test_train_gpu.py tests/test_reference_fixtures.py -m gpu -q 2>&1 | grep -E "passed|failed|^FAILED"
