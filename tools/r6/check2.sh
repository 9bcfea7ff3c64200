#!/bin/bash
set -u
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests/test_cnn_gpu.py tests/test_train_gpu.py tests/test_cnn_bwd_gpu.py -m gpu -x -q 2>&1 | tail -3
python tools/dropin_breakdown.py 2>&1 | tail -8
