#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/sess; mkdir -p $O
timeout 600 python -m pytest tests/test_cnn_gpu.py -m gpu -q -k "residual_epilogue or fused_stem" 2>&1 | tail -15
python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log
grep -E "^FAILED|^ERROR|passed|failed|rc=" $O/gputest.log | tail -8; grep -E "^E " $O/gputest.log | head -12
python -m pytest tests -q -m "not gpu" -x 2>&1 | tail -2
