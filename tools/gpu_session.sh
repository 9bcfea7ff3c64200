#!/bin/bash
timeout 900 python -m pytest tests/test_cnn_gpu.py -m gpu -q -k "residual_epilogue" 2>&1 | grep -E "^E   |passed|failed|Error" | cut -c1-300 | head -20
