#!/bin/bash
# A/B of the tile forms on the latency legs (batch 1 and 8)
O=gpurun_out/$1; mkdir -p $O
python -m pytest tests/test_cnn_gpu.py -m gpu -q -k "tile_forms or small_tile" 2>&1 | tail -3
for f1 in 256 192 384; do for fw in 256 192 384; do
  XL_TILE_FORM_1X1=$f1 XL_TILE_FORM_WINO=$fw python tools/latency_ab.py 1 8 2>/dev/null | tail -1
done; done | tee $O/latency_ab.txt
python tools/latency_ab.py 1 8 2>/dev/null | tail -1 | tee -a $O/latency_ab.txt
XL_NO_SMALL_TILES=1 python tools/latency_ab.py 1 8 2>/dev/null | tail -1 | tee -a $O/latency_ab.txt
