#!/bin/bash
# the measurement / fallback switches DESIGN.md lists must keep working: run the relevant GPU tests under each of them
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $O
run() { name=$1; shift; env "$@" python -m pytest tests/test_cnn_gpu.py tests/test_full_size_gpu.py tests/test_chain_gpu.py -m gpu -q > $O/sw_$name.log 2>&1; echo "$name: $(grep -E 'passed|failed' $O/sw_$name.log | tail -1) $(grep -E "^FAILED" $O/sw_$name.log | cut -c1-110 | tr "\n" " ")"; }
run fp32 XL_GEMM_SPLIT_BF16=0
run vsplit XL_WINO_V_SPLIT=1
run stem_c16 XL_STEM_FORM=c16
run stem_8x2 XL_STEM_FORM=8x2
run nofold XL_NO_FOLD_GN=1
run nodefer XL_NO_DEFERRED_GN=1
run nosplit1x1 XL_NO_SPLIT_1X1=1
run tmajor XL_WINO_M_TILE_MAJOR=1
run outdma XL_WINO_OUT_DMA=1
run persist4 XL_WINO_V_SPLIT=1 XL_GEMM_PERSIST_WAVES=4
run nograph XL_CNN_GRAPH=0
