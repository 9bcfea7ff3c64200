#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r4j; mkdir -p $O
timeout 900 python -m pytest tests/test_cnn_gpu.py tests/test_cnn_bwd_gpu.py tests/test_train_gpu.py tests/test_reference_fixtures.py tests/test_full_size_gpu.py -m gpu -q -x > $O/t.log 2>&1; echo "rc=$?"; grep -E "^FAILED|^ERROR|passed|failed" $O/t.log | tail; grep -E "^E " $O/t.log | head -20
python tools/train_step_bench.py --full --steps 8 2>&1 | tail -1
XL_NO_S2_DGRAD=1 python tools/train_step_bench.py --full --steps 8 2>&1 | tail -1
python tools/stem12_bench.py
B="python bench.py --no-secondary --no-cpu-baseline"
$B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'])"
XL_NO_STEM12=1 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench two-kernel stem', d['value'], d['ms_per_step'])"
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktt -- python $GRAFT_REPO_ROOT/tools/train_step_bench.py --full --steps 5 > $O/ktt.log 2>&1
cd $GRAFT_REPO_ROOT
cp $(ls $O/ktt/*/*kernel_stats.csv | head -1) $O/train_kstats.csv; rm -rf $O/ktt
python tools/kstats_show.py $O/train_kstats.csv s2_dgrad igemm_conv wgrad_kernel conv1
