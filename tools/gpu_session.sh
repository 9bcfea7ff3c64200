#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/sess; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_cnn_bwd_gpu.py tests/test_train_gpu.py -m gpu -q 2>&1 | grep -E "passed|failed|rc="; echo "tests done"
cd /tmp; export TMPDIR=/tmp
for v in new old; do
  [ $v = old ] && export XL_WGRAD_SMALL_SPLITS_OLD=1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$v -o r -- python $GRAFT_REPO_ROOT/tools/train_step_bench.py --full --steps 5 2>&1 | grep "HIP path"
  python - $v <<'PY'
import csv,glob,os,sys
f=glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/sess/kt_%s/**/*kernel_stats.csv'%sys.argv[1],recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'wgrad_kernel' in r['Name'] or 'wgrad_reduce' in r['Name']: print(sys.argv[1], '%-80s %5s %10.1f'%(r['Name'][:80], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
