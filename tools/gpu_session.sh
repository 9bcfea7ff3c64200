#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/sess; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_cnn_bwd_gpu.py tests/test_train_gpu.py tests/test_reference_fixtures.py tests/test_cnn_gpu.py -m gpu -q -x 2>&1 | grep -E "^E   |passed|failed|Error" | cut -c1-300 | head
for i in 1 2; do python tools/train_step_bench.py --full --steps 10 2>&1 | tail -1; done
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o r -- python $GRAFT_REPO_ROOT/tools/train_step_bench.py --full --steps 5 > /dev/null 2>&1
python - <<'PY'
import csv,glob,os
f=glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/sess/kt/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    if 'gn' in r['Name']: print('%-90s %5s %10.1f'%(r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3))
PY
