#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/sess; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 -i $GRAFT_REPO_ROOT/tools/pmc_r2.txt --kernel-trace --output-format csv -d $O/pmc -- python $GRAFT_REPO_ROOT/tools/train_step_bench.py --full --steps 2 > $O/pmc.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $O/pmc $O/sum.csv > /dev/null
python tools/pmc_derive.py $O/sum.csv $O/der.csv > /dev/null
rm -rf $O/pmc
