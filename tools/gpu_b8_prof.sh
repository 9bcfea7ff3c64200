#!/bin/bash
# kernel statistics of the 8-frame leg (eager op list: XL_CNN_GRAPH=0, so that every kernel is traced by name)
O=gpurun_out/$1; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
XL_CNN_GRAPH=0 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o b8 --output-format csv -- python $GRAFT_REPO_ROOT/tools/latency_ab.py ${2:-8} > $GRAFT_REPO_ROOT/$O/run.log 2>&1
cd $GRAFT_REPO_ROOT
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/b8_kernel_stats.csv
find $O/prof -name "*kernel_trace.csv" -delete
head -40 $O/b8_kernel_stats.csv | cut -c1-200
tail -2 $O/run.log
