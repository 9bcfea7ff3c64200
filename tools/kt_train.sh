#!/bin/bash
# kernel statistics of the full batch-16 training step (tools/train_step_bench.py --full) -> gpurun_out/train_kstats.csv
set -eu
: "${GRAFT_REPO_ROOT:?run under gpurun (or export GRAFT_REPO_ROOT)}"
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktt -- python "$GRAFT_REPO_ROOT/tools/train_step_bench.py" --full --steps 5 > /tmp/ktt.log 2>&1
tail -1 /tmp/ktt.log
cp $(ls /tmp/ktt/*/*kernel_stats.csv | head -1) "$GRAFT_REPO_ROOT/gpurun_out/train_kstats.csv"
