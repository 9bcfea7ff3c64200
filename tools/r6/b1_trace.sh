#!/bin/bash
set -u
: "${GRAFT_REPO_ROOT:?}"
O="$GRAFT_REPO_ROOT/gpurun_out/r6_b1"; rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$O/kt" -- python "$GRAFT_REPO_ROOT/tools/b1_trace.py" run 1 > "$O/kt.log" 2>&1
cd "$GRAFT_REPO_ROOT"
python tools/b1_trace.py show "$(ls $O/kt/*/*kernel_trace.csv | head -1)" > "$O/timeline.txt"
rm -rf "$O/kt"
cat "$O/timeline.txt"
