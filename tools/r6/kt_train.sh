#!/bin/bash
set -u
: "${GRAFT_REPO_ROOT:?}"
O="$GRAFT_REPO_ROOT/gpurun_out/r6_train"; rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/ktt" -- python "$GRAFT_REPO_ROOT/tools/train_step_bench.py" --full --steps 5 > "$O/ktt.log" 2>&1
cd "$GRAFT_REPO_ROOT"
cp "$(ls $O/ktt/*/*kernel_stats.csv | head -1)" "$O/train_step_kernel_stats.csv"; rm -rf "$O/ktt"
tail -2 "$O/ktt.log"
python - "$O/train_step_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = 7.0
tot = 0.0
for r in rows:
    ms = float(r["TotalDurationNs"]) / 1e6 / steps
    tot += ms
    if ms > 0.08:
        print("%-80s %6.1f calls/step %8.1f us  %7.3f ms/step" % (r["Name"].replace("(anonymous namespace)::", "")[:80], float(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3, ms))
print("sum of kernel time per step: %.2f ms" % tot)
PY
