#!/bin/bash
# Per-kernel durations of a bench step in the pipeline (rocprofv3 --kernel-trace --stats), optional env in $1 ("A=1,B=2"), tag in $2
set -u
: "${GRAFT_REPO_ROOT:?}"
ENVS="${1:-}"; TAG="${2:-kstats}"
O="$GRAFT_REPO_ROOT/gpurun_out/r6_$TAG"; rm -rf "$O"; mkdir -p "$O"
envcmd="env"; IFS=',' read -ra E <<< "$ENVS"; for e in "${E[@]}"; do [[ -n "$e" ]] && envcmd="$envcmd $e"; done
cd /tmp && export TMPDIR=/tmp
$envcmd rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt" -- python "$GRAFT_REPO_ROOT/bench.py" --no-secondary --no-cpu-baseline --steps 8 > "$O/kt.log" 2>&1
cd "$GRAFT_REPO_ROOT"
cp "$(ls $O/kt/*/*kernel_stats.csv | head -1)" "$O/kernel_stats.csv"; rm -rf "$O/kt"
tail -1 "$O/kt.log" | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('bench under rocprof:', d['value'], 'img/s', d['ms_per_step'], 'ms/step')
except Exception as e: print('no bench line', e)"
python - "$O/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = 10.0   # 8 timed + 2 warmup
tot = 0.0
for r in rows:
    ms = float(r["TotalDurationNs"]) / 1e6 / steps
    tot += ms
    if ms > 0.05:
        print("%-84s %6.1f calls/step %9.1f us  %7.3f ms/step" % (r["Name"].replace("(anonymous namespace)::", "")[:84], float(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3, ms))
print("sum of kernel time per step: %.2f ms" % tot)
PY
