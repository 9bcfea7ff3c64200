#!/bin/bash
set -u
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
O="$GRAFT_REPO_ROOT/gpurun_out/r6_check1"; mkdir -p "$O"
timeout 1200 python -m pytest tests/test_pair_gpu.py tests/test_bench_plan_gpu.py tests/test_reference_fixtures.py -m gpu -x -q -s 2>&1 | tail -60 > "$O/tests.txt"
cat "$O/tests.txt"
timeout 600 python bench.py --no-secondary --no-cpu-baseline > "$O/bench.json" 2> "$O/bench.err"; tail -3 "$O/bench.err"
python -c "
import json; d=json.load(open('$O/bench.json')); r=d['roofline']; print(d['value'], 'img/s; ms/step', d['ms_per_step'], 'dominant', r.get('avg_launch_ms'), 'frac', r['frac'])"
