#!/bin/bash
# the complete -m gpu suite, smoke(), then the default bench line (all legs)
set -u
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
O="$GRAFT_REPO_ROOT/gpurun_out/r6_full"; mkdir -p "$O"
( time timeout 3000 python -m pytest tests/ -m gpu -x -q 2>&1 | tail -25 ) > "$O/tests.txt" 2>&1
cat "$O/tests.txt"
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
( time timeout 1500 python bench.py > "$O/bench.json" 2> "$O/bench.err" ) 2>&1 | tail -3
tail -3 "$O/bench.err"
python - <<PY
import json
d = json.load(open("$O/bench.json"))
c = d["config"]
print(d["value"], "img/s", d["ms_per_step"], "ms/step; dominant", d["roofline"].get("avg_launch_ms"), "frac", d["roofline"]["frac"])
for k in sorted(c):
    if any(s in k for s in ("train16", "latency", "dropin", "mlr3", "six_pass", "f32_mfma", "eager")):
        v = c[k]
        if not isinstance(v, (dict, list, str)) or len(str(v)) < 80:
            print("  ", k, v)
print("cpu_baseline", d.get("cpu_baseline", {}).get("value"))
PY
