#!/bin/bash
# usage: gpu_full.sh <outdir> : the GPU suite + the default bench (with secondary legs), summary on stdout
O=gpurun_out/$1; mkdir -p $O
python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log; grep -E "^FAILED|passed|failed" $O/gputest.log | tail -8
python bench.py ${@:2} > $O/bench_full.json 2> $O/bench_full.err
python - <<PY
import json
try:
    d = json.load(open("$O/bench_full.json")); c = d["config"]
    print(d["value"], d["ms_per_step"], {k: c[k] for k in c if k.startswith(("train16", "mlr3", "latency", "b8"))})
    print(d.get("roofline_forward")); print(d.get("cpu_baseline"))
except Exception as e:
    print("bench failed:", e)
PY
tail -3 $O/bench_full.err
