#!/bin/bash
O=gpurun_out/r3f; mkdir -p $O
python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log; grep -E "^FAILED|passed|failed" $O/gputest.log | tail -8
python bench.py --no-cpu-baseline > $O/bench_full.json 2> $O/bench_full.err
XL_NO_SPLIT_TRAIN=1 python bench.py --no-cpu-baseline --steps 3 > $O/bench_full_nosplittrain.json 2> $O/bench_full_nosplittrain.err
python - <<'PY'
import json
for n in ("bench_full", "bench_full_nosplittrain"):
    try:
        d = json.load(open("gpurun_out/r3f/%s.json" % n))
        c = d["config"]
        print(n, d["value"], {k: c[k] for k in c if k.startswith("train16") or k.startswith("mlr3")})
    except Exception as e:
        print(n, "failed", e)
PY
tail -5 $O/bench_full.err
