#!/bin/bash
# usage: gpu_round.sh <outdir> : GPU suite, kernel statistics of the bench, the full bench line
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $O
python -m pytest tests -m gpu -q -x > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log; grep -E "^FAILED|passed|failed|rc=" $O/gputest.log | tail -6
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $GRAFT_REPO_ROOT/bench.py --no-secondary --no-cpu-baseline --steps 6 --warmup 2 > $O/kt.log 2>&1
cd $GRAFT_REPO_ROOT
cp $(ls $O/kt/*/*kernel_stats.csv | head -1) $O/kstats.csv; rm -rf $O/kt
python tools/kstats_show.py $O/kstats.csv | head -22
python bench.py --no-cpu-baseline > $O/bench_full.json 2> $O/bench_full.err
python - <<PY
import json
try:
    d = json.load(open("$O/bench_full.json")); c = d["config"]
    print(d["value"], d["ms_per_step"], {k: c[k] for k in c if k.startswith(("train16_full", "train16_ms", "latency_b1_ms", "b8"))})
except Exception as e:
    print("bench failed:", e)
PY
tail -3 $O/bench_full.err
