#!/bin/bash
# round 4, call b: GPU suite (all failures listed), 3-encoder A/B (last encoder block: Winograd + deferred applies vs the
# round-3 direct lowering), two CNN streams A/B
O=$GRAFT_REPO_ROOT/gpurun_out/r4b; mkdir -p $O
python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log
grep -E "^FAILED|^ERROR|passed|failed|rc=" $O/gputest.log | tail -12
grep -E "worst element|vs the reference|parameter gradients" $O/gputest.log | head
B="python bench.py --no-secondary --no-cpu-baseline"
$B --mlr 3 > $O/mlr3_new.json 2>$O/mlr3_new.err
XL_MLR_LAST_DIRECT=1 $B --mlr 3 > $O/mlr3_old.json 2>/dev/null
$B > $O/s1.json 2>/dev/null
$B --cnn-streams 2 > $O/s2.json 2>/dev/null
$B > $O/s1b.json 2>/dev/null
$B --cnn-streams 2 > $O/s2b.json 2>/dev/null
python - <<PY
import json
for n in ("mlr3_new", "mlr3_old", "s1", "s2", "s1b", "s2b"):
    try:
        d = json.load(open("$O/%s.json" % n)); print(n, d["value"], d["ms_per_step"], d["config"]["median_err_cm"], d["roofline"]["avg_launch_ms"])
    except Exception as e:
        print(n, "failed", e)
PY
tail -3 $O/mlr3_new.err
