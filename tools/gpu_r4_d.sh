#!/bin/bash
# round 4, call d: the fused stem kernel - its bitwise test first, then the suite, kernel statistics and the bench A/B
O=$GRAFT_REPO_ROOT/gpurun_out/r4d; mkdir -p $O
timeout 600 python -m pytest tests/test_cnn_gpu.py -m gpu -q -k "fused_stem" > $O/stem_test.log 2>&1; echo "stem test rc=$?"; tail -15 $O/stem_test.log | cut -c1-300
timeout 900 python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log
grep -E "^FAILED|^ERROR|passed|failed|rc=" $O/gputest.log | tail -12
export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-secondary --no-cpu-baseline --steps 6 --warmup 2"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- $B > $O/kt.log 2>&1
cd $GRAFT_REPO_ROOT
cp $(ls $O/kt/*/*kernel_stats.csv | head -1) $O/kstats.csv; rm -rf $O/kt
python tools/kstats_show.py $O/kstats.csv stem12 conv1_mfma s2_kernel gn_stats
python bench.py --no-secondary --no-cpu-baseline > $O/bench_new.json 2>$O/bench_new.err
XL_NO_STEM12=1 python bench.py --no-secondary --no-cpu-baseline > $O/bench_old.json 2>/dev/null
python bench.py --no-secondary --no-cpu-baseline > $O/bench_new2.json 2>/dev/null
XL_NO_STEM12=1 python bench.py --no-secondary --no-cpu-baseline > $O/bench_old2.json 2>/dev/null
python - <<PY
import json
for n in ("bench_new", "bench_old", "bench_new2", "bench_old2"):
    try:
        d = json.load(open("$O/%s.json" % n)); print(n, d["value"], d["ms_per_step"], d["config"]["median_err_cm"])
    except Exception as e:
        print(n, "failed", e)
PY
tail -3 $O/bench_new.err
