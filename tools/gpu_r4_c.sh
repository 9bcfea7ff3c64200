#!/bin/bash
# round 4, call c: GPU suite, then kernel statistics of the bench with the new / old output-transform chunking
O=$GRAFT_REPO_ROOT/gpurun_out/r4c; mkdir -p $O
python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log
grep -E "^FAILED|^ERROR|passed|failed|rc=" $O/gputest.log | tail -12
grep -E "worst element" $O/gputest.log | head -3
export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-secondary --no-cpu-baseline --steps 6 --warmup 2"
for v in new old; do
  cd /tmp
  if [ $v = old ]; then export XL_WINO_OUT_TPB16=1; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$v -- $B > $O/kt_$v.log 2>&1
  cd $GRAFT_REPO_ROOT
  cp $(ls $O/kt_$v/*/*kernel_stats.csv | head -1) $O/kstats_$v.csv; rm -rf $O/kt_$v
  echo "== $v"; python tools/kstats_show.py $O/kstats_$v.csv wino6_out wino6_in gn_final | head
  tail -1 $O/kt_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
unset XL_WINO_OUT_TPB16
python bench.py --no-secondary --no-cpu-baseline > $O/bench_new.json 2>/dev/null
XL_WINO_OUT_TPB16=1 python bench.py --no-secondary --no-cpu-baseline > $O/bench_old.json 2>/dev/null
python - <<PY
import json
for n in ("bench_new", "bench_old"):
    d = json.load(open("$O/%s.json" % n)); print(n, d["value"], d["ms_per_step"])
PY
