# Round-2 evidence (run with gpurun from the repo root): the bench line, rocprofv3 kernel statistics of the same command,
# and PMC passes (separate --pmc passes, kernel trace only - no sys/hip/hsa tracing) over a short bench run.
set -x
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2
mkdir -p $OUT
python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 400 $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python $GRAFT_REPO_ROOT/bench.py --no-secondary --no-cpu-baseline > $OUT/kt.log 2>&1
rocprofv3 -i $GRAFT_REPO_ROOT/tools/pmc_r2.txt --kernel-trace --output-format csv -d $OUT/pmc -- python $GRAFT_REPO_ROOT/bench.py --no-secondary --no-cpu-baseline --steps 2 --warmup 1 > $OUT/pmc.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $OUT/pmc $OUT/pmc_summary.csv
python tools/pmc_derive.py $OUT/pmc_summary.csv $OUT/pmc_derived.csv > /dev/null
FRAMES=$(python -c "import json;print(json.load(open('$OUT/bench.json'))['config']['batch_per_gpu'])")
cp profiles/traffic.json $OUT/traffic_before.json 2>/dev/null
python tools/traffic_record.py $OUT/pmc_derived.csv $FRAMES "profiles/r2_pmc_summary.csv (rocprofv3 -i tools/pmc_r2.txt over bench.py --steps 2)" > /dev/null
cp profiles/traffic.json $OUT/traffic.json
find $OUT -name "*kernel_stats.csv" | head -3
rm -rf $OUT/pmc/pmc_*/*/*kernel_trace.csv $OUT/pmc/pmc_*/*/*counter_collection.csv   # keep the merged summary only (size)
ls -la $OUT
