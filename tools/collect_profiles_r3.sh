# Round-3 evidence (run with gpurun from the repo root).  Order matters: the PMC passes come first, their HBM-side bytes go
# into profiles/traffic.json (keyed by the hash of the kernel sources), and the bench line written afterwards reads
# `roofline.traffic` from that record.  Counter passes are separate --pmc passes with kernel tracing only.
set -x
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3
rm -rf $OUT; mkdir -p $OUT
B="python $GRAFT_REPO_ROOT/bench.py --no-secondary --no-cpu-baseline"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -i $GRAFT_REPO_ROOT/tools/pmc_r2.txt --kernel-trace --output-format csv -d $OUT/pmc -- $B --steps 2 --warmup 1 > $OUT/pmc.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $OUT/pmc $OUT/pmc_summary.csv
python tools/pmc_derive.py $OUT/pmc_summary.csv $OUT/pmc_derived.csv > /dev/null
rm -f profiles/traffic.json
python tools/traffic_record.py $OUT/pmc_derived.csv 47 "profiles/r3_pmc_summary.csv (rocprofv3 -i tools/pmc_r2.txt over bench.py --steps 2)" > /dev/null
rm -rf $OUT/pmc
for nb in 24 44; do                                                  # HBM-side bytes of the dominant launch at other batch sizes
  cd /tmp
  rocprofv3 -i $GRAFT_REPO_ROOT/tools/pmc_hbm2.txt --kernel-trace --output-format csv -d $OUT/pmc$nb -- $B --batch $nb --steps 2 --warmup 1 > $OUT/pmc$nb.log 2>&1
  cd $GRAFT_REPO_ROOT
  python tools/pmc_summary.py $OUT/pmc$nb $OUT/pmc_summary_b$nb.csv
  python tools/pmc_derive.py $OUT/pmc_summary_b$nb.csv $OUT/pmc_derived_b$nb.csv > /dev/null
  python tools/traffic_record.py $OUT/pmc_derived_b$nb.csv $nb "rocprofv3 -i tools/pmc_hbm2.txt over bench.py --batch $nb --steps 2 (profiles/r3_pmc_derived_b$nb.csv)" > /dev/null
  rm -rf $OUT/pmc$nb
done
cp profiles/traffic.json $OUT/traffic.json
python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 900 $OUT/bench.json
python bench.py --no-secondary --no-cpu-baseline --batch 24 > $OUT/bench_b24.json 2>/dev/null
python bench.py --no-secondary --no-cpu-baseline --batch 44 > $OUT/bench_b44.json 2>/dev/null
XL_GEMM_SPLIT_BF16=0 python bench.py --no-secondary --no-cpu-baseline > $OUT/bench_fp32_mfma.json 2>/dev/null
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $B > $OUT/kt.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob, json
f = glob.glob("$OUT/kt/*/*kernel_trace.csv")[0]
name = "split_conv1x1_kernel<false,false,8,1,256>"
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(f)) if name in r["Kernel_Name"].replace(" ", "")]
ref = sorted(d)[int(0.75 * len(d))]                                  # (a quantile, not the maximum: one slow first launch would set the class)
big = [x for x in d if 0.6 * ref < x < 1.6 * ref]                    # the 512-channel layers (the 256-channel ones share the name)
json.dump({"kernel": name, "launches": len(big), "avg_ms": sum(big) / max(len(big), 1), "all_launches_of_that_name": len(d),
           "command": "rocprofv3 --kernel-trace --stats -- python bench.py --no-secondary --no-cpu-baseline"},
          open("$OUT/kernel_trace_dominant.json", "w"), indent=1)
print(open("$OUT/kernel_trace_dominant.json").read())
PY
cp $(ls $OUT/kt/*/*kernel_stats.csv | head -1) $OUT/bench_kernel_stats.csv
rm -rf $OUT/kt
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ktt -- python $GRAFT_REPO_ROOT/tools/train_step_bench.py --full --steps 5 > $OUT/ktt.log 2>&1
cd $GRAFT_REPO_ROOT
cp $(ls $OUT/ktt/*/*kernel_stats.csv | head -1) $OUT/train_step_kernel_stats.csv
rm -rf $OUT/ktt
tail -2 $OUT/ktt.log
ls -la $OUT
# small batches: the tile forms against each other (batch 1 and 8 through PipelinedLocalizer, best of three runs each),
# kernel statistics of the 8-frame plan (eager op list so that every kernel is traced by name), the stem layers alone
( python tools/latency_ab.py 1 8
  XL_NO_SMALL_TILES=1 python tools/latency_ab.py 1 8
  for f in 256 192 384 128; do XL_TILE_FORM_1X1=$f XL_TILE_FORM_WINO=$f python tools/latency_ab.py 1 8; done
  XL_CNN_GRAPH=0 python tools/latency_ab.py 1 8 ) 2>/dev/null | grep "B=1" > $OUT/latency_tile_forms.txt
cat $OUT/latency_tile_forms.txt
cd /tmp
XL_CNN_GRAPH=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt8 -- python $GRAFT_REPO_ROOT/tools/latency_ab.py 8 > $OUT/kt8.log 2>&1
cd $GRAFT_REPO_ROOT
cp $(ls $OUT/kt8/*/*kernel_stats.csv | head -1) $OUT/b8_kernel_stats.csv
rm -rf $OUT/kt8
( for c in 32 64 128; do python tools/stem_bench.py $c; done; XL_STEM_FORM=8x2 python tools/stem_bench.py 32; XL_STEM_FORM=c16 python tools/stem_bench.py 32
  for b in 4 8 16; do python tools/stem_bench.py 32 $b; done ) 2>/dev/null | grep "^stem" > $OUT/stem_bench.txt
cat $OUT/stem_bench.txt
ls -la $OUT
