# Round-5 evidence (one gpurun call from the repo root).  Order matters: the PMC passes come first, their HBM-side bytes go
# into profiles/traffic.json (keyed by the hash of the kernel sources), and the bench line written afterwards reads
# `roofline.traffic` from that record.  Counter passes are separate --pmc passes with kernel tracing only.
set -ux
: "${GRAFT_REPO_ROOT:?run under gpurun (or export GRAFT_REPO_ROOT)}"
cd "$GRAFT_REPO_ROOT"
OUT="$GRAFT_REPO_ROOT/gpurun_out/r5"
rm -rf "$OUT"; mkdir -p "$OUT"
B="python $GRAFT_REPO_ROOT/bench.py --no-secondary --no-cpu-baseline"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -i $GRAFT_REPO_ROOT/tools/pmc_r5.txt --kernel-trace --output-format csv -d $OUT/pmc -- $B --steps 2 --warmup 1 > $OUT/pmc.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $OUT/pmc $OUT/pmc_summary.csv
python tools/pmc_derive.py $OUT/pmc_summary.csv $OUT/pmc_derived.csv > /dev/null
rm -f profiles/traffic.json
python tools/traffic_record.py $OUT/pmc_derived.csv 95 "profiles/r5_pmc_summary.csv (rocprofv3 -i tools/pmc_r5.txt over bench.py --steps 2)" > /dev/null
rm -rf $OUT/pmc
for nb in 47; do                                                  # HBM-side bytes of the dominant launch at other batch sizes
  cd /tmp
  rocprofv3 -i $GRAFT_REPO_ROOT/tools/pmc_hbm2.txt --kernel-trace --output-format csv -d $OUT/pmc$nb -- $B --batch $nb --steps 2 --warmup 1 > $OUT/pmc$nb.log 2>&1
  cd $GRAFT_REPO_ROOT
  python tools/pmc_summary.py $OUT/pmc$nb $OUT/pmc_summary_b$nb.csv
  python tools/pmc_derive.py $OUT/pmc_summary_b$nb.csv $OUT/pmc_derived_b$nb.csv > /dev/null
  python tools/traffic_record.py $OUT/pmc_derived_b$nb.csv $nb "rocprofv3 -i tools/pmc_hbm2.txt over bench.py --batch $nb --steps 2 (profiles/r5_pmc_derived_b$nb.csv)" > /dev/null
  rm -rf $OUT/pmc$nb
done
cp profiles/traffic.json $OUT/traffic.json
python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 600 $OUT/bench.json
python -c "
import json
def bad(c): raise ValueError(c)
d = json.loads(open('$OUT/bench.json').read(), parse_constant=bad)
print('strict JSON ok', d['value'], d['roofline']['traffic'], d['config']['dsac_pmc'])"
python bench.py --no-secondary --no-cpu-baseline --batch 47 > $OUT/bench_b47.json 2>/dev/null
python bench.py --no-secondary --no-cpu-baseline --batch 24 > $OUT/bench_b24.json 2>/dev/null
python bench.py --no-secondary --no-cpu-baseline --cnn-streams 2 > $OUT/bench_streams2.json 2>/dev/null
python bench.py --no-secondary --no-cpu-baseline --mlr 3 > $OUT/bench_mlr3.json 2>/dev/null
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $B > $OUT/kt.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob, json
f = glob.glob("$OUT/kt/*/*kernel_trace.csv")[0]
name = "pair_gemm_kernel<512,0>"
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(f)) if name in r["Kernel_Name"].replace(" ", "")]
json.dump({"kernel": name, "launches": len(d), "avg_ms": sum(d) / max(len(d), 1), "min_ms": min(d), "max_ms": max(d),
           "note": "every dispatch of this instantiation is the 64 batched GEMMs of a 512->512 layer (one launch shape)",
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
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ktm -- $B --mlr 3 --steps 5 > $OUT/ktm.log 2>&1
cd $GRAFT_REPO_ROOT
cp $(ls $OUT/ktm/*/*kernel_stats.csv | head -1) $OUT/mlr3_kernel_stats.csv
rm -rf $OUT/ktm
ls -la $OUT
