# Round-2 evidence (run with gpurun from the repo root).  Order matters: the PMC passes come first, their HBM-side bytes
# go into profiles/traffic.json (keyed by the hash of the kernel sources), and the bench line written afterwards reads
# `roofline.traffic` from that record.  Counter passes are separate --pmc passes with kernel tracing only.
set -x
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -i $GRAFT_REPO_ROOT/tools/pmc_r2.txt --kernel-trace --output-format csv -d $OUT/pmc -- python $GRAFT_REPO_ROOT/bench.py --no-secondary --no-cpu-baseline --steps 2 --warmup 1 > $OUT/pmc.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $OUT/pmc $OUT/pmc_summary.csv
python tools/pmc_derive.py $OUT/pmc_summary.csv $OUT/pmc_derived.csv > /dev/null
FRAMES=$(grep -o '"batch_per_gpu": [0-9]*' $OUT/pmc.log | head -1 | grep -o '[0-9]*$')
python tools/traffic_record.py $OUT/pmc_derived.csv $FRAMES "profiles/r2_pmc_summary.csv (rocprofv3 -i tools/pmc_r2.txt over bench.py --steps 2)" > /dev/null
cp profiles/traffic.json $OUT/traffic.json
rm -rf $OUT/pmc                                                    # keep the merged summary only (size)
python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 600 $OUT/bench.json
XL_GEMM_SPLIT_BF16=0 python bench.py --no-secondary --no-cpu-baseline > $OUT/bench_fp32_mfma.json 2>/dev/null
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python $GRAFT_REPO_ROOT/bench.py --no-secondary --no-cpu-baseline > $OUT/kt.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob, json
f = glob.glob("$OUT/kt/*/*kernel_trace.csv")[0]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(f)) if "split_gemm_persist_kernel<512>" in r["Kernel_Name"].replace(" ", "")]
big = d
json.dump({"kernel": "split_gemm_persist_kernel<512>", "launches": len(d), "avg_ms": sum(big) / max(len(big), 1), "command": "rocprofv3 --kernel-trace --stats -- python bench.py --no-secondary --no-cpu-baseline"},
          open("$OUT/kernel_trace_dominant.json", "w"), indent=1)
print(open("$OUT/kernel_trace_dominant.json").read())
PY
cp $(ls $OUT/kt/*/*kernel_stats.csv | head -1) $OUT/bench_kernel_stats.csv
rm -rf $OUT/kt
ls -la $OUT
