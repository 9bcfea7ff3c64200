#!/bin/bash
# Round-6 evidence (one gpurun call from the repo root).  Order as in round 5: the PMC passes first, their HBM-side bytes go into
# profiles/traffic.json (keyed by the hash of the kernel sources), the bench line written afterwards reads `roofline.traffic` from
# that record.  Counter passes are separate --pmc passes with kernel tracing only.  Then the round's A/B records: the epilogue store
# pattern and the two-workgroups-per-CU ("ping-pong") form of the dominant GEMM, wall time AND SQ_WAVE_CYCLES, same session, random data.
set -ux
: "${GRAFT_REPO_ROOT:?run under gpurun (or export GRAFT_REPO_ROOT)}"
cd "$GRAFT_REPO_ROOT"
OUT="$GRAFT_REPO_ROOT/gpurun_out/r6"
rm -rf "$OUT"; mkdir -p "$OUT"
B="python $GRAFT_REPO_ROOT/bench.py --no-secondary --no-cpu-baseline"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -i "$GRAFT_REPO_ROOT/tools/pmc_r5.txt" --kernel-trace --output-format csv -d "$OUT/pmc" -- $B --steps 2 --warmup 1 > "$OUT/pmc.log" 2>&1
cd "$GRAFT_REPO_ROOT"
python tools/pmc_summary.py "$OUT/pmc" "$OUT/pmc_summary.csv"
python tools/pmc_derive.py "$OUT/pmc_summary.csv" "$OUT/pmc_derived.csv" > /dev/null
rm -f profiles/traffic.json
python tools/traffic_record.py "$OUT/pmc_derived.csv" 95 "profiles/r6_pmc_summary.csv (rocprofv3 -i tools/pmc_r5.txt over bench.py --steps 2)" > /dev/null
rm -rf "$OUT/pmc"
cd /tmp
rocprofv3 -i "$GRAFT_REPO_ROOT/tools/pmc_hbm2.txt" --kernel-trace --output-format csv -d "$OUT/pmc47" -- $B --batch 47 --steps 2 --warmup 1 > "$OUT/pmc47.log" 2>&1
cd "$GRAFT_REPO_ROOT"
python tools/pmc_summary.py "$OUT/pmc47" "$OUT/pmc_summary_b47.csv"
python tools/pmc_derive.py "$OUT/pmc_summary_b47.csv" "$OUT/pmc_derived_b47.csv" > /dev/null
python tools/traffic_record.py "$OUT/pmc_derived_b47.csv" 47 "rocprofv3 -i tools/pmc_hbm2.txt over bench.py --batch 47 --steps 2 (profiles/r6_pmc_derived_b47.csv)" > /dev/null
rm -rf "$OUT/pmc47"
cp profiles/traffic.json "$OUT/traffic.json"
python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -c 600 "$OUT/bench.json"
python -c "
import json
def bad(c): raise ValueError(c)
d = json.loads(open('$OUT/bench.json').read(), parse_constant=bad)
print('strict JSON ok', d['value'], d['roofline']['traffic'], d['config']['dsac_valu_roofline'])"
python bench.py --no-secondary --no-cpu-baseline --batch 47 > "$OUT/bench_b47.json" 2>/dev/null
python bench.py --no-secondary --no-cpu-baseline --batch 190 > "$OUT/bench_b190.json" 2>/dev/null
python bench.py --no-secondary --no-cpu-baseline --cnn-streams 2 > "$OUT/bench_streams2.json" 2>/dev/null
python bench.py --no-secondary --no-cpu-baseline --mlr 3 > "$OUT/bench_mlr3.json" 2>/dev/null
XL_PAIR_PP=1 python bench.py --no-secondary --no-cpu-baseline > "$OUT/bench_pingpong.json" 2>/dev/null
XL_PAIR_VAR=4 python bench.py --no-secondary --no-cpu-baseline > "$OUT/bench_store32.json" 2>/dev/null
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- $B > "$OUT/kt.log" 2>&1
cd "$GRAFT_REPO_ROOT"
python - <<PY
import csv, glob, json
f = glob.glob("$OUT/kt/*/*kernel_trace.csv")[0]
name = "pair_gemm_kernel<512,0,2>"
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(f)) if name in r["Kernel_Name"].replace(" ", "")]
json.dump({"kernel": name, "launches": len(d), "avg_ms": sum(d) / max(len(d), 1), "min_ms": min(d), "max_ms": max(d),
           "note": "every dispatch of this instantiation is the 64 batched GEMMs of a 512->512 layer (one launch shape)",
           "command": "rocprofv3 --kernel-trace --stats -- python bench.py --no-secondary --no-cpu-baseline"},
          open("$OUT/kernel_trace_dominant.json", "w"), indent=1)
print(open("$OUT/kernel_trace_dominant.json").read())
PY
cp "$(ls $OUT/kt/*/*kernel_stats.csv | head -1)" "$OUT/bench_kernel_stats.csv"
rm -rf "$OUT/kt"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ktt" -- python "$GRAFT_REPO_ROOT/tools/train_step_bench.py" --full --steps 5 > "$OUT/ktt.log" 2>&1
cd "$GRAFT_REPO_ROOT"
cp "$(ls $OUT/ktt/*/*kernel_stats.csv | head -1)" "$OUT/train_step_kernel_stats.csv"
rm -rf "$OUT/ktt"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ktm" -- $B --mlr 3 --steps 5 > "$OUT/ktm.log" 2>&1
cd "$GRAFT_REPO_ROOT"
cp "$(ls $OUT/ktm/*/*kernel_stats.csv | head -1)" "$OUT/mlr3_kernel_stats.csv"
rm -rf "$OUT/ktm"
# ---- the dominant GEMM alone on random data: store pattern and ping-pong form, wall time and wave cycles
: > "$OUT/gemm_ab.txt"
for v in "default:" "store32:XL_PAIR_VAR=4" "pingpong:XL_PAIR_PP=1" "nostores:XL_PAIR_DBG=8" "mfma_only:XL_PAIR_DBG=15"; do
  name=${v%%:*}; e=${v#*:}
  for rep in 1 2 3; do
    echo "== $name rep $rep" >> "$OUT/gemm_ab.txt"
    env XL_PAIR_ONLY_DMA=1 $e python tools/pair_gemm_bench.py 95 2>&1 | grep -E "^pair_dma" >> "$OUT/gemm_ab.txt"
  done
  cd /tmp
  env XL_PAIR_ONLY_DMA=1 $e rocprofv3 -i "$GRAFT_REPO_ROOT/tools/pmc_gemm_r6.txt" --kernel-trace --output-format csv -d "$OUT/pmc_$name" -- python "$GRAFT_REPO_ROOT/tools/pair_gemm_bench.py" 95 > "$OUT/pmc_$name.log" 2>&1
  cd "$GRAFT_REPO_ROOT"
  python tools/pmc_summary.py "$OUT/pmc_$name" "$OUT/pmc_gemm_$name.csv" > /dev/null
  rm -rf "$OUT/pmc_$name"
done
python - <<PY
import csv, collections
rows = []
for name in ("default", "store32", "pingpong", "nostores", "mfma_only"):
    c = collections.defaultdict(dict)
    for r in csv.DictReader(open("$OUT/pmc_gemm_%s.csv" % name)):
        if r["kernel"].startswith("pair_gemm_kernel"):
            c[r["kernel"]][r["counter"]] = (float(r["mean"]), float(r["mean_profiled_us"]), int(r["dispatches"]))
    for k, v in c.items():
        g = v["GRBM_GUI_ACTIVE"][0] / 8.0
        rows.append(dict(form=name, kernel=k, dispatches=v["GRBM_GUI_ACTIVE"][2], profiled_us=round(v["GRBM_GUI_ACTIVE"][1], 1),
                         clock_GHz=round(g / v["GRBM_GUI_ACTIVE"][1] / 1e3, 3), sq_wave_cycles=int(v["SQ_WAVE_CYCLES"][0]),
                         waves_per_simd=round(4.0 * v["SQ_WAVE_CYCLES"][0] / (1024.0 * g), 3),
                         mfma_busy=round(v["SQ_VALU_MFMA_BUSY_CYCLES"][0] / (1024.0 * g), 4),
                         f16_GFLOP=round(v["SQ_INSTS_VALU_MFMA_MOPS_F16"][0] * 512.0 / 1e9, 1)))
with open("$OUT/gemm_ab.csv", "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
for r in rows: print(r)
PY
cat "$OUT/gemm_ab.txt"
XL_PAIR_CLK=1 XL_PAIR_ONLY_DMA=1 python tools/pair_gemm_bench.py 95 2>&1 | grep "pair clk" | tail -3 > "$OUT/pair_clk.txt"
XL_PAIR_VAR=4 XL_PAIR_CLK=1 XL_PAIR_ONLY_DMA=1 python tools/pair_gemm_bench.py 95 2>&1 | grep "pair clk" | tail -3 | sed 's/^/[store32] /' >> "$OUT/pair_clk.txt"
cat "$OUT/pair_clk.txt"
# ---- single frame: the drop-in loop statement by statement, the kernel timeline of a frame
python tools/dropin_breakdown.py > "$OUT/dropin_breakdown.txt" 2>&1; tail -8 "$OUT/dropin_breakdown.txt"
bash tools/r6/b1_trace.sh > /dev/null 2>&1; cp gpurun_out/r6_b1/timeline.txt "$OUT/b1_timeline.txt"; tail -3 "$OUT/b1_timeline.txt"
# ---- the stem kernels alone: counters (LDS conflicts) and the three-workgroups-per-CU form
bash tools/r6/stem_pmc.sh > "$OUT/stem_pmc.txt" 2>&1
for w in 2 3; do for rep in 1 2 3; do echo "WGS $w: $(XL_STEM12_WGS=$w python tools/stem12_bench.py 95 2>&1 | grep '^stem12')"; done; done > "$OUT/stem12_wgs.txt"
cat "$OUT/stem12_wgs.txt"
ls -la "$OUT"
