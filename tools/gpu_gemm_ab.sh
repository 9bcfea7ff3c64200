#!/bin/bash
# A/B of the dominant launch (the 64 batched Winograd GEMMs of a 512 -> 512 layer, 47 frames): three forms of the same product
#   r3     split_conv1x1_kernel<false,false,8,1,256>: V fp32, split inside the kernel (8 waves of 128 x 64)   [default]
#   r2_8w  split_gemm_persist_kernel<512>: V as bf16 planes by LDS-DMA, 8 waves of 128 x 64               [XL_WINO_V_SPLIT=1]
#   r2_4w  split_gemm_persist4_kernel<512>: the same with 4 waves of 128 x 128 (a third fewer LDS reads per MFMA)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3ab; rm -rf $O; mkdir -p $O
XL_GEMM_PERSIST_WAVES=4 XL_WINO_V_SPLIT=1 python -m pytest tests/test_cnn_gpu.py -m gpu -q -k "winograd_split_bf16_gemm or golden or full_size" > $O/test_4w.log 2>&1; tail -2 $O/test_4w.log
XL_GEMM_PERSIST_WAVES=4 XL_WINO_V_SPLIT=1 python -m pytest tests/test_full_size_gpu.py -m gpu -q -k "three_encoder" >> $O/test_4w.log 2>&1; tail -1 $O/test_4w.log
B="python $GRAFT_REPO_ROOT/bench.py --no-secondary --no-cpu-baseline"
for v in r3: r2_8w:XL_WINO_V_SPLIT=1 r2_4w:XL_WINO_V_SPLIT=1,XL_GEMM_PERSIST_WAVES=4; do
  name=${v%%:*}; envs=${v#*:}; envcmd="env"; IFS=',' read -ra E <<< "$envs"; for e in "${E[@]}"; do [[ -n "$e" ]] && envcmd="$envcmd $e"; done
  $envcmd $B > $O/bench_$name.json 2> /dev/null
  python -c "import json; d=json.load(open('$O/bench_$name.json')); r=d['roofline']; print('$name', d['value'], 'img/s; dominant launch', r['avg_launch_ms'], 'ms, frac', r['frac'])"
  cd /tmp && export TMPDIR=/tmp
  $envcmd rocprofv3 -i $GRAFT_REPO_ROOT/tools/pmc_gemm_ab.txt --kernel-trace --output-format csv -d $O/pmc_$name -- $B --steps 2 --warmup 1 > $O/pmc_$name.log 2>&1
  cd $GRAFT_REPO_ROOT
  python tools/pmc_summary.py $O/pmc_$name $O/pmc_summary_$name.csv > /dev/null
  rm -rf $O/pmc_$name
done
python - <<'PY'
import csv, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r3ab")
rows = []
for name, pat in (("r3", "split_conv1x1_kernel<false,false,8,1,256>"), ("r2_8w", "split_gemm_persist_kernel<512>"), ("r2_4w", "split_gemm_persist4_kernel<512>")):
    by = {}
    for r in csv.DictReader(open(os.path.join(O, "pmc_summary_%s.csv" % name))):
        if r["kernel"].startswith(pat):
            by.setdefault(r["kernel"], {})[r["counter"]] = (float(r["mean"]), float(r["mean_profiled_us"]), int(r["dispatches"]))
    if not by:
        continue
    k = max(by, key=lambda k: by[k]["GRBM_GUI_ACTIVE"][1])              # the 512-channel class (longest)
    c = by[k]
    us = c["GRBM_GUI_ACTIVE"][1]
    clk = c["GRBM_GUI_ACTIVE"][0] / 8.0 / us / 1e3                      # GHz: summed over 8 XCDs
    busy = c["SQ_VALU_MFMA_BUSY_CYCLES"][0] / (1024.0 * c["GRBM_GUI_ACTIVE"][0] / 8.0)
    mf = c["SQ_INSTS_VALU_MFMA_MOPS_BF16"][0] * 512.0 / 1e9
    rows.append(dict(form=name, kernel=k, dispatches=c["GRBM_GUI_ACTIVE"][2], profiled_us=round(us, 1), clock_GHz=round(clk, 3),
                     mfma_busy=round(busy, 4), busy_x_clock_GHz=round(busy * clk, 3), bf16_GFLOP=round(mf, 1),
                     lds_instructions=int(c["SQ_INSTS_LDS"][0]), lds_instr_per_GFLOP=round(c["SQ_INSTS_LDS"][0] / mf, 1),
                     lds_active_share=round(4.0 * c["SQ_ACTIVE_INST_LDS"][0] / (1024.0 * c["GRBM_GUI_ACTIVE"][0] / 8.0), 4),
                     lds_bank_conflict=int(c["SQ_LDS_BANK_CONFLICT"][0]),
                     hbm_GB=round((2 * c["FETCH_SIZE"][0] + c["WRITE_SIZE"][0]) * 1024 / 1e9, 3)))
with open(os.path.join(O, "gemm_ab.csv"), "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
    w.writeheader(); w.writerows(rows)
for r in rows:
    print(r)
PY
