"""Derived per-kernel figures from a tools/pmc_summary.py table (one bench run under rocprofv3 --pmc, tools/pmc_r2.txt):
    python tools/pmc_derive.py <pmc_summary.csv> <out.csv>
Conventions (MI355X_MICROARCH.md): HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (KB; FETCH_SIZE counts half of the bytes of
wide coalesced reads on gfx950); SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles;
GRBM_GUI_ACTIVE is summed over the 8 XCDs; busy fractions are per SIMD: counter / (1024 SIMDs x GRBM_GUI_ACTIVE / 8).
Durations are the ones of the profiled passes (kernels run serialised and a few % slower than un-profiled)."""
import collections
import csv
import sys


def main():
    src, dst = sys.argv[1:3]
    by = collections.defaultdict(dict)
    for r in csv.DictReader(open(src)):
        by[(r["kernel"], int(r["grid_size"]), int(r["vgprs"]), int(r["lds_bytes"]))][r["counter"]] = (
            float(r["mean"]), int(r["dispatches"]), float(r["mean_profiled_us"]))
    cols = ["kernel", "grid_size", "vgprs", "lds_bytes", "dispatches_per_pass", "profiled_us", "hbm_GB", "hbm_TBps", "l2_hit",
            "valu_busy", "mfma_busy", "mean_waves_per_simd", "valu_insts", "fp64_valu_share", "fp32_fma_share",
            "int32_share", "salu_per_valu", "lds_conflict_share", "mfma_GFLOP_f32", "mfma_GFLOP_bf16"]
    out = []
    for (k, g, vg, lds), c in sorted(by.items()):
        def v(n):
            return c[n][0] if n in c else float("nan")
        if "FETCH_SIZE" not in c or "GRBM_GUI_ACTIVE" not in c:
            continue
        us = c["FETCH_SIZE"][2]
        simd_cycles = 1024.0 * v("GRBM_GUI_ACTIVE") / 8.0
        hbm = (2.0 * v("FETCH_SIZE") + v("WRITE_SIZE")) * 1024.0
        nv = v("SQ_INSTS_VALU")
        f64 = v("SQ_INSTS_VALU_FMA_F64") + v("SQ_INSTS_VALU_ADD_F64") + v("SQ_INSTS_VALU_MUL_F64") + v("SQ_INSTS_VALU_TRANS_F64")
        out.append([k, g, vg, lds, c["FETCH_SIZE"][1], "%.1f" % us, "%.4f" % (hbm / 1e9), "%.2f" % (hbm / us / 1e6),
                    "%.3f" % (v("TCC_HIT_sum") / max(v("TCC_HIT_sum") + v("TCC_MISS_sum"), 1.0)),
                    "%.3f" % (4.0 * v("SQ_ACTIVE_INST_VALU") / simd_cycles), "%.3f" % (v("SQ_VALU_MFMA_BUSY_CYCLES") / simd_cycles),
                    "%.2f" % (4.0 * v("SQ_WAVE_CYCLES") / simd_cycles), "%.4g" % nv, "%.3f" % (f64 / max(nv, 1.0)),
                    "%.3f" % (v("SQ_INSTS_VALU_FMA_F32") / max(nv, 1.0)), "%.3f" % (v("SQ_INSTS_VALU_INT32") / max(nv, 1.0)),
                    "%.3f" % (v("SQ_INSTS_SALU") / max(nv, 1.0)),
                    "%.4f" % (v("SQ_LDS_BANK_CONFLICT") / max(v("SQ_LDS_IDX_ACTIVE"), 1.0)),
                    "%.2f" % (v("SQ_INSTS_VALU_MFMA_MOPS_F32") * 512.0 / 1e9),
                    # (16-bit matrix pipe: bf16 + fp16 MFMA operations - round 5's pair kernels are fp16; a pass that did not collect
                    #  the F16 counter reads it as 0)
                    "%.2f" % ((v("SQ_INSTS_VALU_MFMA_MOPS_BF16") + (v("SQ_INSTS_VALU_MFMA_MOPS_F16") if "SQ_INSTS_VALU_MFMA_MOPS_F16" in c else 0.0))
                              * 512.0 / 1e9)])
    with open(dst, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(cols)
        w.writerows(out)
    for r in out:
        print(" ".join(str(x) for x in r))


if __name__ == "__main__":
    main()
