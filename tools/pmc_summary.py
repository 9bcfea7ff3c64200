"""Summarise the per-pass rocprofv3 counter_collection.csv files of one `-i <pmc file>` run over bench.py into a compact
table: one row per (kernel of this library, grid size, counter) with the number of dispatches, the mean / min / max
counter value and the mean duration of those dispatches in the profiled pass.  Persistent kernels launch one workgroup per CU
whatever the problem, so a dispatch table can tell their launch shapes apart by NAME only: the dominant kernel (the 64 batched
GEMMs of a 512 -> 512 Winograd layer) has an instantiation of its own, split_conv1x1_kernel<false,false,8,2,256>, and its row
holds launches of exactly that shape.  The other persistent kernels serve several layers; their rows are additionally keyed by
a duration class ("@<us>", powers of 4) - a convenience for reading the table, NOT a shape key: such a row may mix layers.
    python tools/pmc_summary.py <rocprof output dir> <out.csv>
Derived figures (GB/s, VALU busy, ...) are computed by tools/pmc_derive.py from the table."""
import csv
import glob
import math
import os
import re
import sys

csv.field_size_limit(1 << 30)
NAME = re.compile(r"\(anonymous namespace\)::([A-Za-z0-9_]+(?:<[^>]*>)?)")


def main():
    src, dst = sys.argv[1:3]
    acc = {}
    for f in sorted(glob.glob(os.path.join(src, "pmc_*", "*", "*counter_collection.csv"))):
        for r in csv.DictReader(open(f)):
            m = NAME.search(r["Kernel_Name"][:200])
            if not m or "at::native" in r["Kernel_Name"][:40]:
                continue
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            name = m.group(1).replace(" ", "")
            if name.startswith("split_conv1x1_kernel<false,false,") and name.split(",")[3] == "2":
                pass                                  # ZB = 2: one launch shape by construction (Cin = Cout = 512, Z = 64)
            elif name.startswith("pair_gemm_kernel<512"):
                pass                                  # round 5: Cin = 512 -> the 64 GEMMs of a 512 -> 512 layer, one launch shape
            elif "persist" in name or name.startswith(("split_conv1x1_kernel", "split_conv3x3s2_kernel", "pair_conv1x1_kernel",
                                                      "pair_conv1x1_res_kernel", "pair_conv3x3s2_kernel", "pair_gemm_kernel")):
                name += "@%dus" % (4 ** round(math.log(max(d, 1.0), 4)))  # (reading aid only, see the header)
            key = (name, int(r["Grid_Size"]), int(r["VGPR_Count"]) + int(r["Accum_VGPR_Count"]),
                   int(r["LDS_Block_Size"]), r["Counter_Name"])
            v = float(r["Counter_Value"])
            a = acc.setdefault(key, [0, 0.0, v, v, 0.0])
            a[0] += 1; a[1] += v; a[2] = min(a[2], v); a[3] = max(a[3], v); a[4] += d
    with open(dst, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "grid_size", "vgprs", "lds_bytes", "counter", "dispatches", "mean", "min", "max",
                    "mean_profiled_us"])
        for (k, g, vg, lds, c), a in sorted(acc.items()):
            w.writerow([k, g, vg, lds, c, a[0], "%.6g" % (a[1] / a[0]), "%.6g" % a[2], "%.6g" % a[3], "%.2f" % (a[4] / a[0])])
    print("%d rows -> %s" % (len(acc), dst))


if __name__ == "__main__":
    main()
