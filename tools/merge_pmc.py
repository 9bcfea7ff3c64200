"""Merge the per-pass rocprofv3 counter_collection.csv files of one `-i tools/pmc_hbm.txt` run into the compact
profiles/*_pmc.csv form (pass, dispatch, kernel, counter, value), keeping the rows of one kernel.
usage: python tools/merge_pmc.py <rocprof output dir> <kernel substring> <out.csv>"""
import csv
import glob
import os
import sys


def main():
    src, needle, dst = sys.argv[1:4]
    rows = []
    newest = {}                                       # one run per pass directory: the most recent one
    for f in glob.glob(os.path.join(src, "pmc_*", "*", "*counter_collection.csv")):
        p = f[len(src):].strip("/").split("/")[0]
        if p not in newest or os.path.getmtime(f) > os.path.getmtime(newest[p]):
            newest[p] = f
    for p, f in sorted(newest.items()):
        for r in csv.DictReader(open(f)):
            if needle in r["Kernel_Name"]:
                rows.append((p, int(r["Dispatch_Id"]), r["Kernel_Name"][:80], r["Counter_Name"], r["Counter_Value"]))
    rows.sort(key=lambda r: (r[0], r[3], r[1]))
    with open(dst, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["pass", "Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"])
        w.writerows(rows)
    print("%d rows -> %s" % (len(rows), dst))


if __name__ == "__main__":
    main()
