"""Add / replace one record of profiles/traffic.json (read by bench.py for `roofline.traffic`) from a merged PMC csv
(tools/merge_pmc.py output): HBM-side bytes per launch = mean FETCH_SIZE [KB] x 2 (gfx950: FETCH_SIZE reports half of
the bytes of wide coalesced reads, MI355X_MICROARCH.md §HBM) + mean WRITE_SIZE [KB], x 1024.
    python tools/traffic_record.py <merged_pmc.csv> <form: wino64|wino36|direct> <frames per launch>"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    src, form, frames = sys.argv[1], sys.argv[2], int(sys.argv[3])
    import bench
    vals = {}
    kernel = None
    for r in csv.DictReader(open(src)):
        vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        kernel = r["Kernel_Name"]
    fetch = sum(vals["FETCH_SIZE"]) / len(vals["FETCH_SIZE"])
    write = sum(vals["WRITE_SIZE"]) / len(vals["WRITE_SIZE"])
    rec = dict(form=form, frames_per_launch=frames, kernel=kernel, fetch_size_kb=round(fetch, 1), write_size_kb=round(write, 1),
               bytes_per_launch=int((2 * fetch + write) * 1024), dispatches=len(vals["FETCH_SIZE"]),
               source=os.path.relpath(os.path.abspath(src), ROOT), kernel_source_sha256_16=bench.kernel_source_hash())
    if "TCC_HIT_sum" in vals and "TCC_MISS_sum" in vals:
        h, m = sum(vals["TCC_HIT_sum"]), sum(vals["TCC_MISS_sum"])
        rec["l2_hit_rate"] = round(h / (h + m), 4)
    path = bench.TRAFFIC_JSON
    data = json.load(open(path)) if os.path.exists(path) else {"records": []}
    data["records"] = [r for r in data["records"] if not (r["form"] == form and r["frames_per_launch"] == frames)] + [rec]
    with open(path, "w") as f:
        json.dump(data, f, indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
