"""Write profiles/traffic.json (read by bench.py for `roofline.traffic` and the solver's VALU-issue fraction) from a
tools/pmc_derive.py table of a bench run under rocprofv3 --pmc:
    python tools/traffic_record.py <pmc_derived.csv> <frames per launch> <source label>
HBM-side bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE reports half of the bytes of wide coalesced
reads, MI355X_MICROARCH.md, HBM section).  Every record carries the hash of the kernel source it was measured on."""
import csv
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    src, frames, label = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    import bench
    rows = list(csv.DictReader(open(src)))
    recs = []

    def num(x):
        """float, or None for a counter the pass did not collect (never NaN: the record must be strict JSON)."""
        v = float(x)
        return v if math.isfinite(v) else None

    def record(form, kname, r):
        return dict(form=form, frames_per_launch=frames, kernel=kname, bytes_per_launch=int(float(r["hbm_GB"]) * 1e9),
                    l2_hit_rate=num(r["l2_hit"]), mfma_busy=num(r["mfma_busy"]), mfma_GFLOP_bf16=num(r["mfma_GFLOP_bf16"]),
                    dispatches=int(r["dispatches_per_pass"]),
                    profiled_us=float(r["profiled_us"]), source=label, kernel_source_sha256_16=bench.kernel_source_hash())

    forms = {"wino64": ("igemm_conv_kernel<1,1,128,512,0,128,1,0>", 64), "wino36": ("igemm_conv_kernel<1,1,128,512,0,128,1,0>", 36),
             "direct": ("igemm_conv_kernel<3,1,128,512,0,128,0,0>", 1)}
    for form, (kname, z) in forms.items():
        mtiles = -(-frames * (150 if z == 64 else 345 if z == 36 else 5400) // 128)
        grid = mtiles * 4 * z * 256
        for r in rows:
            if r["kernel"] == kname and int(r["grid_size"]) == grid:
                recs.append(record(form, kname, r))
    # split-bf16 Winograd GEMMs (persistent kernel: grid = CU count): the 512-channel layers have their own instantiation
    cand = [r for r in rows if r["kernel"].startswith("split_gemm_persist_kernel<512>")]
    if cand:
        r = max(cand, key=lambda r: float(r["profiled_us"]))
        recs.append(record("split64", r["kernel"], r))
    # the batched GEMMs of the 512 -> 512 layers read V as fp32 and split it inside the kernel.  Since round 4 they have an
    # instantiation of their own (ZB = 2): every dispatch of that name is ONE launch shape, so the row's means are per-shape means
    cand = [r for r in rows if r["kernel"].replace(" ", "") == "split_conv1x1_kernel<false,false,8,2,256>"]
    assert len(cand) <= 1, "the dominant kernel must be one row (one launch shape)"
    if cand:
        r = cand[0]
        assert int(r["dispatches_per_pass"]) % 9 == 0, "9 layers of that shape per forward pass: %s dispatches" % r["dispatches_per_pass"]
        recs.append(record("splitact64", r["kernel"], r))
    # round 5: the same launches as fp16 pairs, both operands by DMA (pair_gemm_kernel<512,0>: Cin = 512, one launch shape)
    # (round 6: the kernel has a third template argument - <512,0,2> = the 8-wave workgroup; <512,0,1> is the XL_PAIR_PP=1 form)
    cand = [r for r in rows if r["kernel"].replace(" ", "") in ("pair_gemm_kernel<512,0>", "pair_gemm_kernel<512,0,2>")]
    assert len(cand) <= 1, "the dominant kernel must be one row (one launch shape)"
    if cand:
        r = cand[0]
        assert int(r["dispatches_per_pass"]) % 9 == 0, "9 layers of that shape per forward pass: %s dispatches" % r["dispatches_per_pass"]
        recs.append(record("pair64", r["kernel"], r))
    solver = {}
    for r in rows:
        if r["kernel"].startswith("xl_dsac_forward_kernel"):
            solver[r["kernel"]] = dict(valu_busy=num(r["valu_busy"]), fp64_share_of_valu_instructions=num(r["fp64_valu_share"]),
                                       mean_waves_per_simd=num(r["mean_waves_per_simd"]), lds_conflict_share=num(r["lds_conflict_share"]),
                                       profiled_us=float(r["profiled_us"]), hbm_GB=num(r["hbm_GB"]))
    data = dict(records=recs, solver=dict(kernels=solver, frames_per_launch=frames, source=label))
    try:                                             # keep the records of other frame counts (bench.py --batch 24 / 44)
        with open(bench.TRAFFIC_JSON) as f:
            old = json.load(f)
        keep = [r for r in old.get("records", []) if r.get("frames_per_launch") != frames
                and r.get("kernel_source_sha256_16") == bench.kernel_source_hash()]
        data["records"] = keep + recs
        # MERGE the solver block: an HBM-only pass (tools/pmc_hbm2.txt) has no SQ counters - it must not replace the VALU
        # figures a full pass recorded (round 3 lost them that way)
        full = lambda sv: any(k.get("valu_busy") is not None for k in sv.get("kernels", {}).values())
        if not full(data["solver"]) and full(old.get("solver", {})):
            data["solver"] = old["solver"]
    except (OSError, ValueError):
        pass
    with open(bench.TRAFFIC_JSON, "w") as f:
        json.dump(data, f, indent=1, allow_nan=False)
    print(json.dumps(data, indent=1)[:1500])


if __name__ == "__main__":
    main()
