#!/bin/bash
# PMC counters of the fused stem kernel alone (tools/stem12_bench.py), raw means per dispatch
set -u
: "${GRAFT_REPO_ROOT:?}"
O="$GRAFT_REPO_ROOT/gpurun_out/r6_stem_pmc"; rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -i "$GRAFT_REPO_ROOT/tools/pmc_r5.txt" --kernel-trace --output-format csv -d "$O/pmc" -- python "$GRAFT_REPO_ROOT/tools/stem12_bench.py" 95 > "$O/pmc.log" 2>&1
cd "$GRAFT_REPO_ROOT"
python tools/pmc_summary.py "$O/pmc" "$O/pmc_summary.csv" > /dev/null
rm -rf "$O/pmc"
python - "$O/pmc_summary.csv" <<'PY'
import csv, sys, collections
by = collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    if "stem12" in r["kernel"] or "conv1_mfma" in r["kernel"] or "pair_conv3x3s2" in r["kernel"]:
        by[r["kernel"][:60]][r["counter"]] = (float(r["mean"]), float(r["mean_profiled_us"]))
for k, c in by.items():
    g = c["GRBM_GUI_ACTIVE"][0] / 8.0
    print(k, "profiled us %.1f, cycles/XCD %.0f (%.2f GHz)" % (c["GRBM_GUI_ACTIVE"][1], g, g / c["GRBM_GUI_ACTIVE"][1] / 1e3))
    simd = 1024.0 * g; cu = 256.0 * g
    for n, (v, _) in sorted(c.items()):
        print("   %-34s %14.0f   per SIMD-cycle %.4f   per CU-cycle %.4f" % (n, v, v / simd, v / cu))
PY
