"""CPU: the committed measurement records bench.py reads must belong to the sources at HEAD (round 6).

`roofline.traffic` and `config.dsac_pmc` of the bench line are look-ups in profiles/traffic.json - PMC counters cannot be read
from inside the process - keyed by launch form, frames per launch and a hash of the GEMM / conv kernel sources.  A kernel edit
without a new PMC collection (tools/collect_profiles_r6.sh) would leave the record stale, and bench.py would report it as
"(STALE ...)" or as null: this test makes that a red CPU suite instead of a silently weaker bench line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_traffic_record_belongs_to_the_kernel_sources_at_head():
    import bench
    with open(bench.TRAFFIC_JSON) as f:
        data = json.load(f, parse_constant=lambda c: (_ for _ in ()).throw(ValueError("non-finite constant %s in traffic.json" % c)))
    recs = data["records"]
    head = bench.kernel_source_hash()
    assert recs, "profiles/traffic.json holds no record"
    stale = [(r["form"], r["frames_per_launch"], r.get("kernel_source_sha256_16")) for r in recs if r.get("kernel_source_sha256_16") != head]
    assert not stale, "records measured on other kernel sources (HEAD = %s): %s - re-run tools/collect_profiles_r6.sh" % (head, stale)
    # the record of the default bench command: the fp16-pair Winograd GEMMs at 95 frames per launch
    byts, source = bench.lookup_traffic("pair64", 95)
    assert byts is not None and byts > 3.803e9 and "STALE" not in source, (byts, source)       # never below the algorithmic bytes
    assert byts < 1.5 * 3.803e9, byts
    sv = bench.lookup_solver_counters()
    assert sv is not None and sv["sample_and_score_valu_busy"] is not None and sv["sample_and_score_fp64_share_of_valu"] is not None
    roof = bench.solver_valu_roofline(95, 256, 5400, 3.9)
    assert 0.0 < roof["frac"] < 1.0 and roof["valu_busy_sample_and_score"] == sv["sample_and_score_valu_busy"]
