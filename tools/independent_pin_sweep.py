"""Independent pin of the solver oracle over the whole tools/parity_sweep.py frame set:
tests/indep_dsac.py (numpy / scipy, reference control flow) against oracle/dsac_oracle.c, frame by frame.
    python tools/independent_pin_sweep.py [K=1536] [n_hyp=256] [workers=all cores]
Writes profiles/r2_independent_pin.json (summary + every disagreeing frame with its cause) and prints the summary."""
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np                     # noqa: E402


def job(args):
    rho, i, n_hyp = args
    os.environ["OMP_NUM_THREADS"] = "1"
    import indep_dsac
    from oracle import dsac_oracle as xo
    from crossloc_amd import synth
    xo.set_num_threads(1)
    # the frames of tools/parity_sweep.py: make_batch(9000 + int(rho * 10) * 1000 + s, 64) keyed image0 = s
    sc = synth.make_scene(9000 + int(rho * 10) * 1000 + i, noise=0.5, outlier_ratio=rho)
    rec = indep_dsac.compare_with_oracle(sc["coords"], n_hyp, i, gt_pose=sc["pose"])
    rec["rho"] = rho
    return rec


def cause(r):
    out = []
    if not r["cells_equal"]:
        out.append("%d hypotheses took a different accepted try: an accept/reject decision (4 reprojection errors < thr, "
                   "or P3P solvable at all) flipped between the two P3P implementations" % r["n_cell_mismatch"])
    if r["winner_indep"] != r["winner_oracle"]:
        out.append("different winner: top-2 score gap %.3g" % r["score_gap_top2"])
    if r["rounds_indep"] != r["rounds_oracle"] or r["inliers_indep"] != r["inliers_oracle"]:
        out.append("refinement path: %d rounds / %d inliers (scipy MINPACK to convergence) vs %d / %d (CvLevMarq state "
                   "machine, <= 20 iterations per round)" % (r["rounds_indep"], r["inliers_indep"], r["rounds_oracle"],
                                                             r["inliers_oracle"]))
    if r.get("worst_score_cause"):
        out.append("largest score difference %.3g: %s" % (r["max_dscore_clean"], r["worst_score_cause"]))
    return "; ".join(out)


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 1536
    n_hyp = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    workers = int(sys.argv[3]) if len(sys.argv) > 3 else (os.cpu_count() or 1)
    jobs = [(rho, i, n_hyp) for rho in (0.6, 0.3, 0.0) for i in range(K // 3)]
    t0 = time.time()
    with mp.Pool(workers) as pool:
        recs = pool.map(job, jobs, chunksize=4)
    summary = {}
    for rho in (0.0, 0.3, 0.6):
        rs = [r for r in recs if r["rho"] == rho]
        summary["rho=%.1f" % rho] = dict(
            frames=len(rs),
            frames_all_cells_and_tries_identical=sum(r["cells_equal"] for r in rs),
            frames_winner_identical=sum(r["winner_indep"] == r["winner_oracle"] for r in rs),
            frames_same_refinement_rounds_and_inliers=sum(r["rounds_indep"] == r["rounds_oracle"] and
                                                          r["inliers_indep"] == r["inliers_oracle"] for r in rs),
            frames_pose_bits_equal=sum(r["pose_bits_equal"] for r in rs),
            max_dpose_cm=100 * max(r["dpose_m"] for r in rs), max_dpose_deg=max(r["dpose_deg"] for r in rs),
            dscore_clean_median=float(np.median([r["median_dscore_clean"] for r in rs])),
            dscore_clean_max=max(r["max_dscore_clean"] for r in rs),
            dscore_clean_frames_above_1e_6=sum(r["max_dscore_clean"] > 1e-6 for r in rs),
            duplicate_cell_hypotheses=sum(r["n_duplicate_cell_hyps"] for r in rs),
            dscore_duplicate_max=max(r["max_dscore_dup"] for r in rs),
            worst_gt_err_cm=100 * max(r["gt_err_m"] for r in rs), worst_gt_err_deg=max(r["gt_err_deg"] for r in rs))
    disagree = []
    for r in recs:
        c = cause(r)
        if c or r["dpose_m"] > 0.01 or r["dpose_deg"] > 0.1:
            disagree.append(dict(rho=r["rho"], image=r["image"], cause=c, dpose_cm=100 * r["dpose_m"],
                                 dpose_deg=r["dpose_deg"], winner=(r["winner_indep"], r["winner_oracle"])))
    out = dict(hypotheses=n_hyp, frames=len(recs), seconds=round(time.time() - t0, 1), workers=workers,
               summary=summary, disagreeing_frames=disagree)
    with open("/tmp/independent_pin_records.json", "w") as f:
        json.dump(recs, f)
    path = os.path.join(ROOT, "profiles", "r2_independent_pin.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(summary, indent=1))
    print("%d frames disagree somewhere; written to %s (%.0f s)" % (len(disagree), path, time.time() - t0))


if __name__ == "__main__":
    main()
