#!/usr/bin/env python
"""Independent pin of `backward_rgb`: oracle/dsac_bwd_oracle.c against tests/indep_dsac.backward_rgb
- the reference's chain of Jacobians (dsacstar.cpp:200-483, dsacstar_derivative.h, dsacstar_loss.h) with every analytic
Jacobian replaced by central differences of the independent forward pieces (numpy / scipy), MINPACK refinement, numpy
pseudo-inverse - over FRAMES synthetic frames x HYPS hypotheses at outlier ratios 0 / 0.3 / 0.6.  CPU only.
Writes profiles/r3_independent_bwd_pin.json.   python tools/independent_bwd_pin.py [frames=32] [hyps=32]"""
import json
import os
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np                                       # noqa: E402
from scipy.spatial.transform import Rotation             # noqa: E402

import indep_dsac as ind                                 # noqa: E402
from crossloc_amd import synth                           # noqa: E402
from oracle import dsac_oracle as xo                     # noqa: E402

warnings.filterwarnings("ignore")
ARGS = (10.0, 480.0, 360.0, 240.0, 1.0, 100.0, 100.0, 100.0, 100.0, 8)    # thr f cx cy wRot wTrans softClamp alpha maxReproj sub


def one(seed, rho, n_hyp):
    sc = synth.make_scene(seed, noise=0.5, outlier_ratio=rho)
    coords, gt = sc["coords"], sc["pose"]
    grad = np.zeros_like(coords)
    Eo, rec = xo.backward_rgb(coords, grad, gt, n_hyp, *ARGS, 1305 + seed, image=seed, debug=True)
    g0 = grad.astype(np.float64)
    refined = {h: (Rotation.from_matrix(rec[h, 6:15].reshape(3, 3)).as_rotvec(), rec[h, 15:18])
               for h in range(n_hyp) if rec[h, 2] > 0}
    out = dict(seed=seed, outlier_ratio=rho, expected_loss_oracle=Eo, active=int((rec[:, 2] > 0).sum()),
               grad_max=float(np.abs(g0).max()))
    for tag, mode, ref in (("chain_on_oracle_poses_oracle_gt", "oracle", refined),
                           ("chain_on_oracle_poses_reference_gt", "reference", refined),
                           ("fully_independent_reference_gt", "reference", None)):
        ind.GT_MODE = mode
        E, g, info = ind.backward_rgb(coords, gt, n_hyp, *ARGS, seed=1305 + seed, image=seed, refined=ref)
        den = max(float(np.abs(g0).max()), 1e-30)
        out[tag] = dict(dE_rel=abs(E - Eo) / max(abs(Eo), 1e-30), grad_max_rel=float(np.abs(g - g0).max() / den),
                        grad_l2_rel=float(np.linalg.norm(g - g0) / max(np.linalg.norm(g0), 1e-30)),
                        active=len(info["active"]))
    ind.GT_MODE = "reference"
    return out


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    n_hyp = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    t0 = time.time()
    recs = []
    for i in range(frames):
        rho = (0.0, 0.3, 0.6)[i % 3]
        recs.append(one(7000 + i, rho, n_hyp))
        sys.stderr.write("frame %d/%d  %.0fs\n" % (i + 1, frames, time.time() - t0))
    summ = {}
    for tag in ("chain_on_oracle_poses_oracle_gt", "chain_on_oracle_poses_reference_gt", "fully_independent_reference_gt"):
        for k in ("dE_rel", "grad_max_rel", "grad_l2_rel"):
            v = np.array([r[tag][k] for r in recs])
            summ["%s.%s" % (tag, k)] = dict(median=float(np.median(v)), max=float(v.max()))
    out = dict(what="oracle/dsac_bwd_oracle.c vs tests/indep_dsac.backward_rgb (numeric Jacobians, MINPACK, numpy pinv)",
               frames=frames, hypotheses=n_hyp, args=dict(zip("thr f cx cy wRot wTrans softClamp alpha maxReproj sub".split(), ARGS)),
               summary=summ, frames_detail=recs, seconds=round(time.time() - t0, 1))
    path = os.path.join(ROOT, "profiles", "r3_independent_bwd_pin.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(summ, indent=1))


if __name__ == "__main__":
    main()
