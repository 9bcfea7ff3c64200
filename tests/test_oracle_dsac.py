"""Pins for the CPU oracle of dsacstar.forward_rgb (oracle/dsac_oracle.c).

The reference ships no tests and its extension cannot be built here (OpenCV absent), so the
oracle is pinned by the analytic known answers SURVEY.md §8(c) lists.
"""
import math
import os
import subprocess
import sys

import numpy as np
import pytest

from crossloc_amd import synth

ARGS = dict(thr=10.0, alpha=100.0, max_reproj=100.0, sub=8)


def _run(O, scene, n_hyp=64, **kw):
    return O.forward_rgb(scene["coords"], n_hyp, ARGS["thr"], scene["focal"], scene["ppx"], scene["ppy"],
                         ARGS["alpha"], ARGS["max_reproj"], ARGS["sub"], debug=True, **kw)


def test_exp_sincos_accuracy(oracle):
    for x in np.linspace(-60.0, 20.0, 4001):
        assert abs(oracle.exp(x) - math.exp(x)) <= 4e-16 * math.exp(x)
    for x in np.linspace(-9.0, 9.0, 4001):
        s, c = oracle.sincos(x)
        assert abs(s - math.sin(x)) < 3e-16 and abs(c - math.cos(x)) < 3e-16
    assert oracle.exp(-800.0) == 0.0 and oracle.exp(800.0) == math.inf


def test_quartic_roots(oracle):
    rng = np.random.default_rng(7)
    assert np.allclose(sorted(oracle.quartic([1, -10, 35, -50, 24])), [1, 2, 3, 4], atol=1e-12)
    assert np.allclose(sorted(oracle.quartic([1, 0, -5, 0, 4])), [-2, -1, 1, 2], atol=1e-12)   # biquadratic
    assert len(oracle.quartic([1, 0, 1, 0, 1])) == 0                                          # no real root
    for _ in range(300):
        r = rng.uniform(-3, 3, size=4)
        A = np.poly(r) * rng.uniform(0.5, 2.0)
        got = np.sort(oracle.quartic(A))
        assert len(got) == 4 and np.allclose(got, np.sort(r), atol=1e-6)
        # two real + complex pair
        A = np.poly([r[0], r[1], complex(r[2], 1.0 + abs(r[3])), complex(r[2], -1.0 - abs(r[3]))]).real
        got = np.sort(oracle.quartic(A))
        assert len(got) == 2 and np.allclose(got, np.sort(r[:2]), atol=1e-6)


def test_rng_draws(oracle):
    seen = set()
    for t in range(200):
        d = oracle.draws(1305, 3, 5, t, 90, 60)
        assert (d[:, 0] >= 0).all() and (d[:, 0] < 90).all() and (d[:, 1] >= 0).all() and (d[:, 1] < 60).all()
        seen.add(tuple(d.ravel()))
    assert len(seen) == 200
    assert not np.array_equal(oracle.draws(1305, 3, 5, 0, 90, 60), oracle.draws(1305, 4, 5, 0, 90, 60))
    assert np.array_equal(oracle.draws(1305, 3, 5, 0, 90, 60), oracle.draws(1305, 3, 5, 0, 90, 60))
    xs = np.array([oracle.draws(1, 0, h, t, 90, 60)[:, 0] for h in range(40) for t in range(50)]).ravel()
    assert abs(xs.mean() - 44.5) < 1.5 and xs.min() == 0 and xs.max() == 89


def test_p3p_known_answer(oracle):
    rng = np.random.default_rng(11)
    ok = 0
    for _ in range(200):
        pose = synth.random_pose(rng)                      # cam->world
        Rwc = pose[:3, :3].T
        twc = -Rwc @ pose[:3, 3]
        uv = np.stack([rng.uniform(20, 700, 4), rng.uniform(20, 460, 4)], 1)
        depth = rng.uniform(150, 400, 4)
        Xc = np.stack([(uv[:, 0] - 360) / 480 * depth, (uv[:, 1] - 240) / 480 * depth, depth], 1)
        Xw = (pose[:3, :3] @ Xc.T).T + pose[:3, 3]
        res = oracle.p3p(Xw, uv, 480.0, 360.0, 240.0)
        assert res is not None
        R, t = res
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-12) and abs(np.linalg.det(R) - 1) < 1e-12
        if np.allclose(R, Rwc, atol=1e-6) and np.allclose(t, twc, rtol=1e-6, atol=1e-5):
            ok += 1
    assert ok >= 198          # the rare miss is a near-degenerate triangle, never a wrong-but-accepted pose
    # duplicate points (allowed by the sampler) must fail cleanly
    P = np.array([[0, 0, 5.0], [0, 0, 5.0], [1, 0, 5], [0, 1, 5]])
    assert oracle.p3p(P, [[1, 1], [1, 1], [2, 2], [3, 3]], 480, 360, 240) is None


def test_closed_form_scores(oracle):
    sc = synth.make_scene(5, noise=0.0, outlier_ratio=0.0)
    Rwc = sc["pose"][:3, :3].T
    twc = -Rwc @ sc["pose"][:3, 3]
    s = oracle.score(sc["coords"], Rwc, twc, 10.0, 100.0, 100.0, 480.0, 360.0, 240.0, 8)
    assert abs(s - 100.0 * (1 - 1 / (1 + math.exp(5.0)))) < 1e-3          # every cell at err ~ 0
    far = twc + np.array([5000.0, 0, 0])
    s = oracle.score(sc["coords"], Rwc, far, 10.0, 100.0, 100.0, 480.0, 360.0, 240.0, 8)
    assert s == pytest.approx(100.0 * (1 - 1 / (1 + math.exp(-45.0))), rel=1e-6)   # every cell clamped to 100 px


def test_gt_coordinates_recover_pose(oracle):
    """test_single_task.py:361 idea: feeding exact coordinates must give ~0 pose error."""
    for s in range(6):
        sc = synth.make_scene(100 + s, noise=0.0, outlier_ratio=0.0)
        pose, d = _run(oracle, sc, image=s)
        t_err, r_err = synth.pose_error(sc["pose"], pose)
        assert t_err < 1e-3 and r_err < 1e-3
        assert d["inliers"] == 5400 and (d["tries"] >= 1).all()


@pytest.mark.parametrize("rho", [0.0, 0.3, 0.6])
def test_noise_outlier_sweep(oracle, rho):
    errs = []
    for s in range(6):
        sc = synth.make_scene(2021 + s, noise=0.5, outlier_ratio=rho)
        pose, d = _run(oracle, sc, n_hyp=64, image=s)
        errs.append(synth.pose_error(sc["pose"], pose))
        assert 1 <= d["rounds"] <= 100
        assert d["inliers"] >= int(0.9 * (1 - rho) * 5400)
        assert np.isfinite(pose).all() and np.allclose(pose[3], [0, 0, 0, 1])
    t, r = np.array(errs).T
    assert np.median(t) < 0.25 and np.median(r) < 0.05


def test_winner_is_first_argmax_and_scores_bounded(oracle):
    sc = synth.make_scene(77, noise=0.5, outlier_ratio=0.3)
    _, d = _run(oracle, sc, n_hyp=64)
    assert d["winner"] == int(np.argmax(d["scores"]))
    assert (d["scores"] >= 0).all() and (d["scores"] <= 100.0).all()


def test_determinism_and_image_keying(oracle):
    sc = synth.make_scene(9, noise=0.5, outlier_ratio=0.3)
    p0, d0 = _run(oracle, sc, image=4)
    p1, d1 = _run(oracle, sc, image=4)
    p2, d2 = _run(oracle, sc, image=5)
    assert np.array_equal(p0, p1) and np.array_equal(d0["cells"], d1["cells"]) and np.array_equal(d0["scores"], d1["scores"])
    assert not np.array_equal(d0["cells"], d2["cells"])
    # first nHyp hypotheses do not depend on how many are requested (counter-based keying)
    _, d3 = _run(oracle, sc, n_hyp=16, image=4)
    assert np.array_equal(d3["cells"], d0["cells"][:16]) and np.array_equal(d3["scores"], d0["scores"][:16])


def test_thread_count_invariance(oracle):
    """The reference result depends on the OpenMP thread count (thread_rand.cpp:19-25); ours must not."""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import numpy as np\n"
            "from oracle import dsac_oracle as O\n"
            "from crossloc_amd import synth\n"
            "sc = synth.make_scene(31, noise=0.5, outlier_ratio=0.3)\n"
            "p, d = O.forward_rgb(sc['coords'], 32, 10.0, 480.0, 360.0, 240.0, 100.0, 100.0, 8, debug=True)\n"
            "sys.stdout.write(p.tobytes().hex() + d['cells'].tobytes().hex() + d['scores'].tobytes().hex())\n"
            % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    outs = []
    for nt in ("1", "4"):
        env = dict(os.environ, OMP_NUM_THREADS=nt)
        outs.append(subprocess.check_output([sys.executable, "-c", code], env=env))
    assert outs[0] == outs[1]


def test_strided_input_matches_contiguous(oracle):
    sc = synth.make_scene(13, noise=0.5, outlier_ratio=0.3)
    big = np.zeros((3, 64, 100), np.float32)
    big[:, 2:62, 5:95] = sc["coords"]
    view = big[:, 2:62, 5:95]
    assert not view.flags["C_CONTIGUOUS"]
    p0, d0 = _run(oracle, sc)
    p1, d1 = oracle.forward_rgb(view, 64, 10.0, 480.0, 360.0, 240.0, 100.0, 100.0, 8, debug=True)
    assert np.array_equal(p0, p1) and np.array_equal(d0["cells"], d1["cells"])


def test_degenerate_inputs_never_nan(oracle):
    """PnP failure handling (dsacstar.cpp:48, dsacstar_util.h:114-116): bounded retries, zero pose."""
    nodata = np.full((3, 60, 90), -1.0, np.float32)          # every point identical -> P3P always fails
    pose, d = oracle.forward_rgb(nodata, 8, 10.0, 480.0, 360.0, 240.0, 100.0, 100.0, 8, max_tries=50, debug=True)
    assert np.array_equal(pose, np.eye(4, dtype=np.float32))
    assert (d["tries"] == -50).all() and d["rounds"] == 0
    rng = np.random.default_rng(3)
    garbage = rng.uniform(-500, 500, size=(3, 60, 90)).astype(np.float32)
    pose, d = oracle.forward_rgb(garbage, 8, 10.0, 480.0, 360.0, 240.0, 100.0, 100.0, 8, max_tries=200, debug=True)
    assert np.isfinite(pose).all()


def test_small_and_ragged_grids(oracle):
    for Ho, Wo in [(8, 12), (7, 13), (60, 90)]:
        sc = synth.make_scene(50, noise=0.0, outlier_ratio=0.0, Ho=Ho, Wo=Wo)
        pose = oracle.forward_rgb(sc["coords"], 16, 10.0, 480.0, sc["ppx"], sc["ppy"], 100.0, 100.0, 8)
        t_err, r_err = synth.pose_error(sc["pose"], pose)
        assert t_err < 1e-2 and r_err < 1e-2


def test_refinement_agrees_with_scipy_least_squares(oracle):
    """Independent implementation of the non-linear PnP the refinement performs (dsacstar_util.h:570-580: minimise the
    summed squared reprojection error over the inlier set from the current pose): scipy's MINPACK Levenberg-Marquardt
    with scipy's own Rodrigues map.  Small noise and no outliers keep every cell an inlier in every round, so the
    refined pose must be THE minimiser over all 5400 cells — and the second round must stop on `count <= best`."""
    from scipy.optimize import least_squares
    from scipy.spatial.transform import Rotation
    for s in range(3):
        sc = synth.make_scene(300 + s, noise=0.05, outlier_ratio=0.0)
        pose, d = _run(oracle, sc, image=s)
        assert d["inliers"] == 5400 and d["rounds"] == 1
        Ho, Wo = sc["coords"].shape[1:]
        X = sc["coords"].reshape(3, -1).T.astype(np.float64)                       # row-major cells: index = y*Wo + x
        ys, xs = np.divmod(np.arange(Ho * Wo), Wo)
        px = np.stack([xs * 8 + 4, ys * 8 + 4], 1).astype(np.float64)
        f, cx, cy = sc["focal"], sc["ppx"], sc["ppy"]

        def resid(p):
            Xc = X @ Rotation.from_rotvec(p[:3]).as_matrix().T + p[3:]
            return np.concatenate([f * Xc[:, 0] / Xc[:, 2] + cx - px[:, 0], f * Xc[:, 1] / Xc[:, 2] + cy - px[:, 1]])

        R0, t0 = d["pose0"][:9].reshape(3, 3), d["pose0"][9:]                      # winning hypothesis (world -> camera)
        R1, t1 = d["pose1"][:9].reshape(3, 3), d["pose1"][9:]                      # after refinement
        p0 = np.concatenate([Rotation.from_matrix(R0).as_rotvec(), t0])
        sol = least_squares(resid, p0, method="lm", xtol=1e-15, ftol=1e-15, gtol=1e-15)
        Rs, ts = Rotation.from_rotvec(sol.x[:3]).as_matrix(), sol.x[3:]
        ang = np.linalg.norm(Rotation.from_matrix(Rs.T @ R1).as_rotvec())
        assert ang < 1e-7, ang                                                      # radians
        assert np.abs(ts - t1).max() < 1e-4, np.abs(ts - t1).max()                  # metres, |t| ~ 700 m
        cost1 = 0.5 * np.sum(resid(np.concatenate([Rotation.from_matrix(R1).as_rotvec(), t1])) ** 2)
        assert cost1 <= sol.cost * (1 + 1e-9) + 1e-12
        # the pose handed back is the inverse, as float32 (dsacstar_util.h:759-770)
        inv = np.eye(4); inv[:3, :3] = R1.T; inv[:3, 3] = -R1.T @ t1
        assert np.allclose(pose, inv.astype(np.float32), rtol=0, atol=1e-4)


def test_rotation_maps_agree_with_scipy(oracle):
    """cv::Rodrigues as restated in the oracle (matrix -> vector and the Jacobian of vector -> matrix) against scipy's
    rotation-vector map, including the small-angle and near-pi branches."""
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(9)
    for r in ([1e-5, -2e-5, 1e-5], [0.3, -0.2, 0.1], [3.0, 0.9, -0.2], list(np.array([0.6, 0.0, 0.8]) * (math.pi - 1e-3)),
              list(np.array([0.6, 0.0, 0.8]) * (math.pi - 1e-6))):
        R = Rotation.from_rotvec(r).as_matrix()
        got = oracle.log_so3(R)                                                     # matrix -> vector (backward's inverse map)
        # within 1e-5 of pi the restatement, like cv::Rodrigues, takes the axis from the diagonal and drops the O(pi-angle)
        # antisymmetric part: exact to ~1e-6 there, to rounding elsewhere
        near_pi = abs(np.linalg.norm(r) - math.pi) < 1e-5
        assert np.allclose(Rotation.from_rotvec(got).as_matrix(), R, atol=2e-6 if near_pi else 1e-9)
        J = oracle.rodrigues_jac(np.asarray(r, np.float64))                         # dR/dr, 9x3
        eps = 1e-6
        for k in range(3):
            dr = np.zeros(3); dr[k] = eps
            num = (Rotation.from_rotvec(np.add(r, dr)).as_matrix() - Rotation.from_rotvec(np.subtract(r, dr)).as_matrix()) / (2 * eps)
            assert np.allclose(np.asarray(J).reshape(9, 3)[:, k], num.reshape(9), atol=2e-6), (r, k)
