"""A second, deliberately DIFFERENT restatement of `dsacstar.forward_rgb` in numpy / scipy — TEST INFRASTRUCTURE ONLY.

Purpose ("independent pin for the solver oracle"): oracle/dsac_oracle.c was written together with the
HIP kernels (wave-shaped reductions, polynomial exp/sincos, Ferrari quartic, frame-based alignment, Cholesky LM on a
left-multiplicative increment).  This file follows the REFERENCE's control flow and data types instead
(/root/reference/dsacstar/dsacstar.cpp:63-178, dsacstar_util.h) and gets every piece of third-party arithmetic from an
unrelated implementation:

  piece                         reference                              here                         oracle / HIP
  ----------------------------  -------------------------------------  ---------------------------  -----------------------------
  minimal solver                cv::solvePnP(P3P) util.h:104-112,185   law-of-cosines system        closed-form Grunert
                                                                       eliminated numerically       coefficients + Ferrari +
                                                                       (np.polymul resultant),      Newton polish + triangle
                                                                       np.roots (companion eig),    frames
                                                                       Kabsch/SVD alignment
  pose parametrisation          Rodrigues vector (types.h:42-44)       scipy Rotation rotvec        rotation matrix
  projection                    cv::projectPoints util.h:199,395       same formula, numpy          same formula, scalar
  soft-inlier score             serial x-major sum, std::exp :324-340  x-major np.cumsum, np.exp    64 lane partials + butterfly,
                                                                                                    polynomial exp
  selection                     softMax + draw(argmax) :684-752        the same two functions       arg-max of the scores
  refinement optimiser          cv::solvePnP(ITERATIVE, guess) :570    MINPACK lmder through        hand-written CvLevMarq state
                                                                       scipy least_squares('lm')    machine, 6x6 Cholesky
                                                                       on (rotvec, t)
  RNG                           mt19937 per thread (not reproducible)  the build's counter-based    same spec
                                                                       splitmix64 spec, restated
                                                                       with Python integers

Only the RNG *specification* (key = seed, image, hypothesis, try; draw x then y) is shared, because
"the same sampled cells" is only definable on a shared stream.
"""
import math

import numpy as np
from scipy.optimize import fsolve, least_squares
from scipy.spatial.transform import Rotation

M64 = (1 << 64) - 1
GOLDEN = 0x9E3779B97F4A7C15
MAX_REF_STEPS = 100                 # dsacstar.cpp:47
MAX_HYPOTHESES_TRIES = 1000000      # dsacstar.cpp:48
EPS = 1e-8                          # dsacstar_types.h: probabilities below it are skipped by draw()
POLISH_DISTANCES = True             # see p3p_plus_one


# ------------------------------------------------------------------------------------------ RNG (shared specification)

def _mix64(z):
    z &= M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def draw_cells(seed, image, hyp, t, Wo, Ho):
    """The 4 (x, y) cells of try `t` of hypothesis `hyp`: x first, then y (dsacstar_util.h:171-172)."""
    s = _mix64(seed + GOLDEN * (image + 1))
    s = _mix64(s ^ ((hyp << 32) | t))
    out = []
    for j in range(4):
        rx = _mix64(s + GOLDEN * (2 * j + 1)) >> 32
        ry = _mix64(s + GOLDEN * (2 * j + 2)) >> 32
        out.append(((rx * Wo) >> 32, (ry * Ho) >> 32))
    return out


# ------------------------------------------------------------------------------------------ geometry

def project(rvec, tvec, X, f, cx, cy):
    """cv::projectPoints without distortion: X [n,3] float64 -> pixels [n,2] as float32 (the reference stores Point2f).
    No cheirality test; only z == 0 is guarded."""
    R = Rotation.from_rotvec(rvec).as_matrix()
    Xc = X @ R.T + tvec
    z = np.where(Xc[:, 2] != 0.0, 1.0 / np.where(Xc[:, 2] != 0.0, Xc[:, 2], 1.0), 1.0)
    u = Xc[:, 0] * z * f + cx
    v = Xc[:, 1] * z * f + cy
    return np.stack([u, v], 1).astype(np.float32)


def _norm2f(d):
    """cv::norm(Point2f): the float differences squared and summed in double."""
    d = d.astype(np.float64)
    return np.sqrt(d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1])


def _kabsch(Pw, Pc):
    """Rigid (R, t) with Pc ~ R Pw + t (Arun et al. 1987, SVD)."""
    mw, mc = Pw.mean(0), Pc.mean(0)
    H = (Pw - mw).T @ (Pc - mc)
    U, S, Vt = np.linalg.svd(H)
    d = np.sign(np.linalg.det(Vt.T @ U.T))
    R = Vt.T @ np.diag([1.0, 1.0, d]) @ U.T
    return R, mc - R @ mw


def p3p_plus_one(P, uv, f, cx, cy, all_candidates=None):
    """cv::solvePnP(SOLVEPNP_P3P) call-site contract: the first three correspondences give up to four poses, the
    fourth picks the one with the smallest reprojection error.  Returns (rvec, tvec) or None.
    all_candidates: optional list that receives (4th-point squared error, R, t) of every admissible solution."""
    m = np.stack([(uv[:3, 0] - cx) / f, (uv[:3, 1] - cy) / f, np.ones(3)], 1)
    fb = m / np.linalg.norm(m, axis=1, keepdims=True)
    ca, cb, cg = fb[1] @ fb[2], fb[0] @ fb[2], fb[0] @ fb[1]
    a2 = np.sum((P[1] - P[2]) ** 2)
    b2 = np.sum((P[0] - P[2]) ** 2)
    c2 = np.sum((P[0] - P[1]) ** 2)
    if not (a2 > 0 and b2 > 0 and c2 > 0):
        return None
    if np.linalg.norm(np.cross(P[1] - P[0], P[2] - P[0])) == 0.0:
        return None
    # distances s1, s2 = u s1, s3 = v s1 (law of cosines).  Eliminating s1^2 leaves two quadratics in u with polynomial
    # coefficients in v (highest power first):
    #   b2 u^2 - 2 b2 ca v u + (b2 v^2 - a2 w(v)) = 0,   b2 u^2 - 2 b2 cg u + (b2 - c2 w(v)) = 0,   w = v^2 - 2 cb v + 1
    w = np.array([1.0, -2.0 * cb, 1.0])
    B1 = np.array([-2.0 * b2 * ca, 0.0])
    C1 = np.array([b2, 0.0, 0.0]) - a2 * w
    B2 = np.array([-2.0 * b2 * cg])
    C2 = np.array([0.0, 0.0, b2]) - c2 * w
    dB = np.polysub(B1, B2)                       # their difference is linear in u:  dB u + dC = 0
    dC = np.polysub(C1, C2)
    quartic = np.polyadd(np.polysub(b2 * np.polymul(dC, dC), np.polymul(np.polymul(B2, dC), dB)),
                         np.polymul(C2, np.polymul(dB, dB)))
    if not np.all(np.isfinite(quartic)) or quartic[0] == 0.0:
        return None
    best, best_e = None, None
    for root in np.roots(quartic):
        if abs(root.imag) > 1e-9 * max(1.0, abs(root.real)):
            continue
        v = root.real
        den = np.polyval(dB, v)
        if not v > 0.0 or den == 0.0:
            continue
        u = -np.polyval(dC, v) / den
        wv = np.polyval(w, v)
        if not (u > 0.0 and wv > 0.0):
            continue
        s = np.array([1.0, u, v]) * math.sqrt(b2 / wv)
        if POLISH_DISTANCES:
            # the quartic's roots carry ~1e-10 relative error (companion-matrix eigenvalues) which an ill-conditioned
            # triangle amplifies to ~1e-6 in the pose: MINPACK hybrd on the three law-of-cosines equations removes it
            s = fsolve(lambda q: [q[1] * q[1] + q[2] * q[2] - 2 * q[1] * q[2] * ca - a2,
                                  q[0] * q[0] + q[2] * q[2] - 2 * q[0] * q[2] * cb - b2,
                                  q[0] * q[0] + q[1] * q[1] - 2 * q[0] * q[1] * cg - c2], s, xtol=1e-15)
            if not np.all(s > 0.0):
                continue
        Pc = s[:, None] * fb
        R, t = _kabsch(P[:3], Pc)
        Xc = R @ P[3] + t
        e = (cx + f * Xc[0] / Xc[2] - uv[3, 0]) ** 2 + (cy + f * Xc[1] / Xc[2] - uv[3, 1]) ** 2
        if all_candidates is not None:
            all_candidates.append((float(e), R, t))
        if best is None or e < best_e:
            best, best_e = (R, t), e
    if best is None:
        return None
    return Rotation.from_matrix(best[0]).as_rotvec(), best[1]


# ------------------------------------------------------------------------------------------ pipeline stages

class Frame:
    """Scene coordinates [3,Ho,Wo] float32 with the two cell orders the reference uses."""

    def __init__(self, coords, sub):
        self.c = np.asarray(coords, np.float32)
        _, self.Ho, self.Wo = self.c.shape
        self.sub = sub
        # x-major order of getReproErrs / refineHyp (x outer, y inner; dsacstar_util.h:375-376, 547-548)
        xs, ys = np.meshgrid(np.arange(self.Wo), np.arange(self.Ho), indexing="ij")
        self.xs, self.ys = xs.ravel(), ys.ravel()
        self.X = self.c[:, self.ys, self.xs].T.astype(np.float64)                  # [N,3], x-major
        # createSampling (dsacstar_util.h:59-76): integer pixel centres
        self.pix = np.stack([self.xs * sub + sub // 2, self.ys * sub + sub // 2], 1).astype(np.float32)

    def errors(self, hyp, f, cx, cy, max_reproj):
        """getReproErrs with calcJ=false (dsacstar_util.h:356-446), x-major float32 vector."""
        proj = project(hyp[0], hyp[1], self.X, f, cx, cy)
        n = _norm2f(self.pix - proj).astype(np.float32)
        return np.minimum(n, np.float32(max_reproj))


def sample_hypothesis(fr, seed, image, h, thr, f, cx, cy, max_tries):
    """One iteration of the `h` loop of sampleHypotheses (dsacstar_util.h:157-220)."""
    hyp = (np.zeros(3), np.zeros(3))
    cells = None
    for t in range(max_tries):
        cells = draw_cells(seed, image, h, t, fr.Wo, fr.Ho)
        P = np.array([[fr.c[0, y, x], fr.c[1, y, x], fr.c[2, y, x]] for x, y in cells], np.float64)
        uv32 = np.array([[x * fr.sub + fr.sub // 2, y * fr.sub + fr.sub // 2] for x, y in cells], np.float32)
        sol = p3p_plus_one(P, uv32.astype(np.float64), f, cx, cy)
        if sol is None:
            hyp = (np.zeros(3), np.zeros(3))              # safeSolvePnP zeroes both on failure (:114-116)
            continue
        hyp = sol
        proj = project(hyp[0], hyp[1], P, f, cx, cy)
        if np.all(_norm2f(uv32 - proj) < thr):            # :208-219
            return hyp, cells, t + 1
    return hyp, cells, -max_tries


def hyp_score(errs, thr, alpha, Wo, Ho):
    """getHypScores (dsacstar_util.h:316-343): float beta and soft threshold argument, double logistic, SERIAL x-major
    accumulation (np.cumsum adds left to right), then `*= alpha / cols / rows` in float."""
    beta = np.float32(5.0) / np.float32(thr)
    st = (beta * (errs - np.float32(thr))).astype(np.float64)
    st = 1.0 / (1.0 + np.exp(-st))
    total = np.cumsum(1.0 - st)[-1]
    fac = np.float32(alpha) / np.float32(Wo) / np.float32(Ho)
    return float(total * np.float64(fac))


def soft_max(scores):
    """dsacstar_util.h:684-704"""
    sf = np.exp(np.asarray(scores) - np.max(scores))
    return sf / np.cumsum(sf)[-1]


def draw_argmax(probs):
    """draw(probs, training=false), dsacstar_util.h:727-752: first maximum among probs >= EPS."""
    max_prob, max_idx = -1.0, 0
    for i, p in enumerate(probs):
        if p < EPS:
            continue
        if max_prob < 0 or p > max_prob:
            max_prob, max_idx = p, i
    return max_idx


def refine(fr, hyp, errs, thr, f, cx, cy, max_reproj):
    """refineHyp (dsacstar_util.h:522-597).  Returns (hyp, rounds accepted, final inlier count)."""
    best, rounds, final = 4, 0, 0
    refine.last_inliers = None                           # inlierMap of the last accepted re-fit (dsacstar_util.h:583)
    for _ in range(MAX_REF_STEPS):
        inl = errs < np.float32(thr)
        n = int(inl.sum())
        if n <= best:
            break
        best = n
        X, pix = fr.X[inl], fr.pix[inl].astype(np.float64)

        def resid(p):
            R = Rotation.from_rotvec(p[:3]).as_matrix()
            Xc = X @ R.T + p[3:]
            return np.concatenate([Xc[:, 0] / Xc[:, 2] * f + cx - pix[:, 0], Xc[:, 1] / Xc[:, 2] * f + cy - pix[:, 1]])
        sol = least_squares(resid, np.concatenate(hyp), method="lm", xtol=1e-14, ftol=1e-14, gtol=1e-14, max_nfev=2000)
        if not np.all(np.isfinite(sol.x)):
            break
        hyp = (sol.x[:3], sol.x[3:])
        rounds, final = rounds + 1, n
        refine.last_inliers = inl.copy()
        errs = fr.errors(hyp, f, cx, cy, max_reproj)
    return hyp, rounds, final


def pose2trans(hyp):
    """dsacstar_util.h:759-770: [R t; 0 1]^-1 through a general inverse, stored as float32 (dsacstar.cpp:172-177)."""
    T = np.eye(4)
    T[:3, :3] = Rotation.from_rotvec(hyp[0]).as_matrix()
    T[:3, 3] = hyp[1]
    return np.linalg.inv(T).astype(np.float32)


def forward_rgb(coords, n_hyp, thr, focal, ppx, ppy, alpha, max_reproj, sub, seed=1305, image=0,
                max_tries=MAX_HYPOTHESES_TRIES):
    """dsacstar_rgb_forward (dsacstar.cpp:63-178).  Returns (out_pose [4,4] float32, info dict)."""
    fr = Frame(coords, sub)
    f, cx, cy = float(np.float32(focal)), float(np.float32(ppx)), float(np.float32(ppy))
    hyps, cells, tries, scores = [], [], [], []
    for h in range(n_hyp):
        hyp, c4, t = sample_hypothesis(fr, seed, image, h, thr, f, cx, cy, max_tries)
        hyps.append(hyp)
        cells.append([y * fr.Wo + x for x, y in c4])
        tries.append(t)
        scores.append(hyp_score(fr.errors(hyp, f, cx, cy, max_reproj), thr, alpha, fr.Wo, fr.Ho))
    win = draw_argmax(soft_max(scores))
    hyp0 = hyps[win]
    hyp, rounds, inliers = refine(fr, hyp0, fr.errors(hyp0, f, cx, cy, max_reproj), thr, f, cx, cy, max_reproj)
    return pose2trans(hyp), dict(cells=np.array(cells, np.int32), tries=np.array(tries, np.int32),
                                 scores=np.array(scores), winner=win, rounds=rounds, inliers=inliers,
                                 pose0=pose2trans(hyp0))


# ------------------------------------------------------------------------------------------ backward_rgb, independently
#
# dsacstar_rgb_backward (dsacstar.cpp:200-483) defines the gradient of the expected pose loss through a chain of Jacobians
# that the reference evaluates ANALYTICALLY (cv::projectPoints' Jacobian, dProjectdObj dsacstar_derivative.h:50-107, dLoss
# dsacstar_loss.h:86-212, the soft-max / soft-inlier derivatives :204-262, 340-352) plus one it evaluates numerically itself
# (dPNP :140-200, central differences of the minimal solver).  Here the CHAIN is the reference's - same paths, same clamps,
# same early-outs - but every analytic Jacobian is replaced by central differences of the independent forward pieces above
# (project / scipy rotations / numpy norms), the pseudo-inverse is numpy's SVD one, the refinement is MINPACK, the minimal
# solver is p3p_plus_one.  What oracle/dsac_bwd_oracle.c (and the HIP kernels, bit-identical to it) must reproduce.

PROB_THRESH = 0.001                 # dsacstar_derivative.h:36
MAXLOSS = 10000000.0                # dsacstar_loss.h:35
EPS_R = 1e-8                        # dsacstar_util.h:45 (EPS of the C++ side)


def _cdiff(fn, x, steps):
    """Central differences of fn: R^n -> R^m; returns [m, n]."""
    x = np.asarray(x, np.float64)
    cols = []
    for i, h in enumerate(steps):
        e = np.zeros_like(x)
        e[i] = h
        cols.append((np.asarray(fn(x + e), np.float64) - np.asarray(fn(x - e), np.float64)) / (2.0 * h))
    return np.stack(cols, -1)


def _project64(rvec, tvec, X, f, cx, cy):
    """The projection in float64 throughout (project() rounds to float32 like cv::projectPoints' output - not differentiable
    numerically)."""
    Xc = X @ Rotation.from_rotvec(rvec).as_matrix().T + tvec
    return np.stack([Xc[:, 0] / Xc[:, 2] * f + cx, Xc[:, 1] / Xc[:, 2] * f + cy], 1)


def _proj_jac(hyp, X, f, cx, cy):
    """d proj / d (rvec, tvec) for all points, numerically: [N, 2, 6]."""
    def fn(p):
        return _project64(p[:3], p[3:], X, f, cx, cy)
    return _cdiff(fn, np.concatenate(hyp), [1e-6] * 3 + [1e-5 * max(1.0, float(np.abs(hyp[1]).max()))] * 3)


def _resid_rows(hyp, X, pix, f, cx, cy, max_reproj):
    """Rows of `jacobeanR` / `jacobeanHyp` (dsacstar.cpp:389-405, dsacstar_util.h:420-434): d |proj - pix| / d pose with
    the float32 projections the reference holds (cv::Point2f), zero above max_reproj."""
    proj32 = project(hyp[0], hyp[1], X, f, cx, cy).astype(np.float32)
    d = proj32.astype(np.float64) - pix
    err = np.maximum(np.sqrt((d * d).sum(1)), EPS_R)
    J = _proj_jac(hyp, X, f, cx, cy)                                     # [N,2,6]
    rows = (d[:, :, None] * J).sum(1) / err[:, None]
    rows[err > max_reproj] = 0.0
    return rows


def _dproject_dobj(hyp, X, pix, f, cx, cy, max_reproj):
    """dProjectdObj for every point: d |proj(X) - pix| / dX by central differences, [N,3]; zero where the point is (nearly)
    in the camera plane or the error is above max_reproj (dsacstar_derivative.h:69-85)."""
    R = Rotation.from_rotvec(hyp[0]).as_matrix()

    def err_of(Xp):
        Xc = Xp @ R.T + hyp[1]
        px, py = f * Xc[:, 0] / Xc[:, 2] + cx, f * Xc[:, 1] / Xc[:, 2] + cy
        return np.sqrt((pix[:, 0] - px) ** 2 + (pix[:, 1] - py) ** 2)
    Xc = X @ R.T + hyp[1]
    out = np.zeros((X.shape[0], 3))
    scale = max(1.0, float(np.abs(X).max()))
    h = 1e-7 * scale
    for k in range(3):
        e = np.zeros(3)
        e[k] = h
        out[:, k] = (err_of(X + e) - err_of(X - e)) / (2 * h)
    bad = (np.abs(Xc[:, 2]) < EPS_R) | (err_of(X) > max_reproj)
    out[bad] = 0.0
    return out


def pose_loss(hyp, gt_pose, w_rot, w_trans, cut):
    """loss() of dsacstar_loss.h:68-85 on est = pose2trans(hyp) (float64) and the ground-truth cam->world matrix."""
    T = np.eye(4)
    T[:3, :3] = Rotation.from_rotvec(hyp[0]).as_matrix()
    T[:3, 3] = hyp[1]
    est = np.linalg.inv(T)
    gt = np.asarray(gt_pose, np.float64)
    tr = np.trace(gt[:3, :3] @ est[:3, :3].T)
    rot_err = 180.0 * math.acos((min(3.0, max(-1.0, tr)) - 1.0) / 2.0) / math.pi
    t_err = float(np.linalg.norm(est[:3, 3] - gt[:3, 3]))
    loss = w_rot * rot_err + w_trans * t_err
    if loss > cut:
        loss = math.sqrt(cut * loss)
    return min(loss, MAXLOSS)


GT_MODE = "reference"               # see _dloss


def _dloss(hyp, gt_pose, w_rot, w_trans, cut):
    """dLoss (dsacstar_loss.h:86-212), numerically: the gradient of w_rot * angle + w_trans * distance of the inverted
    poses w.r.t. (rvec, tvec); above `cut` the reference scales it by 0.5 / sqrt(raw) (:104-108, 201-202 - NOT the
    derivative of sqrt(cut * raw): its own definition, reproduced); zero above MAXLOSS, for a perfect estimate, and when
    not finite."""
    # hypGT = trans2pose(gtTrans) (dsacstar_util.h:777-790): the inverse of the float ground-truth matrix, its rotation
    # through cv::Rodrigues to a vector and back - i.e. ORTHONORMALISED (the float matrix is not, at the 1e-7 level, which
    # matters next to angles of 1e-4 rad); loss() above uses the raw matrix, dLoss the pose: reproduced
    # GT_MODE "reference": exactly that - t2 from the GENERAL inverse of the float matrix, so invT2 = R2^T t2 is the camera
    # centre moved by ~|C| * 1e-7 (5e-5 m at 500 m).  GT_MODE "oracle": oracle/dsac_bwd_oracle.c's documented deviation, a
    # rigid inverse (t2 = -R2 C, invT2 = -C exactly, consistent with loss()).  Next to a translation error of centimetres
    # the two differ by ~1e-3 in dLoss, ~1e-4 in the final gradient.
    g64 = np.asarray(gt_pose, np.float64)
    ginv = np.linalg.inv(g64)
    Rg = Rotation.from_matrix(ginv[:3, :3]).as_matrix()                # world -> camera rotation of the ground truth
    Cg = -(Rg.T @ ginv[:3, 3]) if GT_MODE == "reference" else g64[:3, 3].copy()      # invT2 = R2^T t2 = -Cg

    def raw(p):
        R = Rotation.from_rotvec(p[:3]).as_matrix()
        tr = np.trace(R @ Rg.T)
        rot = 180.0 * math.acos((min(3.0, max(-1.0, tr)) - 1.0) / 2.0) / math.pi
        dist = float(np.linalg.norm(R.T @ p[3:] + Cg))                # |invT1 - invT2|, invT = R^T t = -(camera centre)
        return np.array([w_rot * rot + w_trans * dist]), rot, dist
    p0 = np.concatenate(hyp)
    val, rot_err, t_err = raw(p0)
    loss = float(val[0])
    cut_loss = loss > cut
    if cut_loss:
        loss = math.sqrt(loss)
    if loss > MAXLOSS or (t_err + rot_err) < EPS_R:
        return np.zeros(6)
    g = _cdiff(lambda p: raw(p)[0], p0, [1e-7] * 3 + [1e-6 * max(1.0, float(np.abs(p0[3:]).max()))] * 3)[0]
    if cut_loss:
        g = g * (0.5 / loss)
    return g if np.all(np.isfinite(g)) else np.zeros(6)


def _dpnp(P4, uv4, f, cx, cy):
    """dPNP (dsacstar_derivative.h:140-200) for a minimal set: central differences of the P3P pose w.r.t. the first three
    object points with the reference's float32 step sequence (+= eps, -= 2 eps, += eps on cv::Point3f); 6 x 12."""
    eps = np.float32(0.001)
    two_eps = float(np.float32(2) * eps)
    obj = np.asarray(P4, np.float32).copy()
    J = np.zeros((6, 12))
    for i in range(3):
        for j in range(3):
            obj[i, j] = obj[i, j] + eps
            fs = p3p_plus_one(obj.astype(np.float64), uv4, f, cx, cy)
            if fs is None:
                return np.zeros((6, 12))
            obj[i, j] = obj[i, j] - np.float32(2) * eps
            bs = p3p_plus_one(obj.astype(np.float64), uv4, f, cx, cy)
            if bs is None:
                return np.zeros((6, 12))
            obj[i, j] = obj[i, j] + eps
            col = np.concatenate([(fs[0] - bs[0]) / two_eps, (fs[1] - bs[1]) / two_eps])
            if not np.all(np.isfinite(col)):
                return np.zeros((6, 12))
            J[:, i * 3 + j] = col
    return J


def backward_rgb(coords, gt_pose, n_hyp, thr, focal, ppx, ppy, w_rot, w_trans, soft_clamp, alpha, max_reproj, sub, seed,
                 image=0, max_tries=MAX_HYPOTHESES_TRIES, refined=None):
    """dsacstar_rgb_backward (dsacstar.cpp:200-483).  Returns (expected loss, gradient [3,Ho,Wo] float64 - the values the
    reference ADDS to its float gradient tensor -, info).
    `refined` {h: (rvec, tvec)}: use these refined poses instead of MINPACK's.  Where an iterative solver stops is not
    part of the algorithm's definition (cv::solvePnP: 20 iterations or a relative step below FLT_EPSILON; MINPACK: 1e-14):
    two solvers agree on the optimum to ~0.05 mm at |X| ~ 500 m, which the centimetre-scale pose loss shows as ~1e-3
    relative - so the chain of Jacobians is compared on the SAME refined poses, and the fully independent run is reported
    next to it."""
    gt_pose = np.asarray(gt_pose, np.float32).astype(np.float64)      # the reference receives a float tensor (:244-249)
    fr = Frame(coords, sub)
    f, cx, cy = float(np.float32(focal)), float(np.float32(ppx)), float(np.float32(ppy))
    N = fr.Wo * fr.Ho
    pix = fr.pix.astype(np.float64)
    hyps, cells, scores, errs_all = [], [], [], []
    for h in range(n_hyp):
        hyp, c4, _ = sample_hypothesis(fr, seed, image, h, thr, f, cx, cy, max_tries)
        hyps.append(hyp)
        cells.append(c4)
        e = fr.errors(hyp, f, cx, cy, max_reproj)
        errs_all.append(e)
        scores.append(hyp_score(e, thr, alpha, fr.Wo, fr.Ho))
    probs = soft_max(scores)
    ref_hyps, inl_maps = [], []
    for h in range(n_hyp):
        if probs[h] < PROB_THRESH:
            ref_hyps.append(hyps[h]); inl_maps.append(None)
            continue
        rh, _, _ = refine(fr, hyps[h], errs_all[h], thr, f, cx, cy, max_reproj)
        if refined is not None and h in refined and refine.last_inliers is not None:
            rh = (np.asarray(refined[h][0], np.float64), np.asarray(refined[h][1], np.float64))
        ref_hyps.append(rh)
        inl_maps.append(refine.last_inliers)
    losses = np.array([pose_loss(ref_hyps[h], gt_pose, w_rot, w_trans, soft_clamp) for h in range(n_hyp)])
    expected = float(np.cumsum(probs * losses)[-1])
    grad = np.zeros((N, 3))                                              # x-major cell order, like fr.X
    beta = float(np.float32(5.0) / np.float32(thr))
    active = [h for h in range(n_hyp) if probs[h] >= PROB_THRESH]
    for h in active:
        # ---- path I: through the refined pose (dsacstar.cpp:353-445)
        g1 = np.zeros((N, 3))
        inl = inl_maps[h]
        if inl is not None and int(inl.sum()) >= 4:
            X, px = fr.X[inl], pix[inl]
            JR = _resid_rows(ref_hyps[h], X, px, f, cx, cy, max_reproj)             # [n,6]
            JR = -np.linalg.pinv(JR.T @ JR) @ JR.T                                   # 6 x n
            if np.abs(JR).max() > 10:
                JR[:] = 0.0
            dNdO = _dproject_dobj(ref_hyps[h], X, px, f, cx, cy, max_reproj)        # [n,3]
            dL = _dloss(ref_hyps[h], gt_pose, w_rot, w_trans, soft_clamp)            # [6]
            g1[inl] = (dL @ JR)[:, None] * dNdO
        # ---- path II: through the score (dsacstar_derivative.h:204-352)
        sog = probs[h] * losses[h] - probs[h] * float(np.cumsum(probs * losses)[-1])
        st = 1.0 / (1.0 + np.exp(-(np.float32(beta) * (errs_all[h] - np.float32(thr))).astype(np.float64)))
        dRe = -st * (1.0 - st) * beta * sog * (float(np.float32(alpha)) / fr.Wo / fr.Ho)
        dPdO = _dproject_dobj(hyps[h], fr.X, pix, f, cx, cy, max_reproj)
        g2 = dPdO * dRe[:, None]
        c4 = cells[h]
        P4 = np.array([[fr.c[0, y, x], fr.c[1, y, x], fr.c[2, y, x]] for x, y in c4], np.float32)
        uv4 = np.array([[x * sub + sub // 2, y * sub + sub // 2] for x, y in c4], np.float64)
        dHdO = _dpnp(P4, uv4, f, cx, cy)
        if np.abs(dHdO).max() > 10:
            dHdO[:] = 0.0
        dPdH = _resid_rows(hyps[h], fr.X, pix, f, cx, cy, max_reproj)               # [N,6]
        support = (dRe[:, None] * dPdH).sum(0) @ dHdO                                # [12]
        for i, (x, y) in enumerate(c4):
            g2[x * fr.Ho + y] += support[3 * i:3 * i + 3]
        grad += probs[h] * g1 + g2
    out = np.zeros((3, fr.Ho, fr.Wo))
    out[:, fr.ys, fr.xs] = grad.T
    return expected, out, dict(probs=probs, losses=losses, active=active, scores=np.array(scores),
                               inliers=[None if m is None else int(m.sum()) for m in inl_maps])


# ------------------------------------------------------------------------------------------ comparison with the oracle

def compare_with_oracle(coords, n_hyp, image, gt_pose=None, thr=10.0, focal=480.0, ppx=360.0, ppy=240.0, alpha=100.0,
                        max_reproj=100.0, sub=8):
    """Runs this restatement and oracle/dsac_oracle.c on one frame and returns a flat record of agreements.
    Duplicate-cell hypotheses (the 4th draw repeats one of the first three: allowed, dsacstar_util.h:168-183) are
    reported separately: the 4th point then cannot disambiguate the P3P solutions and the choice among them is decided
    by rounding in ANY implementation, OpenCV's included."""
    from oracle import dsac_oracle as xo
    from crossloc_amd import synth
    pose, info = forward_rgb(coords, n_hyp, thr, focal, ppx, ppy, alpha, max_reproj, sub, image=image)
    ref, rd = xo.forward_rgb(coords, n_hyp, thr, focal, ppx, ppy, alpha, max_reproj, sub, image=image, debug=True)
    same_cells = np.all(info["cells"] == rd["cells"], axis=1) & (info["tries"] == rd["tries"])
    dup = np.array([len(set(c[:3])) < 3 or c[3] in c[:3] for c in rd["cells"].tolist()])
    ds = np.abs(info["scores"] - rd["scores"])
    clean = same_cells & ~dup
    order = np.argsort(-rd["scores"])
    dt, dr = synth.pose_error(ref.astype(np.float64), pose.astype(np.float64))
    # why do the scores of some non-duplicate hypotheses differ by more than rounding?  Look at the worst one.
    worst_cause = ""
    if clean.any() and ds[clean].max() > 1e-6:
        h = int(np.argmax(np.where(clean, ds, -1.0)))
        fr = Frame(coords, sub)
        cells = draw_cells(1305, image, h, int(rd["tries"][h]) - 1, fr.Wo, fr.Ho)
        P = np.array([[fr.c[0, y, x], fr.c[1, y, x], fr.c[2, y, x]] for x, y in cells], np.float64)
        uv = np.array([[x * sub + sub // 2, y * sub + sub // 2] for x, y in cells], np.float64)
        cands = []
        p3p_plus_one(P, uv, float(focal), float(ppx), float(ppy), all_candidates=cands)
        errs = sorted(c[0] for c in cands)
        tri = np.linalg.norm(np.cross(P[1] - P[0], P[2] - P[0])) / (np.linalg.norm(P[1] - P[0]) * np.linalg.norm(P[2] - P[0]))
        o = xo.p3p(P, uv, float(focal), float(ppx), float(ppy))
        oerr = None
        if o is not None:
            Xc = o[0] @ P[3] + o[1]
            oerr = float((ppx + focal * Xc[0] / Xc[2] - uv[3, 0]) ** 2 + (ppy + focal * Xc[1] / Xc[2] - uv[3, 1]) ** 2)
        if oerr is not None and errs and abs(oerr - errs[0]) > 1e-3 * max(oerr, errs[0], 1e-12):
            worst_cause = ("hypothesis %d: the two quartic solvers do not see the same set of real roots (a near-double root: "
                           "a complex pair for one, two real roots for the other): 4th-point squared error of the solution "
                           "kept %.3g px^2 (oracle, closed-form Ferrari) vs %.3g px^2 (np.roots); both pass the acceptance "
                           "test" % (h, oerr, errs[0]))
        elif len(errs) > 1 and errs[1] - errs[0] < 1e-3 * max(errs[1], 1e-12):
            worst_cause = ("hypothesis %d: two P3P solutions fit the 4th point equally well (squared errors %.3g / %.3g px^2): "
                           "the selection is decided by rounding" % (h, errs[0], errs[1]))
        else:
            worst_cause = ("hypothesis %d: ill-conditioned minimal set (sine of the world triangle's angle %.2g, "
                           "%d admissible P3P solutions): the two solvers' poses differ by their root precision" % (h, tri, len(errs)))
    rec = dict(image=int(image), n_hyp=int(n_hyp), cells_equal=bool(same_cells.all()), worst_score_cause=worst_cause,
               n_cell_mismatch=int((~same_cells).sum()), n_duplicate_cell_hyps=int(dup.sum()),
               winner_indep=int(info["winner"]), winner_oracle=int(rd["winner"]),
               score_gap_top2=float(rd["scores"][order[0]] - rd["scores"][order[1]]) if n_hyp > 1 else 0.0,
               max_dscore_clean=float(ds[clean].max()) if clean.any() else 0.0,
               median_dscore_clean=float(np.median(ds[clean])) if clean.any() else 0.0,
               max_dscore_dup=float(ds[same_cells & dup].max()) if (same_cells & dup).any() else 0.0,
               rounds_indep=int(info["rounds"]), rounds_oracle=int(rd["rounds"]),
               inliers_indep=int(info["inliers"]), inliers_oracle=int(rd["inliers"]),
               dpose_m=float(dt), dpose_deg=float(dr), pose_bits_equal=bool(np.array_equal(ref, pose)))
    if gt_pose is not None:
        rec["gt_err_m"], rec["gt_err_deg"] = (float(v) for v in synth.pose_error(gt_pose, pose))
    return rec
