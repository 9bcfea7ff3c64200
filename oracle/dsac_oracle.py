"""ctypes front-end of oracle/dsac_oracle.c — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product path (crossloc_amd/, dsacstar.py) must never do so.  Parity vs the reference binary
is unpinned (OpenCV absent, see dsac_oracle.c header); the oracle is pinned by analytic tests.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# XL_ORACLE_LIB: another build of the same sources (tests/test_oracle_sanitizers.py: the ASan + UBSan one)
_LIB_PATH = os.environ.get("XL_ORACLE_LIB") or os.path.join(_HERE, "libxl_oracle.so")
_lib = None


def build(force=False):
    """Compile the C restatement (gcc, seconds)."""
    src = os.path.join(_HERE, "dsac_oracle.c")
    srcs = [src, os.path.join(_HERE, "dsac_bwd_oracle.c")]
    if os.environ.get("XL_ORACLE_LIB"):
        return _LIB_PATH             # (built by whoever set the variable)
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()                      # no-op when the library is newer than its sources
        L = ctypes.CDLL(_LIB_PATH)
        c_i64, c_u64, c_u32, c_int, c_f = ctypes.c_int64, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int, ctypes.c_float
        vp = ctypes.c_void_p
        L.xo_dsac_forward_rgb.restype = c_int
        L.xo_dsac_forward_rgb.argtypes = [vp, c_i64, c_i64, c_i64, c_int, c_int, vp, c_int, c_f, c_f, c_f, c_f,
                                          c_f, c_f, c_int, c_u64, c_u64, c_u32, vp, vp, vp, vp]
        L.xo_test_p3p.restype = c_int
        L.xo_test_p3p.argtypes = [vp, vp, ctypes.c_double, ctypes.c_double, ctypes.c_double, vp]
        L.xo_test_score.restype = ctypes.c_double
        L.xo_test_score.argtypes = [vp, c_i64, c_i64, c_i64, c_int, c_int, vp, c_f, c_f, c_f, c_f, c_f, c_f, c_int]
        L.xo_test_exp.restype = ctypes.c_double
        L.xo_test_exp.argtypes = [ctypes.c_double]
        L.xo_test_sincos.restype = None
        L.xo_test_sincos.argtypes = [ctypes.c_double, vp, vp]
        L.xo_test_quartic.restype = c_int
        L.xo_test_quartic.argtypes = [vp, vp]
        L.xo_test_draws.restype = None
        L.xo_test_draws.argtypes = [c_u64, c_u64, c_u32, c_u32, c_int, c_int, vp]
        L.xo_num_threads.restype = c_int
        c_d = ctypes.c_double
        L.xo_dsac_backward_rgb.restype = c_int
        L.xo_dsac_backward_rgb.argtypes = [vp, c_i64, c_i64, c_i64, c_int, c_int, vp, c_i64, c_i64, c_i64, vp, c_int,
                                           c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_int, c_u64, c_u64, c_u32,
                                           vp, vp]
        L.xo_test_atan2.restype = c_d
        L.xo_test_atan2.argtypes = [c_d, c_d]
        L.xo_test_log_so3.argtypes = [vp, vp]
        L.xo_test_rodrigues_jac.argtypes = [vp, vp]
        L.xo_test_pinv6.argtypes = [vp, vp]
        L.xo_test_resid_row.restype = c_d
        L.xo_test_resid_row.argtypes = [vp, vp, vp, c_f, c_f, c_d, c_d, c_d, c_f, vp]
        L.xo_test_dproject_dobj.argtypes = [vp, vp, c_f, c_f, c_d, c_d, c_d, c_f, vp]
        L.xo_test_pose_loss.restype = c_d
        L.xo_test_pose_loss.argtypes = [vp, vp, c_d, c_d, c_d]
        L.xo_test_dloss.argtypes = [vp, vp, vp, c_d, c_d, c_d, vp]
        L.xo_test_dpnp.argtypes = [vp, vp, c_d, c_d, c_d, vp]
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def forward_rgb(coords, n_hyp, thr, focal, ppx, ppy, alpha, max_reproj, sub,
                seed=1305, image=0, max_tries=1000000, debug=False):
    """coords: float32 array [3, Ho, Wo] (any strides). Returns 4x4 float32 cam->world pose
    (and a dict of intermediates when debug=True)."""
    coords = np.asarray(coords)
    assert coords.dtype == np.float32 and coords.ndim == 3 and coords.shape[0] == 3
    _, Ho, Wo = coords.shape
    es = coords.itemsize
    sc, sy, sx = (s // es for s in coords.strides)
    pose = np.zeros((4, 4), np.float32)
    cells = np.zeros((n_hyp, 4), np.int32)
    tries = np.zeros((n_hyp,), np.int32)
    scores = np.zeros((n_hyp,), np.float64)
    dbg = np.zeros((28,), np.float64)
    rc = lib().xo_dsac_forward_rgb(_ptr(coords), sc, sy, sx, Ho, Wo, _ptr(pose), int(n_hyp), float(thr),
                                   float(focal), float(ppx), float(ppy), float(alpha), float(max_reproj),
                                   int(sub), int(seed), int(image), int(max_tries),
                                   _ptr(cells), _ptr(tries), _ptr(scores), _ptr(dbg))
    if rc != 0:
        raise RuntimeError("xo_dsac_forward_rgb failed: %d" % rc)
    if not debug:
        return pose
    return pose, dict(cells=cells, tries=tries, scores=scores, winner=int(dbg[0]), rounds=int(dbg[1]),
                      inliers=int(dbg[2]), lm_evals=int(dbg[3]), pose0=dbg[4:16].copy(), pose1=dbg[16:28].copy())


def p3p(P, uv, f, cx, cy):
    P = np.ascontiguousarray(P, np.float64).reshape(4, 3)
    uv = np.ascontiguousarray(uv, np.float64).reshape(4, 2)
    out = np.zeros(12, np.float64)
    ok = lib().xo_test_p3p(_ptr(P), _ptr(uv), float(f), float(cx), float(cy), _ptr(out))
    return (out[:9].reshape(3, 3), out[9:]) if ok else None


def score(coords, R, t, thr, alpha, max_reproj, focal, ppx, ppy, sub):
    coords = np.asarray(coords)
    assert coords.dtype == np.float32
    es = coords.itemsize
    sc, sy, sx = (s // es for s in coords.strides)
    Rt = np.concatenate([np.asarray(R, np.float64).reshape(9), np.asarray(t, np.float64).reshape(3)])
    return lib().xo_test_score(_ptr(coords), sc, sy, sx, coords.shape[1], coords.shape[2], _ptr(Rt),
                               float(thr), float(alpha), float(max_reproj), float(focal), float(ppx),
                               float(ppy), int(sub))


def exp(x):
    return lib().xo_test_exp(float(x))


def sincos(x):
    s, c = ctypes.c_double(), ctypes.c_double()
    lib().xo_test_sincos(float(x), ctypes.byref(s), ctypes.byref(c))
    return s.value, c.value


def quartic(A):
    A = np.ascontiguousarray(A, np.float64)
    r = np.zeros(4, np.float64)
    n = lib().xo_test_quartic(_ptr(A), _ptr(r))
    return r[:n]


def draws(seed, image, hyp, t, Wo, Ho):
    out = np.zeros(8, np.int32)
    lib().xo_test_draws(int(seed), int(image), int(hyp), int(t), int(Wo), int(Ho), _ptr(out))
    return out.reshape(4, 2)


def num_threads():
    return lib().xo_num_threads()


def set_num_threads(n):
    lib().xo_set_num_threads(int(n))


# ------------------------------------------------------------------------------------------ backward_rgb

BWD_REC = 64    # doubles per hypothesis in the debug record (layout: dsac_bwd_oracle.c, XO_BWD_REC)


def backward_rgb(coords, grad, gt_pose, n_hyp, thr, focal, ppx, ppy, w_rot, w_trans, soft_clamp, alpha, max_reproj,
                 sub, seed, image=0, max_tries=1000000, debug=False):
    """coords: float32 [3,Ho,Wo]; grad: float32 [3,Ho,Wo], accumulated in place; gt_pose: 4x4 cam->world.
    Returns the expected pose loss (and the per-hypothesis records [n_hyp, BWD_REC] when debug=True)."""
    coords = np.asarray(coords)
    assert coords.dtype == np.float32 and coords.ndim == 3 and coords.shape[0] == 3
    assert grad.dtype == np.float32 and grad.shape == coords.shape
    _, Ho, Wo = coords.shape
    es = coords.itemsize
    sc, sy, sx = (s // es for s in coords.strides)
    gc, gy, gx = (s // es for s in grad.strides)
    gt = np.ascontiguousarray(gt_pose, np.float32).reshape(16)
    loss = np.zeros(1, np.float64)
    rec = np.zeros((n_hyp, BWD_REC), np.float64)
    rc = lib().xo_dsac_backward_rgb(_ptr(coords), sc, sy, sx, Ho, Wo, _ptr(grad), gc, gy, gx, _ptr(gt), int(n_hyp),
                                    float(thr), float(focal), float(ppx), float(ppy), float(w_rot), float(w_trans),
                                    float(soft_clamp), float(alpha), float(max_reproj), int(sub), int(seed), int(image),
                                    int(max_tries), _ptr(loss), _ptr(rec))
    if rc != 0:
        raise RuntimeError("xo_dsac_backward_rgb failed: %d" % rc)
    return (float(loss[0]), rec) if debug else float(loss[0])


def atan2(y, x):
    return lib().xo_test_atan2(float(y), float(x))


def log_so3(R):
    R = np.ascontiguousarray(R, np.float64).reshape(9)
    r = np.zeros(3, np.float64)
    lib().xo_test_log_so3(_ptr(R), _ptr(r))
    return r


def rodrigues_jac(r):
    r = np.ascontiguousarray(r, np.float64).reshape(3)
    d = np.zeros(27, np.float64)
    lib().xo_test_rodrigues_jac(_ptr(r), _ptr(d))
    return d.reshape(9, 3)


def pinv6(A):
    A = np.ascontiguousarray(A, np.float64).reshape(36)
    o = np.zeros(36, np.float64)
    lib().xo_test_pinv6(_ptr(A), _ptr(o))
    return o.reshape(6, 6)


def _rt12(R, t):
    return np.concatenate([np.asarray(R, np.float64).reshape(9), np.asarray(t, np.float64).reshape(3)])


def resid_row(R, t, r, X, px, py, f, cx, cy, max_reproj):
    Rt, r, X = _rt12(R, t), np.ascontiguousarray(r, np.float64), np.ascontiguousarray(X, np.float64)
    J = np.zeros(6, np.float64)
    e = lib().xo_test_resid_row(_ptr(Rt), _ptr(r), _ptr(X), float(px), float(py), float(f), float(cx), float(cy),
                                float(max_reproj), _ptr(J))
    return e, J


def dproject_dobj(R, t, X, px, py, f, cx, cy, max_reproj):
    Rt, X = _rt12(R, t), np.ascontiguousarray(X, np.float64)
    o = np.zeros(3, np.float64)
    lib().xo_test_dproject_dobj(_ptr(Rt), _ptr(X), float(px), float(py), float(f), float(cx), float(cy),
                                float(max_reproj), _ptr(o))
    return o


def pose_loss(R, t, gt_pose, w_rot, w_trans, cut):
    Rt = _rt12(R, t)
    gt = np.ascontiguousarray(gt_pose, np.float32).reshape(16)
    return lib().xo_test_pose_loss(_ptr(Rt), _ptr(gt), float(w_rot), float(w_trans), float(cut))


def dloss(R, t, r, gt_pose, w_rot, w_trans, cut):
    Rt, r = _rt12(R, t), np.ascontiguousarray(r, np.float64)
    gt = np.ascontiguousarray(gt_pose, np.float32).reshape(16)
    j = np.zeros(6, np.float64)
    lib().xo_test_dloss(_ptr(Rt), _ptr(r), _ptr(gt), float(w_rot), float(w_trans), float(cut), _ptr(j))
    return j


def dpnp(obj, uv, f, cx, cy):
    obj = np.ascontiguousarray(obj, np.float32).reshape(12)
    uv = np.ascontiguousarray(uv, np.float64).reshape(8)
    J = np.zeros(72, np.float64)
    lib().xo_test_dpnp(_ptr(obj), _ptr(uv), float(f), float(cx), float(cy), _ptr(J))
    return J.reshape(6, 12)
