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
_LIB_PATH = os.path.join(_HERE, "libxl_oracle.so")
_lib = None


def build(force=False):
    """Compile the C restatement (gcc, seconds)."""
    src = os.path.join(_HERE, "dsac_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
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
