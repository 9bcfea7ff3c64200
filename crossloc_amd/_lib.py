"""ctypes binding of libcrossloc_hip.so (the C ABI declared in include/*.h).

The product path has no CPU fallback: if the HIP library is missing this raises, loudly.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcrossloc_hip.so")
_lib = None

c_i32, c_i64, c_u32, c_u64 = ctypes.c_int, ctypes.c_int64, ctypes.c_uint32, ctypes.c_uint64
c_f, c_vp = ctypes.c_float, ctypes.c_void_p


class XlError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        # a library built from OTHER sources than the ones next to it (a tree copied between an edit and a rebuild) binds
        # the wrong ABI: rebuild it here when the toolchain is present (one minute, once), otherwise say so loudly
        from . import build as _build
        try:
            stale = not os.path.exists(LIB_PATH) or (os.path.exists(_build.STAMP) and _build.needs_build())
        except OSError as e:
            # a deployed tree that ships the library without csrc/ or include/: nothing to compare against, load what is there
            if not os.path.exists(LIB_PATH):
                raise XlError("libcrossloc_hip.so is not built (%s) and its sources are not readable (%s); there is no CPU "
                              "fallback." % (LIB_PATH, e))
            import warnings
            warnings.warn("crossloc_amd: cannot check libcrossloc_hip.so against its sources (%s); loading it as is" % (e,))
            stale = False
        if stale:
            try:
                _build.build()                                           # serialised across processes by a file lock
            except Exception as e:                                       # no hipcc on this machine
                if not os.path.exists(LIB_PATH):
                    raise XlError("libcrossloc_hip.so is not built (%s) and could not be built here (%s). Run "
                                  "`python -m crossloc_amd.build` (or __graft_entry__.build()); there is no CPU fallback."
                                  % (LIB_PATH, e))
                raise XlError("libcrossloc_hip.so is STALE (built from other sources than csrc/ and include/ hold now) and "
                              "could not be rebuilt here (%s). Run `python -m crossloc_amd.build`." % (e,))
        # torch first (like the reference: README.md:51 "import torch before dsacstar"): its HIP runtime must be the
        # one already mapped when this library resolves libamdhip64, otherwise the process holds two runtimes and
        # the second one sees no device
        import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        L.xl_status_string.restype = ctypes.c_char_p
        L.xl_status_string.argtypes = [c_i32]
        L.xl_last_hip_error.restype = ctypes.c_char_p
        L.xl_dsac_forward_rgb_batch.restype = c_i32
        L.xl_dsac_forward_rgb_batch.argtypes = [c_vp, c_i64, c_i64, c_i64, c_i64, c_i32, c_i32, c_i32, c_vp,
                                                c_i32, c_f, c_f, c_f, c_f, c_f, c_f, c_i32, c_vp,
                                                c_u64, c_u64, c_u64, c_u32, c_vp, c_vp, c_vp, c_vp, c_vp]
        L.xl_dsac_forward_rgb_host.restype = c_i32
        L.xl_dsac_forward_rgb_host.argtypes = [c_vp, c_i64, c_i64, c_i64, c_i32, c_i32, c_vp, c_i32, c_f, c_f, c_f,
                                               c_f, c_f, c_f, c_i32, c_u64, c_u64, c_u32, c_vp, c_vp, c_vp, c_vp]
        L.xl_dsac_backward_rgb_batch.restype = c_i32
        L.xl_dsac_backward_rgb_batch.argtypes = [c_vp, c_i64, c_i64, c_i64, c_i64, c_i32, c_i32, c_i32,
                                                 c_vp, c_i64, c_i64, c_i64, c_i64, c_vp, c_vp,
                                                 c_i32, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i32,
                                                 c_vp, c_u64, c_u64, c_u64, c_u32, c_vp, c_vp]
        for name in ("xl_dsac_forward_rgbd", "xl_dsac_backward_rgbd"):
            getattr(L, name).restype = c_i32
        _lib = L
    return _lib


def check(status):
    if status != 0:
        L = lib()
        msg = L.xl_status_string(status).decode()
        if status == -3:
            msg += ": " + L.xl_last_hip_error().decode()
        raise XlError("crossloc_hip: %s (status %d)" % (msg, status))
