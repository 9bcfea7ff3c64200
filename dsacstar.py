"""Top-level `dsacstar` import name of the reference extension (utils/evaluation.py:11 does `import dsacstar`).

forward_rgb / backward_rgb / forward_rgbd / backward_rgbd (dsacstar.cpp:887-892) come from the COMPILED binding
crossloc_amd/_dsacstar_native (csrc/dsacstar_ext.cpp: pybind11 + at::Tensor over the C ABI of libcrossloc_hip.so) when it is
built - `python -m crossloc_amd.build`, __graft_entry__.build() - and from the ctypes shim crossloc_amd/dsacstar.py otherwise
(XL_DSACSTAR_PY=1 forces the shim).  The batched entry points and the sampler's image counter live in the shim either way.
`dsacstar.NATIVE` is the compiled module or None."""
import os as _os

from crossloc_amd.dsacstar import *  # noqa: F401,F403
from crossloc_amd.dsacstar import (RANSAC_SEED, MAX_HYPOTHESES_TRIES, MAX_REF_STEPS, backward_rgb,  # noqa: F401
                                   backward_rgb_batch, backward_rgbd, forward_rgb, forward_rgb_batch, forward_rgbd,
                                   set_image_index)
from crossloc_amd import dsacstar as _impl

NATIVE = None
if not _os.environ.get("XL_DSACSTAR_PY"):
    try:
        from crossloc_amd import _lib as _l, build as _b
        _l.lib()                                        # the (fresh) library first: the binding links against the same file
        if _os.path.exists(_b.EXT_SRC):
            _b.build_ext()                              # no-op when its stamp matches; needs g++ and the torch headers otherwise
        from crossloc_amd import _dsacstar_native as NATIVE
        forward_rgb, backward_rgb = NATIVE.forward_rgb, NATIVE.backward_rgb
        forward_rgbd, backward_rgbd = NATIVE.forward_rgbd, NATIVE.backward_rgbd
    except Exception as _e:                             # not built / cannot be built here: the ctypes shim serves the same names
        NATIVE = None
        NATIVE_ERROR = repr(_e)


def __getattr__(name):
    return getattr(_impl, name)
