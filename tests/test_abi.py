"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/*.h declares; the Python shims mirror the reference names.  No compute (no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    names = []
    inc = os.path.join(ROOT, "include")
    for fn in sorted(os.listdir(inc)):
        src = open(os.path.join(inc, fn)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names += re.findall(r"\b(xl_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


@pytest.fixture(scope="module")
def hiplib():
    from crossloc_amd import build
    return ctypes.CDLL(build.build())


def test_library_exports_every_declared_symbol(hiplib):
    names = _declared_symbols()
    assert "xl_dsac_forward_rgb_batch" in names and "xl_dsac_forward_rgb_host" in names
    for n in names:
        assert hasattr(hiplib, n), "missing export: " + n


def test_status_strings_and_stub_entry_points(hiplib):
    hiplib.xl_status_string.restype = ctypes.c_char_p
    assert hiplib.xl_status_string(0) == b"ok"
    for code in (-1, -2, -3, -4):
        assert len(hiplib.xl_status_string(code)) > 0
    for n in ("xl_dsac_forward_rgbd", "xl_dsac_backward_rgbd"):
        assert getattr(hiplib, n)() == -4
    assert hiplib.xl_dsac_backward_rgb_batch(None, 0, 0, 0, 0, 1, 60, 90, None) == -1
    # argument validation happens before any HIP call, so it is testable without a GPU
    assert hiplib.xl_dsac_forward_rgb_batch(None, 0, 0, 0, 0, 1, 60, 90, None, 64) == -1


def test_dsacstar_module_surface(hiplib):
    import dsacstar
    for n in ("forward_rgb", "backward_rgb", "forward_rgbd", "backward_rgbd"):   # dsacstar.cpp:887-892
        assert callable(getattr(dsacstar, n))
    with pytest.raises(NotImplementedError):
        dsacstar.forward_rgbd()
    with pytest.raises(NotImplementedError):
        dsacstar.backward_rgbd()
    import torch as _t
    with pytest.raises(RuntimeError):              # shape check before any device work (no GPU here)
        dsacstar.backward_rgb(_t.zeros(1, 3, 60, 90), _t.zeros(1, 3, 60, 91), _t.eye(4), 64, 10.0, 480.0, 360.0, 240.0,
                              1.0, 1.0, 100.0, 100.0, 100.0, 8, 1)
    import torch
    with pytest.raises(RuntimeError):
        dsacstar.forward_rgb(torch.zeros(3, 60, 90), torch.zeros(4, 4), 64, 10.0, 480.0, 360.0, 240.0, 100.0, 100.0, 8)
    with pytest.raises(RuntimeError):
        dsacstar.forward_rgb(torch.zeros(1, 3, 60, 90, dtype=torch.float64), torch.zeros(4, 4), 64, 10.0, 480.0,
                             360.0, 240.0, 100.0, 100.0, 8)
