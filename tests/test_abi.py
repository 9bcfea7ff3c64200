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


def test_dsacstar_resolves_to_the_compiled_binding(hiplib):
    """`import dsacstar` serves forward_rgb / backward_rgb / forward_rgbd / backward_rgbd (dsacstar.cpp:887-892) from the compiled
    pybind11 / ATen module, like the reference's own extension; the ctypes shim stays importable as the fallback."""
    import shutil
    if not shutil.which("g++"):
        pytest.skip("no g++ here: the binding cannot be built")
    import dsacstar
    import crossloc_amd.dsacstar as shim
    assert dsacstar.NATIVE is not None, getattr(dsacstar, "NATIVE_ERROR", None)
    assert dsacstar.NATIVE.__file__.endswith(".so")
    for n in ("forward_rgb", "backward_rgb", "forward_rgbd", "backward_rgbd"):
        assert getattr(dsacstar, n) is getattr(dsacstar.NATIVE, n) and getattr(dsacstar, n) is not getattr(shim, n)
    assert dsacstar.forward_rgb_batch is shim.forward_rgb_batch          # the batched entry points live in the shim


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


def test_concurrent_imports_with_a_stale_stamp_leave_one_loadable_library():
    """bench.py --gpus N / torchrun: every rank runs _lib.lib() at the same moment.  With a stale stamp exactly one of them
    may compile (file lock in crossloc_amd/build.py), the library is moved into place atomically, and all of them load it."""
    import shutil
    import subprocess
    import sys
    from crossloc_amd import build
    if not (os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("hipcc")):
        pytest.skip("no hipcc here: a stale library cannot be rebuilt")
    build.build()
    with open(build.STAMP, "w") as f:
        f.write("stale\n")
    assert build.needs_build()
    code = ("import ctypes, os, sys, time; sys.path.insert(0, %r); t0 = time.time();"
            "from crossloc_amd import _lib; L = _lib.lib();"
            "L.xl_status_string.restype = ctypes.c_char_p; assert L.xl_status_string(0) == b'ok';"
            "print('built' if time.time() - t0 > 8 else 'waited-or-found')" % ROOT)
    procs = [subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for _ in range(3)]
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-400:] for o in outs]
    assert not build.needs_build()
    with open(build.STAMP) as f:
        assert f.read().strip() == build.source_hash()
    assert not [d for d in os.listdir(os.path.dirname(build.LIB)) if d.startswith(".build.")]     # no leftovers
    ctypes.CDLL(build.LIB)


def test_storage_use_count_semantics_the_gradient_buffers_rely_on():
    """crossloc_amd.networks._sole_owner: a result buffer of the backward pass is reused only when NOTHING else
    references its storage.  That reads torch's private storage use count with a threshold of 2 (the tensor + the storage object the
    query creates): pinned here for the PyTorch in use - alone: sole owner; with a live view, or a second tensor on the storage: not."""
    torch = pytest.importorskip("torch")
    from crossloc_amd.networks import _sole_owner
    buf = torch.zeros(64)
    assert _sole_owner(buf)
    view = buf[8:16]
    assert not _sole_owner(buf)
    del view
    assert _sole_owner(buf)
    keep = buf.view(8, 8)
    assert not _sole_owner(buf)
    del keep
    assert _sole_owner(buf)
    assert not getattr(_sole_owner, "warned", False)       # the private API exists in this PyTorch: no fallback warning
