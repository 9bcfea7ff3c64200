"""Builds libcrossloc_hip.so (all HIP kernels + the C ABI) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with gpurun snapshots.
-ffp-contract=off is part of the arithmetic contract of the solver (see csrc/xl_dsac.hip).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INC = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, "libcrossloc_hip.so")
SOURCES = ["xl_dsac.hip", "xl_cnn.hip", "xl_cnn_bwd.hip", "xl_gemm_split.hip", "xl_gemm_pair.hip", "xl_stem_split.hip", "xl_stem_pair.hip", "xl_stem_fused.hip", "xl_stem_dgrad.hip", "xl_wgrad_split.hip", "xl_wgrad_pair.hip", "xl_pack.hip", "xl_loss.hip", "xl_optim.hip", "xl_data.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wno-unused-result"]


def sources():
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


STAMP = LIB + ".srchash"


def source_hash():
    """Content hash of everything the library is built from (sources, private and public headers, flags): the
    rebuild decision must not depend on file times, which a snapshot copy of the tree does not preserve."""
    import hashlib
    h = hashlib.sha256(" ".join(FLAGS).encode())
    deps = sources() + sorted(os.path.join(INC, f) for f in os.listdir(INC))
    deps += sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))
    for d in deps:
        h.update(os.path.basename(d).encode())
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def needs_build():
    if not os.path.exists(LIB) or not os.path.exists(STAMP):
        return True
    with open(STAMP) as f:
        return f.read().strip() != source_hash()


class _BuildLock:
    """Exclusive advisory lock (flock on libcrossloc_hip.so.lock) around the rebuild decision and the build: under
    torch.distributed.run every rank imports the package at the same moment, and only one of them may compile."""

    def __enter__(self):
        import fcntl
        self.f = open(LIB + ".lock", "w")
        fcntl.flock(self.f, fcntl.LOCK_EX)
        return self

    def __exit__(self, *exc):
        import fcntl
        fcntl.flock(self.f, fcntl.LOCK_UN)
        self.f.close()


def build(force=False, verbose=False):
    """Compile and link the library if the sources changed.  Safe to call from several processes at once: the ranks
    serialise on a file lock, the one that gets it first builds, the others find the stamp up to date when their turn
    comes.  Objects go to a directory private to the building process and the library and its stamp are moved into place
    with os.replace(), so a process that loads the library without the lock never maps a half-written file."""
    if not force and not needs_build():
        return LIB
    import shutil
    import tempfile
    with _BuildLock():
        if not force and not needs_build():                      # another process built it while this one waited
            return LIB
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        if not os.path.exists(hipcc):
            hipcc = "hipcc"
        want = source_hash()
        tmp = tempfile.mkdtemp(prefix=".build.", dir=HERE)
        try:
            objs, procs = [], []
            for src in sources():
                obj = os.path.join(tmp, os.path.basename(src)[:-4] + ".o")
                cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
                if verbose:
                    print(" ".join(cmd), file=sys.stderr)
                procs.append((subprocess.Popen(cmd), cmd))
                objs.append(obj)
            failed = [cmd for p, cmd in procs if p.wait() != 0]
            if failed:
                raise RuntimeError("hipcc failed: " + " ".join(failed[0]))
            out = os.path.join(tmp, "libcrossloc_hip.so")
            subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
            with open(os.path.join(tmp, "srchash"), "w") as f:
                f.write(want + "\n")
            if os.path.exists(STAMP):
                os.remove(STAMP)                                 # never an old library beside a new stamp, or vice versa
            os.replace(out, LIB)
            os.replace(os.path.join(tmp, "srchash"), STAMP)
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    return LIB


EXT_SRC = os.path.join(CSRC, "dsacstar_ext.cpp")
EXT_NAME = "_dsacstar_native"


def ext_path():
    import sysconfig
    return os.path.join(HERE, EXT_NAME + sysconfig.get_config_var("EXT_SUFFIX"))


def _ext_hash():
    import hashlib
    import torch
    h = hashlib.sha256(torch.__version__.encode())
    for d in (EXT_SRC, os.path.join(INC, "crossloc_dsac.h")):
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def build_ext(force=False, verbose=False):
    """The compiled `dsacstar` binding (csrc/dsacstar_ext.cpp: pybind11 + ATen, host-only C++, links libcrossloc_hip.so): built
    in-tree with g++ against the torch headers of this interpreter, under the same file lock as the library.  Optional: the
    ctypes shim is the fallback (dsacstar.py), so a missing compiler or torch header is reported, not fatal."""
    out, stamp = ext_path(), ext_path() + ".srchash"
    want = _ext_hash()
    if not force and os.path.exists(out) and os.path.exists(stamp) and open(stamp).read().strip() == want:
        return out
    import shutil
    import sysconfig
    import tempfile
    import torch
    from torch.utils import cpp_extension as ce
    build()                                                             # the library it links against
    with _BuildLock():
        if not force and os.path.exists(out) and os.path.exists(stamp) and open(stamp).read().strip() == want:
            return out
        tmp = tempfile.mkdtemp(prefix=".build.", dir=HERE)
        try:
            so = os.path.join(tmp, os.path.basename(out))
            cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-attributes", EXT_SRC, "-o", so,
                   "-DTORCH_EXTENSION_NAME=" + EXT_NAME, "-DTORCH_API_INCLUDE_EXTENSION_H",
                   "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch.compiled_with_cxx11_abi()),
                   "-I" + sysconfig.get_paths()["include"]]
            cmd += ["-I" + p for p in ce.include_paths()]
            libdirs = ce.library_paths()
            cmd += ["-L" + p for p in libdirs] + ["-L" + HERE]
            cmd += ["-lc10", "-ltorch_cpu", "-ltorch", "-ltorch_python", "-l:libcrossloc_hip.so",
                    "-Wl,-rpath,$ORIGIN"] + ["-Wl,-rpath," + p for p in libdirs]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.check_call(cmd)
            with open(os.path.join(tmp, "srchash"), "w") as f:
                f.write(want + "\n")
            if os.path.exists(stamp):
                os.remove(stamp)
            os.replace(so, out)
            os.replace(os.path.join(tmp, "srchash"), stamp)
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_ext(force="--force" in sys.argv, verbose=True))
