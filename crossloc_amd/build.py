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
SOURCES = ["xl_dsac.hip", "xl_cnn.hip", "xl_cnn_bwd.hip", "xl_gemm_split.hip", "xl_stem_split.hip", "xl_wgrad_split.hip", "xl_pack.hip", "xl_loss.hip", "xl_optim.hip", "xl_data.hip"]
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


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    objs, procs = [], []
    for src in sources():
        obj = src[:-4] + ".o"
        cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((subprocess.Popen(cmd), cmd))
        objs.append(obj)
    for p, cmd in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    with open(STAMP, "w") as f:
        f.write(source_hash() + "\n")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
