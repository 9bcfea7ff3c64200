"""GPU: two processes through the PRODUCT path at once (SURVEY.md 8e preflight for the 8-GPU runs the driver launches).

A one-GPU box cannot host an RCCL group of two ranks (RCCL refuses two ranks on one device), so `XL_BENCH_SHARED_GPU=1` puts
both ranks of `bench.py --gpus 2` on cuda:0 with gloo for the gather / max-over-ranks: everything else is the real thing - the
launcher re-exec under torch.distributed.run, two processes loading libcrossloc_hip.so (and racing through _lib.lib()), the
real CNN and solver kernels of both ranks sharing the chip, the per-rank global image indices, the all-gather of the error rows.
Checked: the line is strict JSON, 2 x 3 x 8 = 48 rows were gathered, and every pose of both ranks is BITWISE the pose a single
process computes for the same global image index from the same scene coordinates."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _reject_constant(name):
    raise ValueError("non-RFC-8259 token in the bench line: " + name)


def test_two_ranks_share_one_gpu_through_the_product_path(tmp_path):
    from crossloc_amd import synth
    import dsacstar
    B, K, NH = 8, 3, 256
    env = dict(os.environ, XL_BENCH_SHARED_GPU="1", XL_BENCH_DUMP_POSES=str(tmp_path / "poses"), HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("XL_BENCH_STUB", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", str(K), "--warmup", "1", "--batch", str(B),
           "--no-cpu-baseline", "--no-secondary"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0], parse_constant=_reject_constant)
    assert out["n_gpus"] == 2 and out["steps"] == K and out["config"]["batch_per_gpu"] == B
    assert out["config"]["rows_gathered"] == 2 * K * B == 48
    assert out["config"]["rccl_ranks"] == 0 and "preflight" in out["config"]
    assert out["value"] > 0 and out["roofline"]["launches_timed"] > 0
    # per-image poses of both ranks against a one-process run of the same global indices
    dev = torch.device("cuda:0")
    for rank in range(2):
        got = np.load(str(tmp_path / "poses") + ".rank%d.npy" % rank)
        assert got.shape == (K * B, 4, 4)
        coords = torch.from_numpy(synth.make_batch(2021 + 1000 * rank, B, noise=0.5, outlier_ratio=0.3)[0]).to(dev)
        for s in range(K):
            poses = torch.zeros((B, 4, 4), dtype=torch.float32, device=dev)
            dsacstar.forward_rgb_batch(coords, poses, NH, 10.0, synth.FOCAL, 360.0, 240.0, 100.0, 100.0, 8,
                                       image0=(s * 2 + rank) * B)
            torch.cuda.synchronize()
            assert np.array_equal(poses.cpu().numpy(), got[s * B:(s + 1) * B]), (rank, s)
