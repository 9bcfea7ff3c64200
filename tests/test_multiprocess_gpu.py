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


@pytest.mark.parametrize("R,B,K", [(2, 8, 3), (8, 4, 2)])
def test_ranks_share_one_gpu_through_the_product_path(tmp_path, R, B, K):
    """R = 2: the round-4 preflight.  R = 8 (round 5): the rank count of the driver's SCALE run - eight processes of
    `bench.py --gpus 8 --batch 4 --steps 2` on cuda:0, 64 rows gathered, every pose bitwise the one-process result, the line
    carries the per-rank step times."""
    from crossloc_amd import synth
    import dsacstar
    NH = 256
    env = dict(os.environ, XL_BENCH_SHARED_GPU="1", XL_BENCH_DUMP_POSES=str(tmp_path / "poses"), HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("XL_BENCH_STUB", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(R), "--steps", str(K), "--warmup", "1", "--batch", str(B),
           "--no-cpu-baseline", "--no-secondary"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0], parse_constant=_reject_constant)
    assert out["n_gpus"] == R and out["steps"] == K and out["config"]["batch_per_gpu"] == B
    assert out["config"]["rows_gathered"] == R * K * B
    assert out["config"]["rccl_ranks"] == 0 and "preflight" in out["config"]
    assert out["value"] > 0 and out["roofline"]["launches_timed"] > 0
    pr = out["config"]["per_rank_ms_per_step"]
    assert 0 < pr["min"] <= pr["max"] <= out["ms_per_step"] * 1.001
    # per-image poses of every rank against a one-process run of the same global indices
    dev = torch.device("cuda:0")
    for rank in range(R):
        got = np.load(str(tmp_path / "poses") + ".rank%d.npy" % rank)
        assert got.shape == (K * B, 4, 4)
        coords = torch.from_numpy(synth.make_batch(2021 + 1000 * rank, B, noise=0.5, outlier_ratio=0.3)[0]).to(dev)
        for s in range(K):
            poses = torch.zeros((B, 4, 4), dtype=torch.float32, device=dev)
            dsacstar.forward_rgb_batch(coords, poses, NH, 10.0, synth.FOCAL, 360.0, 240.0, 100.0, 100.0, 8,
                                       image0=(s * R + rank) * B)
            torch.cuda.synchronize()
            assert np.array_equal(poses.cpu().numpy(), got[s * B:(s + 1) * B]), (rank, s)


def test_a_rank_of_the_8_gpu_run_fits_its_gpu_many_times_over():
    """weights replicated, every rank runs the 95-frame plan of the headline on its own GPU.  What one rank holds -
    the plan's activation pool, its packed weights, the frames and the solver's buffers - by arithmetic from the plan's own tensors
    and by the allocator's peak: below 1/8 of the 288 GB of an MI355X (so even eight ranks on ONE device, the shared-GPU rehearsal
    at full batch, would fit)."""
    from crossloc_amd import networks, synth
    from crossloc_amd.weights import seeded_state_dict
    dev = torch.device("cuda:0")
    torch.cuda.reset_peak_memory_stats(dev)
    base = torch.cuda.memory_allocated(dev)
    net = networks.TransPoseNet(torch.tensor(synth.SCENE_MEAN, dtype=torch.float32), False, False, 2, 2, 3, 1)
    net.load_state_dict(seeded_state_dict(net, seed=2021))
    net = net.to(dev).eval()
    x = torch.rand((95, 3, 480, 720), device=dev)
    with torch.no_grad():
        y = net(x)
    torch.cuda.synchronize()
    plan = list(net._plans.values())[0]
    held = {}
    for t in list(plan.keep) + [e[0] for e in plan.packed.values()] + [e[0] for e in plan.packed_split.values()] + \
            [e[0] for e in plan.packed_1x1.values()] + [e[0] for e in plan.packed_pair.values()] + [plan.stats, plan.coeff]:
        held[t.untyped_storage().data_ptr()] = t.untyped_storage().nbytes()
    plan_bytes = sum(held.values()) + x.numel() * 4 + y.numel() * 4 + sum(p.numel() * 4 for p in net.parameters())
    peak = torch.cuda.max_memory_allocated(dev) - base
    budget = 288e9 / 8
    print("95-frame plan: %.2f GB held by the plan + frames + weights, allocator peak %.2f GB; 1/8 of HBM = %.1f GB" % (
        plan_bytes / 1e9, peak / 1e9, budget / 1e9))
    assert plan_bytes < budget and peak < budget
