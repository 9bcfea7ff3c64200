"""The N>1 path on CPU (gloo, world_size 2): image-level sharding i % R, ONE all-gather of per-image
(t_err, r_err) padded with NaN for ragged shards, identical global order / median for every rank count
(SURVEY.md §8e).  The per-image localisation itself is done by the CPU oracle here — the product path
needs a GPU — so this exercises exactly the distributed logic of crossloc_amd.evaluation."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from crossloc_amd import evaluation, synth

K = 5            # ragged over 2 ranks: 3 + 2 images
N_HYP = 16


def _localize(indices):
    from oracle import dsac_oracle
    rows = []
    for i in indices:
        sc = synth.make_scene(3000 + i, noise=0.5, outlier_ratio=0.3)
        pose = dsac_oracle.forward_rgb(sc["coords"], N_HYP, 10.0, 480.0, 360.0, 240.0, 100.0, 100.0, 8, image=i)
        rows.append(evaluation.get_pose_err(sc["pose"], pose))
    return torch.tensor(rows, dtype=torch.float64).reshape(-1, 2)


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    local = _localize(evaluation.shard_indices(K, rank, world))
    allv = evaluation.gather_errors(local, K, rank, world)
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), allv.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_indices_cover_every_image_once():
    for world in (1, 2, 4, 8):
        got = sorted(i for r in range(world) for i in evaluation.shard_indices(1024 + 3, r, world))
        assert got == list(range(1027))


def test_two_rank_gather_equals_single_process(tmp_path):
    single = evaluation.gather_errors(_localize(range(K)), K, 0, 1).numpy()
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / "rank0.npy")
    r1 = np.load(tmp_path / "rank1.npy")
    assert r0.shape == (K, 2) and not np.isnan(r0).any()
    assert np.array_equal(r0, r1)                       # every rank holds the same gathered table
    assert np.array_equal(r0, single)                   # and it is bit-identical to the 1-rank run
    assert np.median(r0[:, 0]) == np.median(single[:, 0]) and np.median(r0[:, 1]) == np.median(single[:, 1])


def test_accuracy_report_matches_reference_format():
    t = [0.5, 2.0, 4.0, 12.0, 40.0]
    r = [0.1, 2.5, 4.9, 8.0, 20.0]
    stats, text = evaluation.accuracy_report(t, r, [1.0, 2.0, 3.0])
    assert stats["3m3deg"] == pytest.approx(40.0) and stats["5m5deg"] == pytest.approx(60.0)      # strict <
    assert stats["30m10deg"] == pytest.approx(80.0) and stats["10m7deg"] == pytest.approx(60.0)
    assert "Median Error: 4.9 deg, 4.00 m" in text                                               # evaluation.py:226
    assert "Coordinate regression error: mean 2.0, std 0.8, median 2.0" in text


def test_get_pose_err_rotation_identities():
    def rot_z(deg):
        a = np.radians(deg)
        T = np.eye(4)
        T[:2, :2] = [[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]
        return T
    for deg in (0.0, 1e-3, 90.0, 179.999):
        t, r = evaluation.get_pose_err(rot_z(deg), np.eye(4))
        assert t == 0.0 and r == pytest.approx(deg, abs=1e-9)
    T = np.eye(4)
    T[:3, 3] = [3.0, 4.0, 12.0]
    assert evaluation.get_pose_err(T, np.eye(4))[0] == pytest.approx(13.0)
    te, re = evaluation.pose_errors(torch.tensor(np.stack([rot_z(30.0), T])), torch.eye(4).repeat(2, 1, 1))
    assert re[0].item() == pytest.approx(30.0, abs=1e-9) and te[1].item() == pytest.approx(13.0)


def _grad_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from crossloc_amd import optim
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.zeros(5, 3)), torch.nn.Parameter(torch.zeros(7)), torch.nn.Parameter(torch.zeros(2))]
    params[0].grad = torch.full((5, 3), float(rank + 1))
    params[1].grad = torch.arange(7, dtype=torch.float32) * (rank + 1)
    # params[2] has no gradient on any rank (frozen): must be skipped consistently
    optim.allreduce_gradients(params, world)
    np.save(os.path.join(out_dir, "g%d.npy" % rank), np.concatenate([params[0].grad.ravel().numpy(), params[1].grad.numpy()]))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_allreduce_averages_over_ranks(tmp_path):
    """Data-parallel training step: one flat all-reduce, mean over ranks, every rank identical."""
    mp.spawn(_grad_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    g0, g1 = np.load(tmp_path / "g0.npy"), np.load(tmp_path / "g1.npy")
    assert np.array_equal(g0, g1)
    assert np.allclose(g0[:15], 1.5) and np.allclose(g0[15:], np.arange(7) * 1.5)


def _bench_worker(rank, world, port, out_dir):
    """bench.py's own gather + median code path (gather_and_median) under gloo."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    rows = torch.tensor([[0.01 * (3 * i + rank + 1), 0.001 * (i + 7 * rank)] for i in range(6)], dtype=torch.float64)
    med = bench.gather_and_median(rows, world, dist)
    np.save(os.path.join(out_dir, "bench_rank%d.npy" % rank), np.array(med))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_gather_and_median_two_ranks(tmp_path):
    import bench
    mp.spawn(_bench_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "bench_rank0.npy"), np.load(tmp_path / "bench_rank1.npy")
    rows = torch.tensor([[0.01 * (3 * i + r + 1), 0.001 * (i + 7 * r)] for r in range(2) for i in range(6)], dtype=torch.float64)
    want = bench.gather_and_median(rows, 1)
    assert np.array_equal(r0, r1) and np.allclose(r0, want) and int(r0[2]) == 12
    assert abs(want[0] - 100.0 * float(torch.median(rows[:, 0]))) < 1e-12


def _run_bench(args, extra_env):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=600)


def test_bench_self_spawns_two_ranks_without_a_launcher():
    """`python bench.py --gpus 2` with no WORLD_SIZE: bench.py re-executes itself under torch.distributed.run (2 ranks on
    127.0.0.1) and its own init / barrier / max-over-ranks / all-gather code runs - here on CPU ranks (gloo) around the
    stand-in localiser of XL_BENCH_STUB, whose per-image errors are a known function of the global image index."""
    import json
    r = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "4"], {"XL_BENCH_STUB": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                           # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["config"]["backend"] == "gloo"
    assert out["config"]["ranks"] == 2 and out["config"]["rows_gathered"] == 2 * 3 * 4
    idx = np.arange(24, dtype=np.float64)                            # image indices 0..23 are covered exactly once
    assert out["config"]["median_err_cm"] == pytest.approx(100.0 * np.sort(0.01 * (1.0 + idx % 7))[(24 - 1) // 2])
    assert "STUB" in out["data"] and "roofline" not in out


def test_bench_two_gpus_on_a_box_without_them_fails_after_the_spawn_with_a_clear_message():
    """The real (RCCL) path on a node with fewer GPUs than ranks: every spawned rank stops with `needs 2 GPUs`, the
    launcher reports the failure - no hang in the rendezvous, no SystemExit before the spawn."""
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("this node has two GPUs")
    r = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0"], {})
    assert r.returncode != 0
    assert "needs 2 GPUs on this node" in (r.stderr + r.stdout)
    assert "torch.distributed" in r.stderr or "ChildFailedError" in r.stderr      # it did go through the launcher


def test_focal_argument_forms():
    """One focal length for the batch may arrive as a Python number, a numpy scalar or a 0-dim / one-element tensor."""
    for f in (480.0, 480, np.float32(480.0), np.float64(480.0), torch.tensor(480.0), torch.tensor([480.0]), [480.0]):
        assert evaluation._focal_args(f, 3) == (480.0, None)
    f0, per = evaluation._focal_args([470.0, 480.0, 490.0], 3)
    assert f0 == 470.0 and per.tolist() == [470.0, 480.0, 490.0]
    with pytest.raises(RuntimeError):
        evaluation._focal_args([470.0, 480.0], 3)
