"""GPU: parity on the plan bench.py TIMES (round 6; /root/reference/test_single_task.py:347-363 is the loop it stands for).

The headline number comes from ONE step shape: 95 frames of 480x720 through the single-task network with 2 + 2 extra residual
blocks (7168 tiles of the batched Winograd GEMMs, tiles straddling image boundaries, per-range stem launches), then the planted
scene coordinates through `dsacstar.forward_rgb_batch` at 256 hypotheses in the launch form 95 frames select.  The reference
fixtures are otherwise checked at batch 1 and the solver's bit-exactness at B = 6, so this file repeats both checks INSIDE the
benchmarked step:

  * the fixture frame of tests/golden/full_size.npz (`golden_inputs.full_size_image("single")`) sits in slots 0, 47 and 94 of a
    batch of random frames; those three outputs are held to the reference module's fp32 output at the fixture tolerance and to
    its float64 output at the yardstick of test_reference_fixtures.py (8x / 12x the reference's own fp32 distance on sigma);
  * the planted coordinates go through the solver with debug outputs on, and frames 0, 1, 23, 24, 46, 47, 48, 70, 71, 93, 94
    are compared BITWISE with oracle/dsac_oracle.c (cells, tries, fp64 scores, winner, refinement counts, both poses);
  * near-tie robustness: the planted coordinates perturbed by the forward's measured distance from float64 (4e-5 m, uniform
    +-) move no pose by more than 1 mm / 0.001 deg on 64 frames - what the sigma-channel distance of weak item 2 does to a pose.
"""
import ctypes
import os
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import golden_inputs                                            # noqa: E402

from crossloc_amd import _lib, evaluation, networks, synth      # noqa: E402
from crossloc_amd.weights import seeded_state_dict              # noqa: E402

pytestmark = pytest.mark.gpu
FULL = np.load(os.path.join(HERE, "golden", "full_size.npz"))
B, H, W, NH = 95, 480, 720, 256
SLOTS = (0, 47, 94)


def _bench_network():
    """bench.py:main - the network the headline is quoted on (utils/learning.py:302-305 sizes, seed 2021 = the fixtures' weights)."""
    mean = torch.tensor(synth.SCENE_MEAN, dtype=torch.float32)
    net = networks.TransPoseNet(mean, False, False, 2, 2, 3, 1)
    net.load_state_dict(seeded_state_dict(net, seed=2021))
    return net.cuda().eval(), mean


def test_the_benchmarked_step_against_the_reference_fixture_and_the_solver_oracle(oracle):
    import dsacstar
    from test_dsac_gpu import _assert_same
    net, mean = _bench_network()
    g = torch.Generator(device="cpu").manual_seed(2021)
    images = torch.rand((B, 3, H, W), generator=g)
    frame = torch.from_numpy(golden_inputs.full_size_image("single"))[0]
    for s in SLOTS:
        images[s] = frame
    images = images.cuda()
    coords_np, _, poses_np = synth.make_batch(2021, B, noise=0.5, outlier_ratio=0.3)
    coords = torch.from_numpy(coords_np).cuda()

    pipe = evaluation.PipelinedLocalizer(net, NH, synth.FOCAL, H, W)
    pred, events = pipe.forward_cnn(images, plant=None)
    torch.cuda.synchronize()
    # ---- the plan is the one bench.py times: 95 frames, pair GEMMs, 64 batched F(6x6,3x3) products of 7168 tiles
    plan = net._plans[(B, H, W, torch.cuda.current_device(), False, 1)]
    wino = [op for op in plan.ops if op.type == networks.XL_OP_CONV and op.nchunks2 == 64 and op.Cin == 512 and op.Cout == 512]
    if not any(os.environ.get(k) for k in ("XL_NO_WINOGRAD", "XL_WINOGRAD")):
        assert len(wino) == 9 and all(op.B * op.Ho * op.Wo == B * 150 for op in wino)
        assert (-(-B * 150 // 256)) * 2 * 64 == 7168
        if os.environ.get("XL_GEMM_PAIR", "1") not in ("", "0") and os.environ.get("XL_GEMM_SPLIT_BF16", "il") not in ("", "0"):
            assert all(op.flags & networks.CONV_PAIR_F16 for op in wino)
    # ---- the fixture frame in three slots of the batch against the reference module's own outputs
    y = pred.cpu().double()
    ref32 = torch.from_numpy(FULL["single_y"]).double()
    y64 = FULL["single_y64"]
    m = mean[None, :, None, None].double()
    rng = (ref32[:, :3] - m).abs().max().item()
    e32 = np.abs(FULL["single_y"].astype(np.float64) - y64)
    r32 = e32[:, 3] / np.abs(y64[:, 3])
    for s in SLOTS:
        ys = y[s:s + 1]
        err = (ys[:, :3] - ref32[:, :3]).abs().max().item()
        assert err <= 2e-4 * max(1.0, rng), (s, err, rng)
        assert torch.allclose(ys[:, 3], ref32[:, 3], rtol=2e-3), s
        e = np.abs(ys.numpy() - y64)
        assert e[:, :3].mean() <= 1.25 * e32[:, :3].mean() and e[:, :3].max() <= 2.0 * e32[:, :3].max(), (s, e[:, :3].mean(), e[:, :3].max())
        r = e[:, 3] / np.abs(y64[:, 3])
        print("slot %d of 95: coordinates %.2e m from the reference; sigma vs float64 median %.2e (reference fp32 %.2e), max %.2e (%.2e)"
              % (s, err, np.median(r), np.median(r32), r.max(), r32.max()))
        assert np.median(r) <= 8.0 * np.median(r32) and r.max() <= 12.0 * r32.max(), (s, np.median(r), r.max())
    # (a frame's result depends on its slot only through the grouping of the fp64 GroupNorm partial sums: last fp32 bits)
    assert torch.allclose(y[0, :3], y[47, :3], rtol=0, atol=3e-4) and torch.allclose(y[0, 3], y[94, 3], rtol=2e-4)

    # ---- the solver on the network's own output tensor (strided NCHW view), coordinates planted, as bench.py's step does
    L = _lib.lib()
    L.xl_dsac_forward_sub_blocks.argtypes = [ctypes.c_int, ctypes.c_int]
    L.xl_dsac_forward_sub_blocks.restype = ctypes.c_int
    S95, S6 = L.xl_dsac_forward_sub_blocks(B, NH), L.xl_dsac_forward_sub_blocks(6, NH)
    if not any(os.environ.get(k) for k in ("XL_DSAC_NO_SPLIT", "XL_DSAC_SPLIT")):
        assert S95 > 1 and S95 != S6, "95 frames are expected to select a split launch form of their own (not the B = 6 tests')"
    pred[:, :3].copy_(coords)
    poses = torch.zeros((B, 4, 4), dtype=torch.float32, device="cuda")
    image0 = 3 * B                                                       # a later step of the bench: global image indices 285 ..
    d = dsacstar.forward_rgb_batch(pred[:, :3], poses, NH, 10.0, synth.FOCAL, W / 2.0, H / 2.0, 100.0, 100.0, 8,
                                   image0=image0, debug=True)
    torch.cuda.synchronize()
    pose_g = poses.cpu().numpy()
    dg = {k: v.cpu().numpy() for k, v in d.items()}
    for b in (0, 1, 23, 24, 46, 47, 48, 70, 71, 93, 94):
        _assert_same(oracle, coords_np[b], pose_g, dg, b, NH, image=image0 + b, focal=synth.FOCAL, ppx=W / 2.0, ppy=H / 2.0)
    errs = np.array([synth.pose_error(poses_np[b], pose_g[b]) for b in range(B)])
    assert np.median(errs[:, 0]) < 0.15 and np.median(errs[:, 1]) < 0.03


def test_a_pose_does_not_move_under_the_forward_distance_from_float64():
    """Weak item 2 of round 5 as a number: the HIP forward sits up to 2e-5 (relative, sigma) / 4e-5 m (coordinates at |X| ~ 500 m:
    one fp32 ulp is 6e-5 m) from the float64 evaluation of the reference.  Perturbing the planted scene coordinates of 64 frames by
    uniform +-4e-5 m must leave every pose within 1 mm / 0.001 deg of the unperturbed one, near-ties of the hypothesis
    selection included (256 hypotheses, 30 % gross outliers)."""
    import dsacstar
    n = 64
    coords_np, _, _ = synth.make_batch(7100, n, noise=0.5, outlier_ratio=0.3)
    rng = np.random.default_rng(71)
    pert = (coords_np + rng.uniform(-4e-5, 4e-5, size=coords_np.shape)).astype(np.float32)
    # (no-data cells stay no-data)
    pert = np.where(np.broadcast_to(np.all(coords_np == -1.0, axis=1, keepdims=True), pert.shape), coords_np, pert)
    assert (pert != coords_np).mean() > 0.3                                 # the perturbation survives the fp32 rounding at 500 m
    out = []
    for c in (coords_np, pert):
        poses = torch.zeros((n, 4, 4), dtype=torch.float32, device="cuda")
        dsacstar.forward_rgb_batch(torch.from_numpy(c).cuda(), poses, NH, 10.0, synth.FOCAL, W / 2.0, H / 2.0, 100.0, 100.0, 8, image0=500)
        torch.cuda.synchronize()
        out.append(poses.cpu().numpy())
    moved = np.array([synth.pose_error(out[0][b], out[1][b]) for b in range(n)])
    print("pose movement under a +-4e-5 m coordinate perturbation, 64 frames: max %.3e m, %.3e deg; median %.3e m, %.3e deg"
          % (moved[:, 0].max(), moved[:, 1].max(), np.median(moved[:, 0]), np.median(moved[:, 1])))
    assert moved[:, 0].max() < 1e-3 and moved[:, 1].max() < 1e-3
