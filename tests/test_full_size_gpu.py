"""GPU: the BASELINE configurations at their stated sizes.

configs[4]  3-encoder CrossLoc network (networks.py:421-439, 483-494) at 480x720: batch 2 against the fp32 CPU
            restatement (oracle/cnn_oracle.py, pinned by goldens generated from the reference), and the bench batches
            (24, and one past the point where the transformed tensors of the 1536-channel fusion layer exceed 2 GiB)
            through the property that a frame's result does not depend on the batch it is in.
configs[1]  batch-16 480x720 training step (forward + backward): the default plan (Winograd forward / data / weight
            gradients, conv-epilogue statistics) against the direct-convolution plan, plus additivity of the parameter
            gradients over the frames of the batch (GroupNorm is per image, so grad(B=16) = grad(first 8) + grad(last 8)).
"""
import os

import pytest

torch = pytest.importorskip("torch")

from crossloc_amd import networks, synth                      # noqa: E402
from crossloc_amd.weights import seeded_state_dict            # noqa: E402
from oracle import cnn_oracle                                 # noqa: E402

pytestmark = pytest.mark.gpu
MEAN = torch.tensor(synth.SCENE_MEAN, dtype=torch.float32)


def _images(n, seed):
    return torch.rand(n, 3, 480, 720, generator=torch.Generator().manual_seed(seed))


@pytest.mark.skipif(bool(os.environ.get("XL_NO_WINOGRAD")), reason="asserts the Winograd forms of the default plans (measurement switch set)")
def test_three_encoder_network_full_size_vs_oracle_and_across_batch_sizes():
    net = networks.TransPoseNet(MEAN, False, False, 2, 2, 3, 1, num_mlr=3)
    net.load_state_dict(seeded_state_dict(net, seed=31))
    x = _images(40, 9)
    ref = cnn_oracle.transposenet_forward(net.state_dict(), x[:2], 3, 2, 2)
    net = net.cuda().eval()
    with torch.no_grad():
        y2 = net(x[:2].cuda()).cpu()
    assert y2.shape == (2, 4, 60, 90)
    err = (y2[:, :3] - ref[:, :3]).abs().max().item()
    assert err < 1e-3 * max(1.0, (ref[:, :3] - MEAN[None, :, None, None]).abs().max().item()), err
    assert torch.allclose(y2[:, 3], ref[:, 3], rtol=2e-3)
    # the fusion layer (3x3, 1536 -> 512) must be a Winograd layer at these sizes, whatever the batch
    for B in (24, 40):
        with torch.no_grad():
            yb = net(x[:B].cuda()).cpu()
        plan = [p for k, p in net._plans.items() if k[0] == B][0]
        fusion = [op for op in plan.ops if op.type == networks.XL_OP_CONV and op.Cin == 1536 and op.nchunks2 > 1]
        assert fusion and fusion[0].nchunks2 == 64, "1536-channel fusion layer is not running as F(6x6,3x3) at batch %d" % B
        assert torch.isfinite(yb).all()
        # conv tiles straddle image boundaries differently for other batch sizes (fp64 partial sums regrouped):
        # equal to the last fp32 bit or two at |X| ~ 500 m, not bitwise
        assert torch.allclose(yb[:2, :3], y2[:, :3], rtol=0, atol=3e-4), (yb[:2, :3] - y2[:, :3]).abs().max()
        assert torch.allclose(yb[:2, 3], y2[:, 3], rtol=2e-4)


def test_batch16_training_step_at_full_size(monkeypatch):
    B = 16
    x = _images(B, 6).cuda()
    wgt = torch.randn(B, 4, 60, 90, generator=torch.Generator().manual_seed(7)).cuda()

    def run(lo=0, hi=B):
        net = networks.TransPoseNet(MEAN, False, False, 2, 2, 3, 1)
        net.load_state_dict(seeded_state_dict(net, seed=21))
        net = net.cuda().train()
        y = net(x[lo:hi])
        (y * wgt[lo:hi]).sum().backward()
        return y.detach().cpu(), {n: p.grad.detach().double().cpu() for n, p in net.named_parameters()}

    def rel(a, b):
        den = b.norm().item()
        return (a - b).norm().item() / den if den > 0 else 0.0

    y_w, g_w = run()
    assert y_w.shape == (B, 4, 60, 90) and torch.isfinite(y_w).all()
    assert len(g_w) == 114 and all(torch.isfinite(g).all() for g in g_w.values())
    # additivity over the frames of the batch
    _, g_a = run(0, 8)
    _, g_b = run(8, 16)
    # (not bitwise: the GroupNorm partial sums of the conv epilogues are grouped by conv tile, tiles straddle image
    # boundaries differently in a batch of 8, and a pre-activation within an ulp of 0 may land on the other side of ReLU)
    worst = max(rel(g_a[n] + g_b[n], g_w[n]) for n in g_w)
    assert worst < 2e-2, worst
    # against the direct-convolution lowering (two HIP paths, no CPU oracle at this size); relative L2 per tensor:
    # ReLU-mask flips between the two forward roundings move single elements, not norms
    for k in ("XL_NO_WINOGRAD_TRAIN", "XL_NO_FUSED_STATS"):
        monkeypatch.setenv(k, "1")
    y_d, g_d = run()
    assert (y_w[:, :3] - y_d[:, :3]).abs().max().item() < 5e-4
    worst = max(rel(g_w[n], g_d[n]) for n in g_d)
    assert worst < 5e-2, worst


def test_batch_invariant_single_frame_equals_the_frame_inside_a_batch_at_full_size():
    """batch_invariant plans at 480x720: a frame that runs alone (the tail batch of test_single_task) is BITWISE the frame
    inside a larger batch.  At this size the form of a deferred GroupNorm used to depend on the batch (applied on load by
    the consuming 1x1 layer for B >= 2, as a pass of its own for B = 1, with different rounding): the choice is now made
    from the layer alone and every apply site computes fmaf(x, scale, shift)."""
    net = networks.TransPoseNet(MEAN, False, False, 2, 2, 3, 1)
    net.load_state_dict(seeded_state_dict(net, seed=13))
    net.batch_invariant = True
    net = net.cuda().eval()
    x = _images(5, 3).cuda()
    with torch.no_grad():
        y5 = net(x)
        y1 = net(x[3:4])
        y2 = net(x[3:5])
    assert torch.equal(y1[0], y5[3]) and torch.equal(y2[0], y5[3]) and torch.equal(y2[1], y5[4])
    # the two plans lowered the same op types in the same order (layer-only choices)
    ops = {B: [(op.type, op.flags, op.ksize, op.Cin, op.Cout) for op in p.ops] for (B, *_), p in net._plans.items()}
    assert ops[1] == ops[2] == ops[5]
