"""GPU: the CNN -> solver hand-off of test_single_task.py:347-363 executed as ONE chain — the solver consumes the
network's own output tensor (the strided NCHW `pred[:, :3]` view, sigma channel dropped like :354), produced on a CNN
stream and read on the solver's side stream, ordered by events only.

Untrained weights do not predict a scene (no trained weights exist offline), so two plants make the chain checkable:
  * `plant=`: the synthetic scene is written into the coordinate channels of the output tensor after the head (on the
    CNN stream); everything downstream is the product path of `--solver_input network`;
  * a planted decoder (fc3.weight = 0, fc3.bias = c): the network itself emits a known constant map; the solver's
    degenerate-input behaviour (bounded retries, identity pose, dsacstar_util.h:114-116) is compared with the oracle.
Poses are compared bit-for-bit with the CPU oracle run on the very floats the network tensor holds."""
import os
import subprocess
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from crossloc_amd import evaluation, networks, synth          # noqa: E402
from crossloc_amd.weights import seeded_state_dict            # noqa: E402

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MEAN = torch.tensor(synth.SCENE_MEAN, dtype=torch.float32)


def _net(enc=2, dec=2, seed=5):
    net = networks.TransPoseNet(MEAN, False, False, enc, dec, 3, 1)
    net.load_state_dict(seeded_state_dict(net, seed=seed))
    return net.cuda().eval()


@pytest.mark.parametrize("cnn_streams", [1, 2])
def test_solver_consumes_network_output_across_streams(oracle, cnn_streams):
    B, H, W, NH = 5, 480, 720, 64
    net = _net()
    coords, _, poses = synth.make_batch(700, B, noise=0.5, outlier_ratio=0.3)
    coords_t = torch.from_numpy(coords).cuda()
    images = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(3)).cuda()
    pipe = evaluation.PipelinedLocalizer(net, NH, synth.FOCAL, H, W, cnn_streams=cnn_streams)
    runs = []
    for rep in range(3):                                   # back-to-back batches: solver(s) overlaps CNN(s+1)
        runs.append(pipe.submit(images, image0=11, plant=coords_t))
    pipe.finish()
    torch.cuda.synchronize()
    with torch.no_grad():
        plain = net(images)                                # the same network without the plant: sigma must be the CNN's
    for est, pred in runs:
        assert pred.shape == (B, 4, 60, 90) and not pred[:, :3].is_contiguous()
        assert torch.equal(pred[:, :3], coords_t)
        # stream split (2 sub-batches) changes conv tile composition: equal to the last bits, not bitwise
        assert torch.allclose(pred[:, 3], plain[:, 3], rtol=1e-4) and (pred[:, 3] > 0).all()
        held = pred[:, :3].cpu().numpy()                   # the floats the solver actually read
        got = est.cpu().numpy()
        for b in range(B):
            ref = oracle.forward_rgb(np.ascontiguousarray(held[b]), NH, 10.0, synth.FOCAL, W / 2, H / 2, 100.0, 100.0, 8,
                                     image=11 + b)
            assert np.array_equal(got[b], ref), "frame %d: chained pose differs from the oracle" % b
            t_err, r_err = synth.pose_error(poses[b], got[b])
            assert t_err < 0.5 and r_err < 0.1
    # the side-input form (scene_coords=) gives the identical poses: same floats, different tensor
    est2, _ = evaluation.localize_batch(net, images, NH, synth.FOCAL, H, W, image0=11, scene_coords=coords_t)
    torch.cuda.synchronize()
    assert torch.equal(est2, runs[0][0])


def test_planted_decoder_network_output_feeds_the_solver(oracle):
    """`--solver_input network` with a decoder whose last layer is planted: fc3.weight = 0, fc3.bias = c  =>  the
    network emits mean + c in every cell and sigma = exp(c3).  A constant map is a degenerate scene: every P3P try fails,
    the retry budget is exhausted and the pose is the identity — same bits as the oracle on the emitted floats."""
    B, H, W, NH = 2, 64, 96, 16
    net = _net(1, 1)
    with torch.no_grad():
        net.decoder.fc3.weight.zero_()
        net.decoder.fc3.bias.copy_(torch.tensor([1.5, -2.25, 3.0, 0.5]))
    images = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(4)).cuda()
    import dsacstar
    with torch.no_grad():
        pred = net(images)
    want = MEAN + torch.tensor([1.5, -2.25, 3.0])
    assert torch.equal(pred[:, :3].cpu(), want[None, :, None, None].expand(B, 3, H // 8, W // 8))
    assert torch.allclose(pred[:, 3].cpu(), torch.full((B, H // 8, W // 8), float(np.exp(0.5))), rtol=1e-6)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    out = torch.zeros(B, 4, 4, device="cuda")
    with torch.cuda.stream(side):
        dbg = dsacstar.forward_rgb_batch(pred[:, :3], out, NH, 10.0, 60.0, W / 2, H / 2, 100.0, 100.0, 8, image0=3,
                                         max_tries=200, debug=True)
    side.synchronize()
    held = pred[:, :3].cpu().numpy()
    for b in range(B):
        ref, rd = oracle.forward_rgb(np.ascontiguousarray(held[b]), NH, 10.0, 60.0, W / 2, H / 2, 100.0, 100.0, 8,
                                     image=3 + b, max_tries=200, debug=True)
        assert np.array_equal(out[b].cpu().numpy(), ref)
        assert np.array_equal(dbg["tries"][b].cpu().numpy(), rd["tries"])
        assert np.isfinite(ref).all()


def test_localize_batch_takes_one_focal_per_frame(oracle):
    """Frames of one batch may come from different cameras (the dataset computes the focal length per frame)."""
    B, NH = 3, 64
    focals = [480.0, 520.0, 440.0]
    scenes = [synth.make_scene(900 + b, noise=0.3, outlier_ratio=0.2, focal=focals[b]) for b in range(B)]
    coords = np.stack([s["coords"] for s in scenes])
    net = _net(0, 0)
    images = torch.rand(B, 3, 480, 720, generator=torch.Generator().manual_seed(5)).cuda()
    est, _ = evaluation.localize_batch(net, images, NH, focals, 480, 720, image0=0, plant=torch.from_numpy(coords).cuda())
    torch.cuda.synchronize()
    for b in range(B):
        ref = oracle.forward_rgb(coords[b], NH, 10.0, focals[b], 360.0, 240.0, 100.0, 100.0, 8, image=b)
        assert np.array_equal(est[b].cpu().numpy(), ref)
        t_err, r_err = synth.pose_error(scenes[b]["pose"], ref)
        assert t_err < 0.5 and r_err < 0.1


def test_single_task_entry_point_with_the_network_output_as_solver_input():
    out = subprocess.check_output([sys.executable, "-m", "crossloc_amd.test_single_task", "--synthetic", "6",
                                   "--batch", "3", "--hypotheses", "64", "--solver_input", "planted"],
                                  cwd=ROOT, stderr=subprocess.STDOUT).decode()
    assert "Localised 6 frames" in out and "5m5deg: 100.0%" in out
