"""Generates the golden fixtures in this directory by importing the REFERENCE Python
(/root/reference, available only in the build container) with stub modules for the packages it
imports but the hot path never uses.  Run with `python3 -B tests/golden/make_golden.py` so no
bytecode is written into the read-only reference tree.  Only inputs/outputs (data) are stored;
weights are regenerated from crossloc_amd.weights on both sides.

Reference entry points exercised:
  networks/networks.py:375-502  TransPoseNet (single-task, and num_mlr=3 "CrossLoc")
  loss/coord.py:87-188          scene_coords_regression_loss (MLE and plain)
  loss/depth.py:7-76            depth_regression_loss
  loss/normal.py:8-127          normal_regression_loss
  utils/learning.py:20-35       get_pixel_grid;  loss/coord.py:7-17 get_cam_mat
  networks/networks.py:259-273, 311-349  DenseUpsamplingConvolution / full_size_output decoder (semantics.npz)
  loss/semantics.py:10-18, 44-91         CrossEntropyLoss2d, semantics_classification_loss (semantics.npz)
Round 4 - the BASELINE sizes and the backward pass (inputs regenerated from seeds by golden_inputs.py on both sides):
  full_size.npz   networks/networks.py:466-502 at 480x720, batch 1: single-task and 3-encoder output [1,4,60,90]
  grads.npz       train_single_task.py:262-298: forward, torch.split, scene_coords_regression_loss (MLE), loss.backward()
                  on the reference module, 64x96 batch 2 - loss, rate, dL/dprediction, and for each of the 114 parameters
                  the L2 norm, the sum and a strided sample of .grad (network evaluated in float64: the fixture holds
                  no ReLU-mask flips of its own; the fp32 run's loss is stored beside it)
  grads_full.npz  (round 5) the same step at 480x720, batch 1, float64 network
  losses_b16.npz  loss/coord.py:87-188, loss/depth.py:7-76, loss/normal.py:8-127 at [16,*,60,90]: value, rate, gradient
                  moments and strided samples
"""
import builtins
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

for n in ['git', 'cv2', 'transforms3d', 'transforms3d.quaternions', 'skimage', 'skimage.io', 'skimage.color',
          'skimage.transform', 'torchvision', 'torchvision.transforms', 'dsacstar']:
    sys.modules[n] = types.ModuleType(n)
q = sys.modules['transforms3d.quaternions']; q.mat2quat = q.quat2mat = None
t = sys.modules['skimage.transform']; t.rotate = t.resize = None
sys.modules['torchvision'].transforms = sys.modules['torchvision.transforms']
torch.Tensor.cuda = lambda self, *a, **k: self
torch.nn.Module.cuda = lambda self, *a, **k: self
sys.path.insert(0, '/root/reference')
from networks.networks import TransPoseNet                      # noqa: E402
from loss.coord import scene_coords_regression_loss, get_cam_mat  # noqa: E402
from loss.depth import depth_regression_loss                    # noqa: E402
from loss.normal import normal_regression_loss                  # noqa: E402
from utils.learning import get_pixel_grid                       # noqa: E402

from crossloc_amd.weights import seeded_state_dict              # noqa: E402
sys.path.insert(0, HERE)
import golden_inputs                                            # noqa: E402

_print = builtins.print


def quiet(fn, *a, **k):
    builtins.print = lambda *x, **y: None
    try:
        return fn(*a, **k)
    finally:
        builtins.print = _print


def net_goldens():
    torch.manual_seed(0)
    mean = torch.tensor([-455.934, 417.50, 520.31])
    out = {}
    for tag, num_mlr in (("single", 0), ("mlr3", 3)):
        net = quiet(TransPoseNet, mean, False, False, 2, 2, 3, 1, 32, num_mlr, 0, False)
        net.load_state_dict(seeded_state_dict(net, seed=2021), strict=True)
        net.eval()
        rng = np.random.default_rng(11 + num_mlr)
        x = torch.from_numpy(rng.uniform(0, 1, size=(2, 3, 64, 96)).astype(np.float32))
        feats = {}
        if num_mlr == 0:
            net.encoder.register_forward_hook(lambda m, i, o: feats.__setitem__("enc", o.detach().clone()))
            net.encoder.norm1.register_forward_hook(lambda m, i, o: feats.__setitem__("norm1", o.detach().clone()))
            net.encoder.norm4.register_forward_hook(lambda m, i, o: feats.__setitem__("norm4", o.detach().clone()))
        with torch.no_grad():
            y = net(x)
        out[tag + "_x"] = x.numpy()
        out[tag + "_y"] = y.numpy()
        for k, v in feats.items():
            # strided sample + moments keep the fixture small
            out["%s_%s_sample" % (tag, k)] = v.numpy()[:, ::8, ::2, ::3].copy()
            out["%s_%s_moments" % (tag, k)] = np.array([v.double().mean().item(), v.double().abs().mean().item(),
                                                        v.double().pow(2).mean().item()])
        out[tag + "_nparams"] = np.array(sum(p.numel() for p in net.parameters()))
        out[tag + "_keys"] = np.array(["%s:%s" % (k, "x".join(map(str, v.shape))) for k, v in net.state_dict().items()])
    np.savez_compressed(os.path.join(HERE, "net_forward.npz"), **out)
    _print("net_forward.npz", {k: v.shape for k, v in out.items()})


def tiny_goldens():
    """tiny=True (test_single_task.py:43 `--tiny`, :253, :299 -> utils/evaluation.py:106 -> networks.py:133-135, 194-198, 245-247):
    128-channel residual blocks, no projection on the res2 skip path.  Single-task (2 + 2 extra blocks) and the 3-encoder fusion
    (1 + 1) at 64x96 x 2, single-task at 480x720 x 1; fp32 and float64 outputs of the reference module."""
    mean = torch.tensor([-455.934, 417.50, 520.31])
    out = {}
    for tag, (num_mlr, add, seed, shape) in golden_inputs.TINY_CASES.items():
        net = quiet(TransPoseNet, mean, True, False, add, add, 3, 1, 32, num_mlr, 0, False)
        net.load_state_dict(seeded_state_dict(net, seed=seed), strict=True)
        net.eval()
        x = torch.from_numpy(golden_inputs.tiny_input(tag))
        with torch.no_grad():
            y = net(x)
            y64 = net.double()(x.double())
        out[tag + "_x_checksum"] = np.array(golden_inputs.checksum(x.numpy()))
        out[tag + "_y"] = y.numpy()
        out[tag + "_y64"] = y64.numpy()
        out[tag + "_nparams"] = np.array(sum(p.numel() for p in net.parameters()))
        out[tag + "_keys"] = np.array(["%s:%s" % (k, "x".join(map(str, v.shape))) for k, v in net.state_dict().items()])
    np.savez_compressed(os.path.join(HERE, "net_forward_tiny.npz"), **out)
    _print("net_forward_tiny.npz", {k: v.shape for k, v in out.items()})


def net4_goldens():
    """The 4-encoder (CrossLoc-SE) variant, 1+1 extra residual blocks to keep the fixture generation short."""
    mean = torch.tensor([-455.934, 417.50, 520.31])
    net = quiet(TransPoseNet, mean, False, False, 1, 1, 3, 1, 32, 4, 0, False)
    net.load_state_dict(seeded_state_dict(net, seed=44), strict=True)
    net.eval()
    rng = np.random.default_rng(15)
    x = torch.from_numpy(rng.uniform(0, 1, size=(2, 3, 72, 104)).astype(np.float32))
    with torch.no_grad():
        y = net(x)
    out = dict(mlr4_x=x.numpy(), mlr4_y=y.numpy(),
               mlr4_nparams=np.array(sum(p.numel() for p in net.parameters())),
               mlr4_keys=np.array(["%s:%s" % (k, "x".join(map(str, v.shape))) for k, v in net.state_dict().items()]))
    np.savez_compressed(os.path.join(HERE, "net_forward_mlr4.npz"), **out)
    _print("net_forward_mlr4.npz", {k: v.shape for k, v in out.items()})


def loss_goldens():
    rng = np.random.default_rng(5)
    B, H, W = 2, 8, 12
    out = {}
    pixel_grid = get_pixel_grid(8)
    cam_mat = get_cam_mat(W * 8, H * 8, 60.0)
    # camera poses looking down at a plane; predictions near gt with planted edge cases
    from crossloc_amd import synth
    gt_coords = np.zeros((B, 3, H, W), np.float32)
    poses = np.zeros((B, 4, 4), np.float32)
    for b in range(B):
        sc = synth.make_scene(40 + b, noise=0.0, outlier_ratio=0.0, Ho=H, Wo=W, focal=60.0)
        gt_coords[b] = sc["gt_coords"]; poses[b] = sc["pose"]
    pred = gt_coords + rng.normal(0, 2.0, size=gt_coords.shape).astype(np.float32)
    gt_lab = gt_coords.copy()
    gt_lab[0, :, 0, 0] = -1.0; gt_lab[1, :, 3, 4] = -1.0; gt_lab[0, :, 7, 11] = -1.0       # nodata cells
    pred[0, :, 1, 1] = poses[0, :3, 3] + poses[0, :3, :3] @ np.array([1.0, 2.0, -30.0])   # behind the camera
    pred[0, :, 2, 2] += np.array([400.0, 0, 0], np.float32)                               # over init tolerance
    pred[1, :, 2, 5] += np.array([90.0, 0, 0], np.float32)                                # over soft clamp
    pred[1, :, 4, 7] += np.array([0.0, 49.0, 0], np.float32)
    pred[1, :, 5, 5] = gt_coords[1, :, 5, 5]                                              # exact hit
    unc = np.exp(rng.uniform(-2, 3, size=(B, 1, H, W))).astype(np.float32)
    unc[0, 0, 3, 3] = 1e-9                                                                # clamped sigma
    for mode in ("MLE", None):
        p = torch.tensor(pred, requires_grad=True); u = torch.tensor(unc, requires_grad=True)
        loss, rate = quiet(scene_coords_regression_loss, 0.1, 100.0, 1000.0, 50.0, mode, pixel_grid, -1, cam_mat,
                           p, u, torch.tensor(poses), torch.tensor(gt_lab), 'mean')
        loss.backward()
        tag = "coord_%s" % (mode or "plain")
        out[tag + "_loss"] = np.array(loss.item(), np.float64); out[tag + "_rate"] = np.array(float(rate))
        out[tag + "_dpred"] = p.grad.numpy().copy()
        out[tag + "_dunc"] = (u.grad.numpy().copy() if u.grad is not None else np.zeros_like(unc))
    # tighter clamps so the hard-clamp branch is exercised too
    p = torch.tensor(pred, requires_grad=True); u = torch.tensor(unc, requires_grad=True)
    loss, rate = quiet(scene_coords_regression_loss, 0.1, 20.0, 60.0, 50.0, "MLE", pixel_grid, -1, cam_mat,
                       p, u, torch.tensor(poses), torch.tensor(gt_lab), 'mean')
    loss.backward()
    out["coord_tight_loss"] = np.array(loss.item(), np.float64); out["coord_tight_rate"] = np.array(float(rate))
    out["coord_tight_dpred"] = p.grad.numpy().copy(); out["coord_tight_dunc"] = u.grad.numpy().copy()
    out.update(coord_pred=pred, coord_unc=unc, coord_gt=gt_lab, coord_poses=poses,
               coord_focal=np.array(60.0), coord_hw=np.array([H * 8, W * 8]))

    # depth
    gt_d = rng.uniform(100, 300, size=(B, 1, H, W)).astype(np.float32)
    pd = gt_d + rng.normal(0, 3.0, size=gt_d.shape).astype(np.float32)
    gt_dl = gt_d.copy(); gt_dl[0, 0, 0, 0] = -1.0; gt_dl[1, 0, 6, 2] = -1.0
    pd[0, 0, 1, 1] = 0.05; pd[0, 0, 2, 2] += 50.0; pd[1, 0, 3, 3] = gt_d[1, 0, 3, 3]
    for mode in ("MLE", None):
        p = torch.tensor(pd, requires_grad=True); u = torch.tensor(unc, requires_grad=True)
        loss, rate = quiet(depth_regression_loss, 0.1, 10.0, mode, -1, p, u, torch.tensor(gt_dl), 'mean')
        loss.backward()
        tag = "depth_%s" % (mode or "plain")
        out[tag + "_loss"] = np.array(loss.item(), np.float64); out[tag + "_rate"] = np.array(float(rate))
        out[tag + "_dpred"] = p.grad.numpy().copy()
        out[tag + "_dunc"] = (u.grad.numpy().copy() if u.grad is not None else np.zeros_like(unc))
    out.update(depth_pred=pd, depth_gt=gt_dl)

    # normal
    gn = rng.normal(size=(B, 3, H, W)).astype(np.float32)
    gn /= np.linalg.norm(gn, axis=1, keepdims=True)
    gn_l = gn.copy(); gn_l[0, :, 0, 0] = -1.0; gn_l[1, :, 2, 9] = -1.0
    logits = rng.normal(0, 2.0, size=(B, 2, H, W)).astype(np.float32)
    logits[0, 0, 1, 1] = 40.0; logits[0, 1, 1, 2] = -40.0                                  # saturated sigmoid
    for mode in ("MLE", None):
        p = torch.tensor(logits, requires_grad=True); u = torch.tensor(unc, requires_grad=True)
        loss, rate = quiet(normal_regression_loss, 10.0, mode, -1, p, u, torch.tensor(gn_l), 'mean')
        loss.backward()
        tag = "normal_%s" % (mode or "plain")
        out[tag + "_loss"] = np.array(loss.item(), np.float64); out[tag + "_rate"] = np.array(float(rate))
        out[tag + "_dpred"] = p.grad.numpy().copy()
        out[tag + "_dunc"] = (u.grad.numpy().copy() if u.grad is not None else np.zeros_like(unc))
    out.update(normal_logits=logits, normal_gt=gn_l)
    np.savez_compressed(os.path.join(HERE, "losses.npz"), **out)
    _print("losses.npz", sorted(out.keys()))


def semantics_goldens():
    """Full-size (semantics) head: 6 classes, no uncertainty channel (utils/evaluation.py:92-108)."""
    from loss.semantics import semantics_classification_loss, CrossEntropyLoss2d
    out = {}
    net = quiet(TransPoseNet, torch.zeros(6), False, False, 2, 2, 6, 0, 32, 0, 0, True)
    net.load_state_dict(seeded_state_dict(net, seed=2021), strict=True)
    net.eval()
    for tag, hw in (("sem", (64, 96)), ("sem_resize", (60, 92))):      # 60x92 -> 8x12 -> 64x96 -> bilinear to 60x92
        rng = np.random.default_rng(31 + hw[0])
        x = torch.from_numpy(rng.uniform(0, 1, size=(2, 3) + hw).astype(np.float32))
        with torch.no_grad():
            y = net(x)
        out[tag + "_x"] = x.numpy(); out[tag + "_y"] = y.numpy()
    out["sem_keys"] = np.array(["%s:%s" % (k, "x".join(map(str, v.shape))) for k, v in net.state_dict().items()])
    rng = np.random.default_rng(9)
    B, C, H, W = 2, 6, 16, 24
    logits = rng.normal(0, 2.0, size=(B, C, H, W)).astype(np.float32)
    labels = rng.integers(0, C, size=(B, 1, H, W)).astype(np.float32)
    logits[0, :, 0, 0] = 0.0                                        # tie: argmax takes the first class
    logits[1, 3, 2, 2] = 60.0                                       # saturated soft-max
    for red in ("mean", None):
        p = torch.tensor(logits, requires_grad=True)
        loss, rate = quiet(semantics_classification_loss, None, p, None, torch.tensor(labels), CrossEntropyLoss2d(), red)
        loss.sum().backward()
        tag = "ce_%s" % (red or "none")
        out[tag + "_loss"] = np.atleast_1d(loss.detach().numpy()).astype(np.float64)
        out[tag + "_rate"] = np.array(float(rate))
        out[tag + "_dlogits"] = p.grad.numpy().copy()
    out.update(ce_logits=logits, ce_labels=labels)
    np.savez_compressed(os.path.join(HERE, "semantics.npz"), **out)
    _print("semantics.npz", {k: v.shape for k, v in out.items()})


def full_size_goldens():
    """BASELINE configs[2] / [4] frame size: the reference forward at 480x720, batch 1 (networks/networks.py:466-502)."""
    mean = torch.tensor([-455.934, 417.50, 520.31])
    out = {}
    for tag, num_mlr in (("single", 0), ("mlr3", 3)):
        net = quiet(TransPoseNet, mean, False, False, 2, 2, 3, 1, 32, num_mlr, 0, False)
        net.load_state_dict(seeded_state_dict(net, seed=2021), strict=True)
        net.eval()
        x = golden_inputs.full_size_image(tag)
        with torch.no_grad():
            y = net(torch.from_numpy(x))
            # the same module evaluated in float64 (round 5): how far the reference's OWN fp32 forward is from it is the yardstick
            # the HIP forward - fp16-pair GEMMs, Winograd - is held to (tests/test_reference_fixtures.py)
            y64 = net.to(torch.float64)(torch.from_numpy(x).to(torch.float64))
        out[tag + "_y"] = y.numpy()
        out[tag + "_y64"] = y64.numpy()
        out[tag + "_x_checksum"] = np.array(golden_inputs.checksum(x))
    np.savez_compressed(os.path.join(HERE, "full_size.npz"), **out)
    _print("full_size.npz", {k: v.shape for k, v in out.items()})


def grad_goldens():
    """One training step's forward + loss + backward on the reference module (train_single_task.py:262-298), 64x96, batch 2."""
    mean = torch.tensor([-455.934, 417.50, 520.31])
    x, poses, delta = golden_inputs.grad_inputs()
    H, W = golden_inputs.GRAD_H, golden_inputs.GRAD_W
    pixel_grid = get_pixel_grid(8)
    cam_mat = get_cam_mat(W, H, golden_inputs.GRAD_FOCAL)
    out = {}
    gt = None
    for dtype, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
        net = quiet(TransPoseNet, mean, False, False, 2, 2, 3, 1, 32, 0, 0, False)
        net.load_state_dict(seeded_state_dict(net, seed=2021), strict=True)
        net = net.to(dtype).train()
        pred = net(torch.from_numpy(x).to(dtype))
        if gt is None:
            # labels = the reference's own (float64) prediction + the seeded offset field, two cells without data
            gt = (pred.detach()[:, :3].float() + torch.from_numpy(delta)).numpy()
            gt[0, :, 0, 0] = -1.0
            gt[1, :, 5, 7] = -1.0
        pred32 = pred.float()                       # the loss runs in fp32 like train_single_task.py; autograd carries the
        pred32.retain_grad()                        # gradient back through the cast into the float64 network
        sc, unc = torch.split(pred32, [3, 1], dim=1)                                       # train_single_task.py:269
        loss, rate = quiet(scene_coords_regression_loss, 0.1, 100.0, 1000.0, 50.0, "MLE", pixel_grid, -1, cam_mat,
                           sc, unc, torch.from_numpy(poses), torch.from_numpy(gt), 'mean')
        loss.backward()
        out["loss_" + tag] = np.array(loss.item(), np.float64)
        out["rate_" + tag] = np.array(float(rate))
        if tag == "f32":
            # round 5: the reference's OWN fp32 gradients (sample + norm per tensor): how far fp32 arithmetic sits from the
            # float64 result on this graph - the yardstick the tests hold the HIP path to
            out["param_grad_sample_f32"] = np.stack([np.pad(golden_inputs.strided(p.grad.double().numpy()), (0, 256 - golden_inputs.strided(p.grad.numpy()).size))
                                                     for _, p in net.named_parameters()]).astype(np.float32)
            out["param_grad_l2_f32"] = np.array([float(np.sqrt((p.grad.double().numpy() ** 2).sum())) for _, p in net.named_parameters()])
            continue
        out["y"] = pred.detach().float().numpy()
        out["dpred"] = pred32.grad.numpy().copy()
        names, norms, sums, samples = [], [], [], []
        for name, p in net.named_parameters():
            g = p.grad.double().numpy()
            names.append("%s:%s" % (name, "x".join(map(str, p.shape))))
            norms.append(np.sqrt((g * g).sum()))
            sums.append(g.sum())
            s = golden_inputs.strided(g)
            samples.append(np.pad(s, (0, 256 - s.size)))
        out["param_names"] = np.array(names)
        out["param_grad_l2"] = np.array(norms)
        out["param_grad_sum"] = np.array(sums)
        out["param_grad_sample"] = np.stack(samples).astype(np.float32)
    out.update(gt=gt, x_checksum=np.array(golden_inputs.checksum(x)), poses_checksum=np.array(golden_inputs.checksum(poses)))
    np.savez_compressed(os.path.join(HERE, "grads.npz"), **out)
    _print("grads.npz", {k: v.shape for k, v in out.items()}, "loss f64-net %.6f fp32 %.6f rate %.4f" % (
        out["loss_f64"], out["loss_f32"], out["rate_f64"]))


def grad_full_goldens():
    """Round 5: the same training step (train_single_task.py:262-298) on the reference module at the BASELINE frame size,
    480x720, batch 1, network in float64 (no ReLU-mask flips of its own): loss, rate, dL/dprediction and for each of the 114
    parameters the L2 norm, the sum and a strided 256-element sample of .grad."""
    mean = torch.tensor([-455.934, 417.50, 520.31])
    x, poses, delta = golden_inputs.grad_full_inputs()
    pixel_grid = get_pixel_grid(8)
    cam_mat = get_cam_mat(golden_inputs.FULL_W, golden_inputs.FULL_H, 480.0)
    net = quiet(TransPoseNet, mean, False, False, 2, 2, 3, 1, 32, 0, 0, False)
    net.load_state_dict(seeded_state_dict(net, seed=2021), strict=True)
    net = net.to(torch.float64).train()
    pred = net(torch.from_numpy(x).to(torch.float64))
    gt = (pred.detach()[:, :3].float() + torch.from_numpy(delta)).numpy()
    gt[0, :, 30:33, 40:45] = -1.0
    pred32 = pred.float()
    pred32.retain_grad()
    sc, unc = torch.split(pred32, [3, 1], dim=1)
    loss, rate = quiet(scene_coords_regression_loss, 0.1, 100.0, 1000.0, 50.0, "MLE", pixel_grid, -1, cam_mat,
                       sc, unc, torch.from_numpy(poses), torch.from_numpy(gt), 'mean')
    loss.backward()
    out = dict(loss_f64=np.array(loss.item(), np.float64), rate_f64=np.array(float(rate)), y=pred.detach().float().numpy(),
               dpred=pred32.grad.numpy().copy(), gt=gt)
    names, norms, sums, samples = [], [], [], []
    for name, p in net.named_parameters():
        g = p.grad.double().numpy()
        names.append("%s:%s" % (name, "x".join(map(str, p.shape))))
        norms.append(np.sqrt((g * g).sum()))
        sums.append(g.sum())
        s = golden_inputs.strided(g)
        samples.append(np.pad(s, (0, 256 - s.size)))
    out.update(param_names=np.array(names), param_grad_l2=np.array(norms), param_grad_sum=np.array(sums),
               param_grad_sample=np.stack(samples).astype(np.float32),
               x_checksum=np.array(golden_inputs.checksum(x)), poses_checksum=np.array(golden_inputs.checksum(poses)))
    # the reference's own fp32 run of the same step (same labels): its distance from the float64 gradients is the yardstick
    net32 = quiet(TransPoseNet, mean, False, False, 2, 2, 3, 1, 32, 0, 0, False)
    net32.load_state_dict(seeded_state_dict(net32, seed=2021), strict=True)
    net32 = net32.train()
    p32 = net32(torch.from_numpy(x))
    sc, unc = torch.split(p32, [3, 1], dim=1)
    loss32, _ = quiet(scene_coords_regression_loss, 0.1, 100.0, 1000.0, 50.0, "MLE", pixel_grid, -1, cam_mat,
                      sc, unc, torch.from_numpy(poses), torch.from_numpy(gt), 'mean')
    loss32.backward()
    out["loss_f32"] = np.array(loss32.item(), np.float64)
    out["param_grad_sample_f32"] = np.stack([np.pad(golden_inputs.strided(p.grad.double().numpy()), (0, 256 - golden_inputs.strided(p.grad.numpy()).size))
                                             for _, p in net32.named_parameters()]).astype(np.float32)
    out["param_grad_l2_f32"] = np.array([float(np.sqrt((p.grad.double().numpy() ** 2).sum())) for _, p in net32.named_parameters()])
    np.savez_compressed(os.path.join(HERE, "grads_full.npz"), **out)
    _print("grads_full.npz", {k: v.shape for k, v in out.items()}, "loss %.6f rate %.4f" % (out["loss_f64"], out["rate_f64"]))


def loss_b16_goldens():
    """The three per-pixel losses at the BASELINE configs[1] batch, [16,*,60,90] (MLE and plain)."""
    I = golden_inputs.loss_b16_inputs()
    pixel_grid = get_pixel_grid(8)
    cam_mat = get_cam_mat(720, 480, 480.0)
    out = {}

    def record(tag, loss, rate, p, u, f64=False):
        loss.backward()
        if not f64:
            out[tag + "_loss"] = np.array(loss.item(), np.float64)
            out[tag + "_rate"] = np.array(float(rate))
        for nm, t in (("dpred", p), ("dunc", u)):
            g = t.grad.double().numpy() if t.grad is not None else np.zeros(tuple(t.shape))
            if f64:
                # the same reference functions evaluated in float64: how far the reference's OWN fp32 result is from the
                # exact gradient, element by element (fp32 cancellation in |PX - PX_gt| at |X| ~ 500 m) - the tests allow
                # the HIP path that band on top of their tolerance
                out["%s_%s_sample64" % (tag, nm)] = g[:, :, ::4, ::5].astype(np.float32)
                continue
            out["%s_%s_moments" % (tag, nm)] = np.array([g.sum(), np.abs(g).sum(), np.sqrt((g * g).sum())])
            out["%s_%s_sample" % (tag, nm)] = g[:, :, ::4, ::5].astype(np.float32)

    for dt in (torch.float32, torch.float64):
        f64 = dt == torch.float64
        torch.set_default_dtype(dt)
        pg, cm = pixel_grid.to(dt), cam_mat.to(dt)

        def T(name, grad=False):
            return torch.tensor(I[name], dtype=dt, requires_grad=grad)
        for mode in ("MLE", None):
            p, u = T("pred", True), T("unc", True)
            loss, rate = quiet(scene_coords_regression_loss, 0.1, 100.0, 1000.0, 50.0, mode, pg, -1, cm,
                               p, u, T("poses"), T("gt"), 'mean')
            record("coord_%s" % (mode or "plain"), loss, rate, p, u, f64)
            p, u = T("depth_pred", True), T("unc", True)
            loss, rate = quiet(depth_regression_loss, 0.1, 10.0, mode, -1, p, u, T("depth_gt"), 'mean')
            record("depth_%s" % (mode or "plain"), loss, rate, p, u, f64)
            p, u = T("normal_logits", True), T("unc", True)
            loss, rate = quiet(normal_regression_loss, 10.0, mode, -1, p, u, T("normal_gt"), 'mean')
            record("normal_%s" % (mode or "plain"), loss, rate, p, u, f64)
    torch.set_default_dtype(torch.float32)
    out["input_checksums"] = np.array([golden_inputs.checksum(I[k]) for k in sorted(I)])
    np.savez_compressed(os.path.join(HERE, "losses_b16.npz"), **out)
    _print("losses_b16.npz", {k: (v.shape if v.ndim else float(v)) for k, v in out.items()})


if __name__ == "__main__":
    which = sys.argv[1:] or ["net", "net4", "loss", "semantics", "full", "grads", "loss16", "gradsfull", "tiny"]
    if "tiny" in which:
        tiny_goldens()
    if "gradsfull" in which:
        grad_full_goldens()
    if "net" in which:
        net_goldens()
    if "net4" in which:
        net4_goldens()
    if "loss" in which:
        loss_goldens()
    if "semantics" in which:
        semantics_goldens()
    if "full" in which:
        full_size_goldens()
    if "grads" in which:
        grad_goldens()
    if "loss16" in which:
        loss_b16_goldens()
