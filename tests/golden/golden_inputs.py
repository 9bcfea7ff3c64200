"""Seeded inputs of the BASELINE-size fixtures (full_size.npz, grads.npz, losses_b16.npz).  The fixtures store only what
the REFERENCE computed from these inputs; the generator (make_golden.py) and the tests both regenerate the inputs here
(numpy PCG64 streams are identical everywhere), so a 4 MB image or a [16,3,60,90] label map is never committed.
Every fixture also stores the float64 sum of each input as a regeneration check."""
import numpy as np

from crossloc_amd import synth

FULL_H, FULL_W = 480, 720


def full_size_image(tag):
    """[1,3,480,720] uniform [0,1) frame of the 480x720 forward fixtures (tag: 'single' | 'mlr3')."""
    rng = np.random.default_rng({"single": 4801, "mlr3": 4803}[tag])
    return rng.uniform(0.0, 1.0, size=(1, 3, FULL_H, FULL_W)).astype(np.float32)


TINY_CASES = {                   # tag -> (num_mlr, extra residual blocks per side, weight seed, input shape)
    "tiny": (0, 2, 77, (2, 3, 64, 96)),
    "tiny_mlr3": (3, 1, 78, (2, 3, 64, 96)),
    "tiny_full": (0, 2, 77, (1, 3, FULL_H, FULL_W)),
}


def tiny_input(tag):
    """Input of the tiny=True fixtures (net_forward_tiny.npz): uniform [0,1) frames."""
    _, _, seed, shape = TINY_CASES[tag]
    return np.random.default_rng(seed + 100 + len(tag)).uniform(0.0, 1.0, size=shape).astype(np.float32)


GRAD_B, GRAD_H, GRAD_W, GRAD_FOCAL = 2, 64, 96, 60.0


def grad_inputs():
    """Inputs of the gradient fixture: images [2,3,64,96], camera poses above the label mean looking down (so that the
    untrained network's predictions - the mean +- a few metres - project into the image), an offset field `delta`
    [2,3,8,12]: the generator sets the labels to reference prediction + delta (stored in the fixture), with two cells NODATA."""
    rng = np.random.default_rng(6496)
    x = rng.uniform(0.0, 1.0, size=(GRAD_B, 3, GRAD_H, GRAD_W)).astype(np.float32)
    poses = np.zeros((GRAD_B, 4, 4), np.float32)
    for b in range(GRAD_B):
        T = np.eye(4)
        T[:3, :3] = synth._rot_xyz(*rng.uniform(-0.1, 0.1, size=3)) @ np.array([[1.0, 0, 0], [0, -1.0, 0], [0, 0, -1.0]])
        T[:3, 3] = synth.SCENE_MEAN + np.array([rng.uniform(-20, 20), rng.uniform(-20, 20), 120.0 + 40.0 * b])
        poses[b] = T
    delta = rng.normal(0.0, 4.0, size=(GRAD_B, 3, GRAD_H // 8, GRAD_W // 8)).astype(np.float32)
    delta[0, :, 2, 3] += np.array([70.0, 0.0, 0.0], np.float32)          # one cell beyond the 50 m init tolerance
    return x, poses, delta


def grad_full_inputs():
    """Inputs of the 480x720 gradient fixture (grads_full.npz, BASELINE configs[1] frame size, batch 1): the 'single' frame, one
    camera pose above the label mean looking down, focal length 480, an offset field `delta` [1,3,60,90] (labels = reference
    prediction + delta, a block of cells NODATA, a few cells beyond the 50 m init tolerance)."""
    rng = np.random.default_rng(480720)
    x = full_size_image("single")
    T = np.eye(4)
    T[:3, :3] = synth._rot_xyz(*rng.uniform(-0.1, 0.1, size=3)) @ np.array([[1.0, 0, 0], [0, -1.0, 0], [0, 0, -1.0]])
    T[:3, 3] = synth.SCENE_MEAN + np.array([rng.uniform(-20, 20), rng.uniform(-20, 20), 240.0])
    poses = T[None].astype(np.float32)
    delta = rng.normal(0.0, 4.0, size=(1, 3, FULL_H // 8, FULL_W // 8)).astype(np.float32)
    delta[0, :, 10:12, 20:24] += np.array([70.0, 0.0, 0.0], np.float32)[:, None, None]
    return x, poses, delta


B16 = 16


def loss_b16_inputs():
    """[16,*,60,90] inputs of the three per-pixel losses at the BASELINE configs[1] batch: synthetic scenes (ray-cast terrain),
    predictions = labels + 2 m noise with 10 % gross outliers, uncertainties exp(U(-2,3)); depth and normal likewise."""
    pred, gt, poses = synth.make_batch(300, B16, noise=2.0, outlier_ratio=0.1)
    rng = np.random.default_rng(1690)
    unc = np.exp(rng.uniform(-2.0, 3.0, size=(B16, 1, 60, 90))).astype(np.float32)
    gt_d = rng.uniform(100.0, 300.0, size=(B16, 1, 60, 90)).astype(np.float32)
    pd = (gt_d + rng.normal(0.0, 3.0, size=gt_d.shape)).astype(np.float32)
    nod = rng.uniform(size=(B16, 1, 60, 90)) < 0.02
    gt_d[nod] = -1.0
    gn = rng.normal(size=(B16, 3, 60, 90)).astype(np.float32)
    gn /= np.linalg.norm(gn, axis=1, keepdims=True)
    gn[np.broadcast_to(rng.uniform(size=(B16, 1, 60, 90)) < 0.02, gn.shape)] = -1.0
    logits = rng.normal(0.0, 2.0, size=(B16, 2, 60, 90)).astype(np.float32)
    return dict(pred=pred, gt=gt, poses=poses.astype(np.float32), unc=unc, depth_pred=pd, depth_gt=gt_d,
                normal_logits=logits, normal_gt=gn)


def checksum(a):
    return float(np.asarray(a, np.float64).sum())


def strided(a, n=256):
    """At most n elements of a flattened array, evenly strided (the gradient samples of the fixtures)."""
    flat = np.asarray(a).reshape(-1)
    step = max(1, flat.size // n)
    return flat[::step][:n].copy()
