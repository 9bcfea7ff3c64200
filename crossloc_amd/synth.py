"""Synthetic scenes for parity tests and the benchmark.

No datasets or trained weights exist offline, so every workload is generated: a sinusoidal terrain
near the naturescape label mean (utils/learning.py:92-94 of the reference), nadir-ish pinhole
cameras 150-350 m above it, scene-coordinate maps obtained by casting the centre of every output
cell (8x+4, 8y+4) onto the terrain, plus Gaussian noise and a fraction of gross outliers.
Everything is a pure function of the integer seed (numpy PCG64).
"""
import numpy as np

SCENE_MEAN = np.array([-455.934, 417.50, 520.31])   # naturescape coord mean, utils/learning.py:92
FOCAL = 480.0
IMAGE_H, IMAGE_W, SUBSAMPLE = 480, 720, 8
NODATA = -1.0


def terrain_height(x, y):
    """Height field (metres) around SCENE_MEAN[2]; x, y are world coordinates."""
    xr, yr = x - SCENE_MEAN[0], y - SCENE_MEAN[1]
    return SCENE_MEAN[2] - 240.0 + 40.0 * np.sin(xr / 90.0) * np.cos(yr / 70.0) + 15.0 * np.sin(xr / 23.0 + yr / 31.0)


def _rot_xyz(rx, ry, rz):
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def random_pose(rng, max_tilt_deg=25.0):
    """cam->world 4x4 (float64). Camera z looks down, x east, y south, then roll/pitch/yaw jitter."""
    nadir = np.array([[1.0, 0, 0], [0, -1.0, 0], [0, 0, -1.0]])   # columns: camera axes in world
    tilt = np.deg2rad(max_tilt_deg)
    R = _rot_xyz(*(rng.uniform(-tilt, tilt, size=3))) @ nadir
    cx = SCENE_MEAN[0] + rng.uniform(-300, 300)
    cy = SCENE_MEAN[1] + rng.uniform(-300, 300)
    cz = terrain_height(cx, cy) + rng.uniform(150.0, 350.0)
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = [cx, cy, cz]
    return T


def raycast(pose, Ho, Wo, sub=SUBSAMPLE, focal=FOCAL, ppx=IMAGE_W / 2, ppy=IMAGE_H / 2):
    """World points hit by the centre ray of every output cell -> [3, Ho, Wo] float64, hit mask."""
    xs = np.arange(Wo) * sub + sub // 2
    ys = np.arange(Ho) * sub + sub // 2
    u, v = np.meshgrid(xs, ys)
    d_cam = np.stack([(u - ppx) / focal, (v - ppy) / focal, np.ones_like(u, float)], 0).reshape(3, -1)
    d = pose[:3, :3] @ d_cam
    o = pose[:3, 3:4]
    hit = d[2] < -1e-3
    dz = np.where(hit, d[2], -1.0)
    lo = (o[2] - (SCENE_MEAN[2] - 240.0 + 56.0)) / -dz
    hi = (o[2] - (SCENE_MEAN[2] - 240.0 - 56.0)) / -dz
    for _ in range(48):
        mid = 0.5 * (lo + hi)
        p = o + d * mid
        above = p[2] > terrain_height(p[0], p[1])
        lo = np.where(above, mid, lo)
        hi = np.where(above, hi, mid)
    p = o + d * (0.5 * (lo + hi))
    return p.reshape(3, Ho, Wo), hit.reshape(Ho, Wo)


def make_scene(seed, noise=0.5, outlier_ratio=0.3, Ho=IMAGE_H // SUBSAMPLE, Wo=IMAGE_W // SUBSAMPLE,
               sub=SUBSAMPLE, focal=FOCAL):
    """One frame: returns dict(coords[3,Ho,Wo] f32 predicted map, gt_coords[3,Ho,Wo] f32 with NODATA,
    pose[4,4] f64 cam->world, focal)."""
    rng = np.random.default_rng(seed)
    ppx, ppy = Wo * sub / 2.0, Ho * sub / 2.0
    pose = random_pose(rng)
    gt, hit = raycast(pose, Ho, Wo, sub, focal, ppx, ppy)
    pred = gt + rng.normal(0.0, noise, size=gt.shape) if noise > 0 else gt.copy()
    n_out = int(round(outlier_ratio * Ho * Wo))
    if n_out > 0:
        idx = rng.choice(Ho * Wo, size=n_out, replace=False)
        lo = np.array([SCENE_MEAN[0] - 750, SCENE_MEAN[1] - 750, SCENE_MEAN[2] - 300])
        hi = np.array([SCENE_MEAN[0] + 750, SCENE_MEAN[1] + 750, SCENE_MEAN[2] - 180])
        pred.reshape(3, -1)[:, idx] = rng.uniform(lo[:, None], hi[:, None], size=(3, n_out))
    gt_lab = np.where(hit[None], gt, NODATA)
    return dict(coords=pred.astype(np.float32), gt_coords=gt_lab.astype(np.float32), pose=pose,
                focal=float(focal), ppx=float(ppx), ppy=float(ppy))


def make_batch(seed0, count, **kw):
    """`count` frames with seeds seed0, seed0+1, ... stacked: coords [B,3,Ho,Wo], gt [B,3,Ho,Wo], poses [B,4,4]."""
    scenes = [make_scene(seed0 + i, **kw) for i in range(count)]
    return (np.stack([s["coords"] for s in scenes]), np.stack([s["gt_coords"] for s in scenes]),
            np.stack([s["pose"] for s in scenes]))


def pose_error(gt_pose, est_pose):
    """(translation error [m], rotation error [deg]) as utils/evaluation.py:121-132 computes them
    (rotation angle of R_est^T R_gt; the angle is what ||cv2.Rodrigues(.)|| returns)."""
    gt_pose = np.asarray(gt_pose, np.float64)
    est_pose = np.asarray(est_pose, np.float64)
    t_err = float(np.linalg.norm(gt_pose[0:3, 3] - est_pose[0:3, 3]))
    r = est_pose[0:3, 0:3].T @ gt_pose[0:3, 0:3]
    # angle from the skew part and the trace (robust near 0 and near pi)
    s = 0.5 * np.linalg.norm([r[2, 1] - r[1, 2], r[0, 2] - r[2, 0], r[1, 0] - r[0, 1]])
    c = 0.5 * (np.trace(r) - 1.0)
    r_err = float(np.degrees(np.arctan2(s, c)))
    return t_err, r_err
