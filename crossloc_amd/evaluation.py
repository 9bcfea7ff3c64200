"""Evaluation harness either side of `dsacstar.forward_rgb` — our counterpart of the reference's
utils/evaluation.py (scene_coords_eval :135-190, get_pose_err :121-132, scene_coords_printout :193-244) and of
the per-image loop of test_single_task.py:328-366, plus what the reference does not have: batched
localisation (CNN forward + HIP DSAC* per batch) and image-level sharding over the GPUs of a node with ONE
all-gather of the per-image errors (the median is not decomposable, hence gather not reduce).
"""
import numpy as np
import os

import torch


def get_pose_err(gt_pose, est_pose):
    """utils/evaluation.py:121-132.  Translation error (m) and rotation error (deg) between two 4x4
    cam->world matrices.  The reference takes ||cv2.Rodrigues(R_est^T R_gt)||, i.e. the rotation angle;
    computed here from the skew part and the trace (no OpenCV)."""
    gt_pose = np.asarray(gt_pose, np.float64)
    est_pose = np.asarray(est_pose, np.float64)
    transl_err = float(np.linalg.norm(gt_pose[0:3, 3] - est_pose[0:3, 3]))
    r = est_pose[0:3, 0:3].T.dot(gt_pose[0:3, 0:3])
    s = 0.5 * np.linalg.norm([r[2, 1] - r[1, 2], r[0, 2] - r[2, 0], r[1, 0] - r[0, 1]])
    c = 0.5 * (np.trace(r) - 1.0)
    return transl_err, float(np.degrees(np.arctan2(s, c)))


def pose_errors(gt_poses, est_poses):
    """Batched get_pose_err on tensors [K,4,4] (any device) -> (t_err[K] m, r_err[K] deg), float64."""
    g = gt_poses.to(torch.float64)
    e = est_poses.to(torch.float64)
    t_err = torch.linalg.norm(g[:, :3, 3] - e[:, :3, 3], dim=1)
    r = e[:, :3, :3].transpose(1, 2) @ g[:, :3, :3]
    sk = torch.stack([r[:, 2, 1] - r[:, 1, 2], r[:, 0, 2] - r[:, 2, 0], r[:, 1, 0] - r[:, 0, 1]], dim=1)
    s = 0.5 * torch.linalg.norm(sk, dim=1)
    c = 0.5 * (r[:, 0, 0] + r[:, 1, 1] + r[:, 2, 2] - 1.0)
    return t_err, torch.rad2deg(torch.atan2(s, c))


def pick_valid_points(coord_input, nodata_value):
    """utils/learning.py:49-71 with boolean=True: [B,C,N] -> bool [B,N], true where no channel is nodata."""
    return torch.sum(coord_input == nodata_value, dim=1) == 0


def scene_coords_eval(scene_coords, gt_coords, gt_pose, nodata_value, focal_length, image_h, image_w,
                      hypotheses, threshold, inlier_alpha, max_pixel_error, output_subsample, verbose=False):
    """utils/evaluation.py:135-190 — batch size one, same arguments and return tuple
    (t_err, r_err, est_xyz, coords_error over has-data cells, out_pose 4x4).  The reference moves the
    prediction to the CPU before the solver (:161); here it may stay on the GPU."""
    import dsacstar
    gt_pose_np = gt_pose[0].detach().cpu().numpy()
    out_pose = torch.zeros((4, 4))
    dsacstar.forward_rgb(scene_coords, out_pose, hypotheses, threshold, focal_length,
                         float(image_w / 2), float(image_h / 2), inlier_alpha, max_pixel_error, output_subsample)
    t_err, r_err = get_pose_err(gt_pose_np, out_pose.numpy())
    est_xyz = out_pose[0:3, 3].tolist()
    sc = scene_coords.detach().cpu().view(scene_coords.size(0), 3, -1)
    gt = gt_coords.detach().cpu().view(gt_coords.size(0), 3, -1)
    mask = pick_valid_points(gt, nodata_value)
    coords_error = torch.norm(gt - sc, dim=1, p=2)
    coords_error_valdata = coords_error[mask].tolist()
    if verbose:
        print("\nRotation Error: %.2f deg, Translation Error: %.1f m, Mean coord prediction error: %.1f m" % (
            r_err, t_err, float(np.mean(coords_error_valdata)) if coords_error_valdata else float("nan")))
    return t_err, r_err, est_xyz, coords_error_valdata, out_pose.clone()


def accuracy_report(t_err_ls, r_err_ls, coords_error_ls=None):
    """The statistics and the text of utils/evaluation.py:207-230 (errors in metres / degrees; strict `<`)."""
    t = np.asarray(t_err_ls, np.float64)
    r = np.asarray(r_err_ls, np.float64)
    n = max(len(t), 1)
    buckets = [("30m10deg", 30.0, 10.0), ("20m10deg", 20.0, 10.0), ("10m7deg", 10.0, 7.0),
               ("10m10deg", 10.0, 10.0), ("5m5deg", 5.0, 5.0), ("3m3deg", 3.0, 3.0)]
    stats = {name: float(np.sum(np.logical_and(t < tm, r < rd))) / n * 100 for name, tm, rd in buckets}
    stats.update(median_r=float(np.median(r)), median_t=float(np.median(t)), mean_r=float(np.mean(r)),
                 std_r=float(np.std(r)), mean_t=float(np.mean(t)), std_t=float(np.std(t)))
    s = '\nAccuracy:'
    s += '\n30m10deg: %.1f%%\n20m10deg: %.1f%%' % (stats["30m10deg"], stats["20m10deg"])
    s += '\n10m7deg: %.1f%%' % stats["10m7deg"]
    s += '\n10m10deg: %.1f%%' % stats["10m10deg"] + '\n5m5deg: %.1f%%' % stats["5m5deg"]
    s += '\n3m3deg: %.1f%%' % stats["3m3deg"]
    s += "\nMedian Error: %.1f deg, %.2f m" % (stats["median_r"], stats["median_t"])
    s += "\nMean Errors: %.1f plus-minus %.1f deg, %.2f plus-minus %.2f m" % (
        stats["mean_r"], stats["std_r"], stats["mean_t"], stats["std_t"])
    if coords_error_ls is not None and len(coords_error_ls):
        ce = np.asarray(coords_error_ls, np.float64)
        s += "\nCoordinate regression error: mean {:.1f}, std {:.1f}, median {:.1f}".format(
            np.mean(ce), np.std(ce), np.median(ce))
    return stats, s


def scene_coords_printout(t_err_ls, r_err_ls, est_xyz_ls, coords_error_ls, testing_log=None, section="test"):
    """utils/evaluation.py:193-244 without the numpy pose dumps: prints (and appends to testing_log)."""
    coords = np.concatenate([np.asarray(c, np.float64).ravel() for c in coords_error_ls]) if len(coords_error_ls) else None
    stats, s = accuracy_report(t_err_ls, r_err_ls, coords)
    print(s)
    if testing_log:
        with open(testing_log, 'a') as f:
            f.write("{:s} Evaluation on section {:s} {:s}".format('=' * 20, section, '=' * 20) + '\n')
            f.write(s)
            f.write('\n')
    return stats


# ------------------------------------------------------------------------------------------ semantics metrics

def confusion_matrix(gt_label, pred_label, num_class):
    """Rows = ground truth, columns = prediction; labels outside [0, num_class) are ignored
    (utils/evaluation.py:373-378).  Tensors [..] of equal shape, any device; returns int64 [num_class, num_class]."""
    g = gt_label.reshape(-1).to(torch.int64)
    p = pred_label.reshape(-1).to(device=g.device, dtype=torch.int64)
    m = (g >= 0) & (g < num_class)
    idx = num_class * g[m] + p[m]
    return torch.bincount(idx, minlength=num_class * num_class).reshape(num_class, num_class)


def segmentation_metrics(cm):
    """Pixel accuracy, mean IoU and frequency-weighted IoU of one confusion matrix (utils/evaluation.py:346-371;
    classes absent from both ground truth and prediction are left out of the mean like np.nanmean does)."""
    cm = np.asarray(cm.cpu() if isinstance(cm, torch.Tensor) else cm, np.float64)
    diag = np.diag(cm)
    with np.errstate(divide="ignore", invalid="ignore"):
        acc = diag.sum() / cm.sum()
        iu = diag / (cm.sum(axis=1) + cm.sum(axis=0) - diag)
        freq = cm.sum(axis=1) / cm.sum()
    miou = float(np.nanmean(iu))
    fwiou = float((freq[freq > 0] * iu[freq > 0]).sum())
    return float(acc), miou, fwiou


def semantic_eval(semantic_logits, gt_label, mute=False):
    """utils/evaluation.py:388-415: per-image accuracy / mIoU / fwIoU of full-size class logits [B,6,H,W] against
    gt_label [B,1,H,W].  Returns (class_prediction [B,H,W] on the CPU, miou[B], fwiou[B], acc[B]) like the reference;
    arg-max and the confusion matrices are computed on the logits' device."""
    num_class = semantic_logits.shape[1]
    gt = gt_label.squeeze(1)
    pred = torch.argmax(semantic_logits, dim=1)               # arg-max of log-softmax = arg-max of the logits
    assert gt.shape == pred.shape
    miou_ls, fwiou_ls, acc_ls = [], [], []
    for g, p in zip(gt, pred):
        acc, miou, fwiou = segmentation_metrics(confusion_matrix(g.to(p.device), p, num_class))
        acc_ls.append(acc); miou_ls.append(miou); fwiou_ls.append(fwiou)
    miou_ls, fwiou_ls, acc_ls = np.array(miou_ls), np.array(fwiou_ls), np.array(acc_ls)
    if not mute:
        print("Metrics within the batch: mean accuracy: {:.2f}%, mean IoU: {:.2f}%, frequency weighted IoU: {:.2f}%".
              format(acc_ls.mean() * 100, miou_ls.mean() * 100, fwiou_ls.mean() * 100))
    return pred.cpu(), miou_ls, fwiou_ls, acc_ls


# ------------------------------------------------------------------------------------------ sharded evaluation

def shard_indices(num_images, rank, world_size):
    """Image i belongs to rank i % world_size."""
    return list(range(rank, num_images, world_size))


def gather_errors(local_vals, num_images, rank, world_size, group=None):
    """All-gather per-image values of the i%R sharding into the global order.

    local_vals: tensor [ceil-ish(K/R), D] on this rank's device (rows for images rank, rank+R, ...).
    Ragged shards are padded with NaN to ceil(K/R) rows and the padding dropped after
    the gather, so every rank ends up with the identical [K, D] tensor whatever R is."""
    per = (num_images + world_size - 1) // world_size
    D = local_vals.shape[1]
    pad = torch.full((per, D), float("nan"), dtype=local_vals.dtype, device=local_vals.device)
    pad[:local_vals.shape[0]] = local_vals
    if world_size == 1 and group is None:
        gathered = [pad]
    else:                                                    # (a one-rank group passed explicitly still runs the collective)
        import torch.distributed as dist
        gathered = [torch.empty_like(pad) for _ in range(world_size)]
        dist.all_gather(gathered, pad, group=group)
    out = torch.empty((per * world_size, D), dtype=local_vals.dtype, device=local_vals.device)
    for r in range(world_size):
        out[r::world_size] = gathered[r]
    return out[:num_images]


def _focal_args(focal, n):
    """focal: one number for the whole batch, or one per frame (sequence / tensor: the dataset computes it per frame from
    calibration/*.txt scaled by the stored image height, dataloader/dataloader.py:263-266).  -> (scalar, focals kwarg)."""
    import numbers
    if isinstance(focal, (numbers.Real, np.generic)):              # Python and numpy scalars alike
        return float(focal), None
    f = torch.as_tensor(focal, dtype=torch.float32).reshape(-1)
    if f.numel() == 1:                                             # 0-dim / one-element tensor or sequence: one focal for all
        return float(f[0]), None
    if f.numel() != n:
        raise RuntimeError("expected %d focal lengths, got %d" % (n, f.numel()))
    return float(f[0]), f


def localize_batch(network, images, n_hyp, focal, image_h, image_w, image0=0, image_stride=1,
                   threshold=10.0, inlier_alpha=100.0, max_pixel_error=100.0, scene_coords=None, plant=None):
    """One batch of the test_single_task.py:347-366 loop on the GPU: eval-mode CNN forward, sigma dropped
    (:354), HIP DSAC* on all images of the batch.  Returns (poses [B,4,4] cuda, predictions [B,4,Ho,Wo]).
    `focal`: a number, or one focal length per frame.
    `scene_coords` overrides the solver input (synthetic scenes: untrained weights do not predict a scene).
    `plant` [B,3,Ho,Wo]: written into the coordinate channels of the network output after the head; the solver then
    consumes the network's own output tensor (the strided `pred[:, :3]` view) exactly as with trained weights."""
    import dsacstar
    with torch.no_grad():
        pred = network(images)
    nt = network.num_task_channel
    if plant is not None:
        pred[:, :nt].copy_(plant)
    coords = pred[:, :nt] if scene_coords is None else scene_coords
    poses = torch.zeros((coords.shape[0], 4, 4), dtype=torch.float32, device=coords.device)
    f0, focals = _focal_args(focal, coords.shape[0])
    dsacstar.forward_rgb_batch(coords, poses, n_hyp, threshold, f0, float(image_w / 2), float(image_h / 2),
                               inlier_alpha, max_pixel_error, network.OUTPUT_SUBSAMPLE,
                               image0=image0, image_stride=image_stride, focals=focals)
    return poses, pred


class PipelinedLocalizer:
    """Software pipeline over batches on HIP streams.  The CNN of a batch runs as `cnn_streams` sub-batches on their own
    streams (with 2, the MFMA-bound GEMMs of one overlap the HBM-bound Winograd transforms / GroupNorm passes of the
    other: +9 % images/s at 24 frames; 1 by default), and the DSAC* solver of batch s (24 workgroups, latency-bound: it cannot fill 256 CUs on its
    own) runs on a side stream under the CNN of batch s+1; events order solver(s) after CNN(s).  The reference
    serialises the two stages per image (GPU forward, .cpu(), CPU solver: test_single_task.py:347-363)."""

    def __init__(self, network, n_hyp, focal, image_h, image_w, threshold=10.0, inlier_alpha=100.0,
                 max_pixel_error=100.0, cnn_streams=1):
        self.net, self.n_hyp, self.focal = network, n_hyp, focal
        self.h, self.w = image_h, image_w
        self.thr, self.alpha, self.maxerr = threshold, inlier_alpha, max_pixel_error
        # (round 5) the solver's stream has the HIGHER priority: its 95 workgroups (64.8 KB of LDS each) otherwise queue behind the
        # persistent workgroups of the next batch's stem and GEMM kernels, and both stages run slow for 5 ms instead of the 2 ms
        # the solver needs when it gets its CUs at once (XL_SOLVER_PRIORITY=0: equal priorities, the round-4 behaviour)
        prio = -1 if os.environ.get("XL_SOLVER_PRIORITY", "1") not in ("", "0") else 0
        self.side = torch.cuda.Stream(priority=prio)
        self.cnn = [torch.cuda.Stream() for _ in range(max(1, cnn_streams))]

    def forward_cnn(self, images, plant=None):
        """The network on `images`, split over the CNN streams.  Returns (predictions, events): the prediction tensor is
        complete once every event has fired (the caller's stream is NOT made to wait).  `plant`: see localize_batch."""
        main = torch.cuda.current_stream()
        n = min(len(self.cnn), images.shape[0])
        bounds = [round(i * images.shape[0] / n) for i in range(n + 1)]
        nt = self.net.num_task_channel
        outs, events = [], []
        for i in range(n):
            st = self.cnn[i]
            st.wait_stream(main)                           # the images (and `plant`) are ready on the caller's stream
            with torch.cuda.stream(st), torch.no_grad():
                outs.append(self.net(images[bounds[i]:bounds[i + 1]], plan_slot=i + 1))
                if plant is not None:
                    outs[-1][:, :nt].copy_(plant[bounds[i]:bounds[i + 1]])
                ev = torch.cuda.Event()
                ev.record(st)
                events.append(ev)
        if n == 1:
            return outs[0], events
        with torch.cuda.stream(self.cnn[0]):
            for ev in events[1:]:
                self.cnn[0].wait_event(ev)
            pred = torch.cat(outs, 0)
            ev = torch.cuda.Event()
            ev.record(self.cnn[0])
        for o, st in zip(outs, self.cnn):
            o.record_stream(self.cnn[0])
        return pred, [ev]

    def submit(self, images, image0=0, image_stride=1, scene_coords=None, plant=None):
        """Enqueue one batch; returns (poses [B,4,4], predictions).  Both are valid after finish() (or after
        synchronising the side stream)."""
        import dsacstar
        pred, events = self.forward_cnn(images, plant)
        coords = pred[:, :self.net.num_task_channel] if scene_coords is None else scene_coords
        poses = torch.empty((coords.shape[0], 4, 4), dtype=torch.float32, device=coords.device)   # (the solver writes all 16)
        for ev in events:
            self.side.wait_event(ev)
        self.side.wait_stream(torch.cuda.current_stream())      # `poses` / `scene_coords` come from the caller's stream
        with torch.cuda.stream(self.side):
            dsacstar.forward_rgb_batch(coords, poses, self.n_hyp, self.thr, self.focal, float(self.w / 2),
                                       float(self.h / 2), self.alpha, self.maxerr, self.net.OUTPUT_SUBSAMPLE,
                                       image0=image0, image_stride=image_stride)
        pred.record_stream(self.side)
        poses.record_stream(self.side)
        return poses, pred

    def finish(self):
        """Make the caller's stream wait for every enqueued CNN and solver launch."""
        for st in self.cnn:
            torch.cuda.current_stream().wait_stream(st)
        torch.cuda.current_stream().wait_stream(self.side)
