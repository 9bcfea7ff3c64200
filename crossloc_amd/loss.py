"""Per-pixel regression losses with the reference's call signatures, running the fused HIP kernels of
csrc/xl_loss.hip (forward + analytic backward in one pass, no host sync).

    loss/coord.py:87-188   scene_coords_regression_loss(min_depth, soft_clamp, hard_clamp, init_tolerance,
                           uncertainty, pixel_grid, nodata_value, cam_mat, scene_coords, uncertainty_map,
                           gt_poses, gt_coords, reduction='mean')
    loss/depth.py:7-76     depth_regression_loss(min_depth, hard_clamp, uncertainty, nodata_value, depth_map,
                           uncertainty_map, gt_depths, reduction='mean')
    loss/normal.py:8-127   normal_regression_loss(hard_clamp, uncertainty, nodata_value, normal_logits,
                           uncertainty_map, gt_normals, reduction='mean')

Each returns (loss, valid_pred_rate) like the reference; `loss` participates in autograd (backward hands out
the gradients the kernel already produced), `valid_pred_rate` is a 0-dim device tensor instead of a Python
float so nothing synchronises (the reference pays `.cpu()` + 4x `.item()` per call, coord.py:132, 170-175).
GPU only: there is no CPU fallback.
"""
import ctypes
import math

import torch

from . import _lib


def _bind():
    L = _lib.lib()
    if not hasattr(L, "_loss_bound"):
        vp, ci, cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
        L.xl_loss_workspace_doubles.restype = ci
        L.xl_loss_workspace_doubles.argtypes = [ci, ci, ci]
        L.xl_loss_coord.restype = ci
        L.xl_loss_coord.argtypes = [vp, vp, vp, vp, ci, ci, ci, cf, cf, cf, cf, cf, cf, cf, cf, cf, ci, ci,
                                    vp, vp, vp, vp, vp]
        L.xl_loss_depth.restype = ci
        L.xl_loss_depth.argtypes = [vp, vp, vp, ci, ci, ci, cf, cf, cf, ci, ci, vp, vp, vp, vp, vp]
        L.xl_loss_normal.restype = ci
        L.xl_loss_normal.argtypes = [vp, vp, vp, ci, ci, ci, cf, cf, ci, ci, vp, vp, vp, vp, vp]
        L.xl_loss_semantics.restype = ci
        L.xl_loss_semantics.argtypes = [vp, vp, ci, ci, ci, ci, ci, vp, vp, vp, vp]
        L._loss_bound = True
    return L


def get_cam_mat(width, height, focal_length):
    """loss/coord.py:7-17 (kept on the host: only f, cx, cy are consumed by the kernel)."""
    cam_mat = torch.eye(3)
    cam_mat[0, 0] = focal_length
    cam_mat[1, 1] = focal_length
    cam_mat[0, 2] = width / 2
    cam_mat[1, 2] = height / 2
    return cam_mat


def get_pixel_grid(SUBSAMPLE):
    """utils/learning.py:20-35: centre-of-cell pixel positions [2,135,135]; vectorised, host tensor."""
    n = math.ceil(1080 / SUBSAMPLE)
    r = torch.arange(n, dtype=torch.float32) * SUBSAMPLE + SUBSAMPLE / 2
    return torch.stack([r[None, :].expand(n, n), r[:, None].expand(n, n)], 0).contiguous()


def _mode(uncertainty):
    if uncertainty is None:
        return 0
    if uncertainty == 'MLE':
        return 1
    raise NotImplementedError


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _prep(t):
    if not t.is_cuda:
        raise RuntimeError("crossloc_amd.loss runs on the GPU only (no CPU fallback)")
    return t.detach().to(torch.float32).contiguous()


class _FusedLoss(torch.autograd.Function):
    """forward = kernel call (loss + gradients); backward = scale the stored gradients."""

    @staticmethod
    def forward(ctx, launch, per_image, pred, unc):
        # the kernels write dense NCHW gradients: never inherit the memory format of `pred` (channels_last inputs)
        dpred = torch.empty(pred.shape, dtype=torch.float32, device=pred.device)
        dunc = torch.empty(unc.shape, dtype=torch.float32, device=unc.device) if unc is not None else None
        out = launch(dpred, dunc)
        ctx.per_image = per_image
        ctx.has_unc = unc is not None
        ctx.save_for_backward(dpred, dunc if dunc is not None else dpred.new_empty(0))
        ctx.mark_non_differentiable(out)
        B = pred.shape[0]
        loss = out[2:2 + B].clone() if per_image else out[0].clone()
        return loss, out

    @staticmethod
    def backward(ctx, gloss, _gout):
        dpred, dunc = ctx.saved_tensors
        if ctx.per_image:
            scale = gloss.reshape(-1, *([1] * (dpred.dim() - 1)))
            gp = dpred * scale
            gu = dunc * gloss.reshape(-1, *([1] * (dunc.dim() - 1))) if ctx.has_unc else None
        else:
            gp = dpred * gloss
            gu = dunc * gloss if ctx.has_unc else None
        return None, None, gp, gu


def _run(kind, launch_args_fn, pred, unc, reduction):
    if reduction not in ('mean', None):
        raise NotImplementedError
    per_image = reduction is None
    B, _, Ho, Wo = pred.shape
    dev = pred.device
    L = _bind()
    p32 = _prep(pred)
    u32 = _prep(unc) if unc is not None else None

    def launch(dpred, dunc):
        ws = torch.empty(L.xl_loss_workspace_doubles(B, Ho, Wo), dtype=torch.float64, device=dev)
        out = torch.empty(2 + B, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            rc = launch_args_fn(L, p32, u32, int(per_image), dpred, dunc, ws, out, stream)
        _lib.check(rc)
        return out

    loss, out = _FusedLoss.apply(launch, per_image, pred, unc)
    return loss, out[1]


def scene_coords_regression_loss(min_depth, soft_clamp, hard_clamp, init_tolerance, uncertainty,
                                 pixel_grid, nodata_value, cam_mat,
                                 scene_coords, uncertainty_map, gt_poses, gt_coords, reduction='mean'):
    """loss/coord.py:87-188.  pixel_grid / cam_mat as returned by get_pixel_grid / get_cam_mat above (host
    tensors; device tensors are accepted at the price of one D2H read)."""
    mode = _mode(uncertainty)
    cm = cam_mat.detach().cpu()
    pg = pixel_grid[:, 0, :2].detach().cpu()
    sub = float(pg[0, 1] - pg[0, 0])
    f, cx, cy = float(cm[0, 0]), float(cm[0, 2]), float(cm[1, 2])
    poses = _prep(gt_poses).reshape(-1, 16)
    gt = _prep(gt_coords)
    B, _, Ho, Wo = scene_coords.shape
    unc = uncertainty_map if mode == 1 else None

    def args(L, p32, u32, per_image, dpred, dunc, ws, out, stream):
        return L.xl_loss_coord(_ptr(p32), _ptr(u32), _ptr(poses), _ptr(gt), B, Ho, Wo, f, cx, cy, sub,
                               float(min_depth), float(soft_clamp), float(hard_clamp), float(init_tolerance),
                               float(nodata_value), mode, per_image, _ptr(dpred), _ptr(dunc), _ptr(ws), _ptr(out), stream)

    return _run("coord", args, scene_coords, unc, reduction)


def depth_regression_loss(min_depth, hard_clamp, uncertainty, nodata_value, depth_map,
                          uncertainty_map, gt_depths, reduction='mean'):
    """loss/depth.py:7-76"""
    mode = _mode(uncertainty)
    gt = _prep(gt_depths)
    B, _, Ho, Wo = depth_map.shape
    unc = uncertainty_map if mode == 1 else None

    def args(L, p32, u32, per_image, dpred, dunc, ws, out, stream):
        return L.xl_loss_depth(_ptr(p32), _ptr(u32), _ptr(gt), B, Ho, Wo, float(min_depth), float(hard_clamp),
                               float(nodata_value), mode, per_image, _ptr(dpred), _ptr(dunc), _ptr(ws), _ptr(out), stream)

    return _run("depth", args, depth_map, unc, reduction)


def normal_regression_loss(hard_clamp, uncertainty, nodata_value, normal_logits,
                           uncertainty_map, gt_normals, reduction='mean'):
    """loss/normal.py:8-127"""
    mode = _mode(uncertainty)
    gt = _prep(gt_normals)
    B, _, Ho, Wo = normal_logits.shape
    unc = uncertainty_map if mode == 1 else None

    def args(L, p32, u32, per_image, dpred, dunc, ws, out, stream):
        return L.xl_loss_normal(_ptr(p32), _ptr(u32), _ptr(gt), B, Ho, Wo, float(hard_clamp), float(nodata_value),
                                mode, per_image, _ptr(dpred), _ptr(dunc), _ptr(ws), _ptr(out), stream)

    return _run("normal", args, normal_logits, unc, reduction)


class CrossEntropyLoss2d(torch.nn.Module):
    """loss/semantics.py:10-18 — the criterion object train_single_task.py:215 constructs.  Only the configuration the
    reference uses (no class weights, reduction='none', default ignore_index) is supported; the fused kernel of
    `semantics_classification_loss` implements it, this class just carries the settings."""

    def __init__(self, weight=None, reduction='none', ignore_index=-100):
        super().__init__()
        if weight is not None or reduction != 'none' or ignore_index != -100:
            raise NotImplementedError("only CrossEntropyLoss2d() as constructed by train_single_task.py:215")


def trim_semantic_label(raw_labels):
    """loss/semantics.py:21-41: raw class ids of the urbanscape / naturescape label maps -> the 6 training classes
    (0 sky, 1 unclassified + ground, 2 low vegetation, 3 buildings, 4 water, 5 bridge deck).  numpy in, numpy out."""
    import numpy as np
    raw_labels = np.asarray(raw_labels)
    out = raw_labels.copy()
    for old, new in zip((0, 1, 2, 3, 6, 9, 17), (0, 1, 1, 2, 3, 4, 5)):
        out[raw_labels == old] = new
    assert out.min() >= 0 and out.max() <= 5, "semantic label outside the 7 known raw classes"
    return out


def semantics_classification_loss(uncertainty, semantic_logits, uncertainty_map, gt_labels, criterion, reduction):
    """loss/semantics.py:44-91: cross entropy over the C class logits [B,C,H,W] against gt_labels [B,1,H,W];
    returns (loss, share of correctly classified pixels).  One fused kernel computes the loss, the arg-max accuracy
    and d loss / d logits; the reference's host synchronisation (.cpu().numpy(), :66) is gone, so the second return
    value is a device scalar tensor."""
    if uncertainty is not None:
        raise NotImplementedError                              # semantics.py:78-81
    if not isinstance(criterion, CrossEntropyLoss2d):
        raise NotImplementedError("criterion must be crossloc_amd.loss.CrossEntropyLoss2d()")
    if reduction not in ('mean', None):
        raise NotImplementedError
    per_image = reduction is None
    B, C, H, W = semantic_logits.shape
    dev = semantic_logits.device
    L = _bind()
    p32 = _prep(semantic_logits)
    lab = gt_labels.detach().to(device=dev, dtype=torch.float32).reshape(B, H, W).contiguous()

    def launch(dpred, dunc):
        ws = torch.empty(L.xl_loss_workspace_doubles(B, H, W), dtype=torch.float64, device=dev)
        out = torch.empty(2 + B, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            rc = L.xl_loss_semantics(_ptr(p32), _ptr(lab), B, C, H, W, int(per_image), _ptr(dpred), _ptr(ws), _ptr(out), stream)
        _lib.check(rc)
        return out

    loss, out = _FusedLoss.apply(launch, per_image, semantic_logits, None)
    return loss, out[1]
