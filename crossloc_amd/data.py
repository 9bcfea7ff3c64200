"""GPU-side frame / label preparation (include/crossloc_data.h) — the work the reference leaves to CPU dataloader
workers (dataloader/dataloader.py:189-232, 349-393, 512-563): resize to the working height, colour jitter, ToTensor,
normalisation, and the common scale + rotation of a training mini-batch.  The decoded uint8 frames are uploaded once;
everything else runs in libcrossloc_hip.so.  No CPU fallback: these functions need GPU tensors."""
import ctypes
import math
import random

import torch

from . import _lib

MEAN = (0.4245, 0.4375, 0.3836)        # urbanscape statistics, dataloader/dataloader.py:193-196
STD = (0.1823, 0.1701, 0.1854)
MEAN_GRAY = (0.4308,)                  # dataloader/dataloader.py:178-180 (grayscale pipeline)
STD_GRAY = (0.1724,)


def _bind():
    L = _lib.lib()
    if not hasattr(L, "_data_bound"):
        vp, ci, cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
        L.xl_data_resized_shape.restype = ci
        L.xl_data_resized_shape.argtypes = [ci, ci, ci, ctypes.POINTER(ci), ctypes.POINTER(ci)]
        L.xl_data_prepare_workspace_bytes.restype = ctypes.c_longlong
        L.xl_data_prepare_workspace_bytes.argtypes = [ci] * 5
        L.xl_data_prepare_images.restype = ci
        L.xl_data_prepare_images.argtypes = [vp, ci, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp]
        L.xl_data_prepare_images_gray.restype = ci
        L.xl_data_prepare_images_gray.argtypes = [vp, ci, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp]
        L.xl_data_batch_augment.restype = ci
        L.xl_data_batch_augment.argtypes = [vp, vp, ci, ci, ci, ci, ci, ci, ctypes.c_double, cf, ci, vp]
        L._data_bound = True
    return L


def resized_shape(h, w, image_height):
    """Output size of torchvision's Resize(image_height): the smaller edge becomes image_height."""
    H, W = ctypes.c_int(), ctypes.c_int()
    _lib.check(_bind().xl_data_resized_shape(int(h), int(w), int(image_height), ctypes.byref(H), ctypes.byref(W)))
    return H.value, W.value


def draw_jitter(aug_brightness=0.1, aug_contrast=0.1, rng=random):
    """One draw of ColorJitter(brightness, contrast).get_params: factors uniform in [max(0, 1 - x), 1 + x] and the
    order of the two adjustments (torchvision permutes its four operations; the relative order of these two is a fair
    coin).  Returns (brightness, contrast, contrast_first)."""
    b = rng.uniform(max(0.0, 1.0 - aug_brightness), 1.0 + aug_brightness)
    c = rng.uniform(max(0.0, 1.0 - aug_contrast), 1.0 + aug_contrast)
    return b, c, 1.0 if rng.random() < 0.5 else 0.0


def prepare_images(frames_u8, image_height=480, jitter=None, normalize=True, grayscale=False):
    """frames_u8: uint8 GPU tensor [B,Hs,Ws,3|4] (decoded frames, HWC) -> float32 [B,3,H,W] network input: PIL-exact
    resize so that the smaller edge is `image_height`, optional colour jitter (list of B (brightness, contrast,
    contrast_first) triples, see draw_jitter), ToTensor, urbanscape normalisation (`normalize`; False = raw [0,1],
    what evaluation uses: utils/evaluation.py:72).  `grayscale`: the one-channel pipeline of dataloader.py:171-187 /
    :359-373 (Pillow's 'L' conversion after the resize, jitter on it, its own mean / std) -> [B,1,H,W]."""
    if not isinstance(frames_u8, torch.Tensor) or frames_u8.dtype != torch.uint8 or frames_u8.dim() != 4:
        raise RuntimeError("prepare_images expects a uint8 tensor [B,H,W,C]")
    if not frames_u8.is_cuda:
        raise RuntimeError("crossloc_amd.data runs on the GPU only (no CPU fallback)")
    x = frames_u8.contiguous()
    B, Hs, Ws, Cs = x.shape
    if Cs not in (3, 4):
        raise RuntimeError("expected RGB or RGBA frames, got %d channels" % Cs)
    L = _bind()
    H, W = resized_shape(Hs, Ws, image_height)
    out = torch.empty((B, 1 if grayscale else 3, H, W), dtype=torch.float32, device=x.device)
    ws = torch.empty(L.xl_data_prepare_workspace_bytes(B, Hs, Ws, H, W), dtype=torch.uint8, device=x.device)
    jit = None
    if jitter is not None:
        if len(jitter) != B:
            raise RuntimeError("expected %d jitter triples, got %d" % (B, len(jitter)))
        jit = (ctypes.c_float * (3 * B))(*[float(v) for t in jitter for v in t])
    m, sd = (MEAN_GRAY, STD_GRAY) if grayscale else (MEAN, STD)
    mean = (ctypes.c_float * len(m))(*m) if normalize else None
    std = (ctypes.c_float * len(sd))(*sd) if normalize else None
    fn = L.xl_data_prepare_images_gray if grayscale else L.xl_data_prepare_images
    with torch.cuda.device(x.device):
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(fn(ctypes.c_void_p(x.data_ptr()), B, Hs, Ws, Cs, int(image_height), jit, mean, std,
                      ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(ws.data_ptr()), st))
    ws.record_stream(torch.cuda.current_stream())
    return out


def batch_augment(x, out_h, out_w, angle_deg, fill, bilinear):
    """x float32 GPU [B,C,H,W] -> [B,C,out_h,out_w]: resize (bilinear like F.interpolate(align_corners=False) for
    images, 'nearest' for label maps) composed with torchvision's nearest-neighbour `rotate` by angle_deg."""
    if not x.is_cuda:
        raise RuntimeError("crossloc_amd.data runs on the GPU only (no CPU fallback)")
    x = x.detach().to(torch.float32).contiguous()
    B, C, H, W = x.shape
    out = torch.empty((B, C, out_h, out_w), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(_bind().xl_data_batch_augment(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(out.data_ptr()), B, C, H, W,
                                                 int(out_h), int(out_w), float(angle_deg), float(fill), 1 if bilinear else 0, st))
    return out


def batch_resize(images, labels, focal_lengths, scale_factor, angle_deg, output_subsample=8, semantics=False):
    """The augmentation half of CamLocDataset.batch_resize (dataloader/dataloader.py:524-551) on GPU tensors: one common
    scale and rotation for the mini-batch.  images [B,3,H,W]; labels: a tensor [B,C,h,w] or a dict of such tensors (the
    multi-label form, :553-563; non-tensor entries pass through as 0); focal lengths are multiplied by the scale (:535).
    Label maps are resized to ceil(image / output_subsample) and filled with -1 outside the rotated frame; a semantics
    map follows the image size and is filled with 0 (:545-547).  The ground-truth poses are NOT rotated in this path
    (the reference only rotates them in its per-item path, :455-464) - reproduced as is."""
    H, W = images.shape[2], images.shape[3]
    image_h, image_w = math.ceil(H * scale_factor), math.ceil(W * scale_factor)
    out_images = batch_augment(images, image_h, image_w, angle_deg, -1.0, True)
    coords_h, coords_w = math.ceil(image_h / output_subsample), math.ceil(image_w / output_subsample)

    def one(t, is_sem):
        if is_sem:
            return batch_augment(t, image_h, image_w, angle_deg, 0.0, False)
        return batch_augment(t, coords_h, coords_w, angle_deg, -1.0, False)
    if isinstance(labels, dict):
        out_labels = {k: (one(v, semantics and k == "semantics") if isinstance(v, torch.Tensor) else 0) for k, v in labels.items()}
    else:
        out_labels = one(labels, semantics)
    return out_images, out_labels, [f * scale_factor for f in focal_lengths]
