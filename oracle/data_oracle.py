"""CPU restatement of the image / label preparation of the reference's training and evaluation data path —
TEST INFRASTRUCTURE ONLY (tests/ and nothing else imports it).

What is restated (reference file:line, relative to /root/reference/):
  per-frame transform pipelines        dataloader/dataloader.py:189-232 (eval / raw), :349-393 (training, batch mode:
                                       scale 1, angle 0 per item)
  common scale + rotation of a batch   dataloader/dataloader.py:512-563 (`batch_resize`)
Third-party arithmetic behind those lines, restated from the published algorithms:
  * torchvision.transforms.Resize on a PIL image = PIL.Image.resize(BILINEAR): Pillow's two-pass separable resampler
    with 22-bit fixed-point coefficients on uint8 (libImaging/Resample.c).  PINNED: tests/test_data_oracle.py compares
    `pil_resize_bilinear` bit for bit with the Pillow installed in the build image (12.2; the reference pins Pillow
    through torchvision 0.10, same resampler).
  * torchvision.transforms.ColorJitter(brightness, contrast) on a PIL image = PIL.ImageEnhance.Brightness / Contrast
    = Image.blend against black / against the rounded mean of the "L" conversion, uint8 in and out.  PINNED against
    the installed Pillow the same way.
  * ToTensor + Normalize: uint8 -> float32 / 255, (x - mean) / std in float32 (torch ops, bit-exact by construction).
  * F.interpolate(bilinear, align_corners=False) and F.interpolate(nearest): torch itself is the checker.
  * torchvision.transforms.functional.rotate(tensor, angle, fill) with its defaults (nearest, expand=False, centre =
    image centre): inverse rotation of the output pixel centres about the image centre, nearest source pixel
    (round-half-even like grid_sample), `fill` outside.  torchvision is NOT installed in the build image and the
    reference ships no vectors, so this one step is restated from the published algorithm only (PARITY UNPINNED for the
    rotation step); it is checked against torch's own affine_grid + grid_sample(nearest), which is what torchvision
    0.10 calls.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _triangle(x):
    x = abs(x)
    return 1.0 - x if x < 1.0 else 0.0


def resample_coefficients(in_size, out_size):
    """Pillow precompute_coeffs() for the bilinear (triangle) filter over the whole axis (box = 0 .. in_size).
    Returns (bounds int32 [out,2] = first source index / tap count, coeffs int32 [out, ksize] in 22-bit fixed point)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = [_triangle((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = sum(k)
        for x in range(xmax):
            v = k[x] / ww if ww != 0.0 else k[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _resample_axis(img, bounds, kk, axis):
    """One pass of ImagingResampleHorizontal/Vertical_8bpc: ss = 2^21 + sum(pixel * k); clip8(ss >> 22)."""
    img = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((bounds.shape[0],) + img.shape[1:], np.uint8)
    for xx in range(bounds.shape[0]):
        x0, n = int(bounds[xx, 0]), int(bounds[xx, 1])
        acc = np.full(img.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for x in range(n):
            acc += img[x0 + x] * int(kk[xx, x])
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def pil_resize_bilinear(img, out_h, out_w):
    """uint8 [H,W,C] -> uint8 [out_h,out_w,C] like PIL.Image.resize((out_w, out_h), Image.BILINEAR): horizontal pass
    first, then vertical, each skipped when the size does not change (ImagingResample)."""
    img = np.asarray(img, np.uint8)
    H, W = img.shape[:2]
    if W != out_w:
        img = _resample_axis(img, *resample_coefficients(W, out_w), axis=1)
    if H != out_h:
        img = _resample_axis(img, *resample_coefficients(H, out_h), axis=0)
    return img


def resize_target(h, w, image_height):
    """torchvision Resize(int): the smaller edge becomes `image_height` (dataloader.py:191, 352)."""
    if (w <= h and w == image_height) or (h <= w and h == image_height):
        return h, w
    if w < h:
        return int(image_height * h / w), image_height
    return image_height, int(image_height * w / h)


def _blend_u8(degenerate, img, factor):
    """PIL.Image.blend(im1 = degenerate, im2 = image, alpha = factor) on uint8 (libImaging/Blend.c): interpolation
    truncated for 0 <= alpha <= 1, extrapolation clipped to 0..255 then truncated."""
    a = np.float32(factor)
    t = degenerate.astype(np.float32) + a * (img.astype(np.float32) - degenerate.astype(np.float32))
    if 0.0 <= factor <= 1.0:
        return t.astype(np.uint8)
    return np.where(t <= 0.0, 0, np.where(t >= 255.0, 255, t)).astype(np.uint8)


def gray_mean_u8(img):
    """int(mean of convert('L') + 0.5) of ImageEnhance.Contrast; L = (R*19595 + G*38470 + B*7471 + 0x8000) >> 16."""
    r, g, b = (img[..., c].astype(np.int64) for c in range(3))
    lum = (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16
    return int(lum.sum() / lum.size + 0.5)


def color_jitter_u8(img, brightness, contrast, contrast_first):
    """ColorJitter with only brightness / contrast active (dataloader.py:356, 374): the two adjustments in the order
    torchvision drew for this call, each on uint8 like PIL.ImageEnhance."""
    def bright(x):
        return _blend_u8(np.zeros_like(x), x, brightness)

    def contr(x):
        return _blend_u8(np.full_like(x, gray_mean_u8(x)), x, contrast)
    return bright(contr(img)) if contrast_first else contr(bright(img))


def to_tensor_normalize(img_u8, mean=None, std=None):
    """ToTensor (+ Normalize): uint8 [H,W,3] -> float32 [3,H,W]."""
    x = np.ascontiguousarray(np.transpose(img_u8, (2, 0, 1))).astype(np.float32) / np.float32(255.0)
    if mean is not None:
        x = (x - np.asarray(mean, np.float32)[:, None, None]) / np.asarray(std, np.float32)[:, None, None]
    return x


def to_gray_u8(img):
    """transforms.Grayscale() on a PIL RGB image = convert('L'): (R*19595 + G*38470 + B*7471 + 0x8000) >> 16, uint8 [H,W]."""
    r, g, b = (img[..., c].astype(np.int64) for c in range(3))
    return ((r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16).astype(np.uint8)


def color_jitter_gray_u8(img, brightness, contrast, contrast_first):
    """ColorJitter on an 'L' image: the same blends on one channel; the contrast mean is int(mean of the image + 0.5)."""
    def bright(x):
        return _blend_u8(np.zeros_like(x), x, brightness)

    def contr(x):
        return _blend_u8(np.full_like(x, int(x.astype(np.int64).sum() / x.size + 0.5)), x, contrast)
    return bright(contr(img)) if contrast_first else contr(bright(img))


def prepare_image(img_u8, image_height, jitter=None, mean=None, std=None, grayscale=False):
    """One frame through the reference's transform pipeline.  jitter = (brightness, contrast, contrast_first) or None.
    grayscale: the one-channel pipeline (dataloader.py:171-187, 359-373), mean / std with one entry -> [1,H,W]."""
    img = np.asarray(img_u8, np.uint8)[..., :3]
    oh, ow = resize_target(img.shape[0], img.shape[1], image_height)
    img = pil_resize_bilinear(img, oh, ow)
    if grayscale:
        img = to_gray_u8(img)
        if jitter is not None:
            img = color_jitter_gray_u8(img, *jitter)
        return to_tensor_normalize(img[..., None], mean, std)
    if jitter is not None:
        img = color_jitter_u8(img, *jitter)
    return to_tensor_normalize(img, mean, std)


# ------------------------------------------------------------------------------------------ batch_resize

def rotate_source_index(out_h, out_w, angle_deg):
    """torchvision rotate (tensor path, defaults): for every output pixel the source pixel (iy, ix) of the SAME-size
    input, or -1 outside.  float32 arithmetic in the order torchvision 0.10 executes it: the inverse matrix of -angle,
    [cos, -sin; sin, cos], divided by the half sizes (`rescaled_theta`), applied to the pixel-centre grid
    x = j + 0.5 - W/2, y = i + 0.5 - H/2, then grid_sample's un-normalisation ((g + 1) * size - 1) / 2 for
    align_corners=False and nearbyint (round half to even)."""
    f = np.float32
    th = math.radians(angle_deg)
    r00, r10 = f(f(math.cos(th)) / f(0.5 * out_w)), f(f(-math.sin(th)) / f(0.5 * out_w))
    r01, r11 = f(f(math.sin(th)) / f(0.5 * out_h)), f(f(math.cos(th)) / f(0.5 * out_h))
    xs = (np.arange(out_w, dtype=f) + f(0.5) - f(out_w * 0.5))
    ys = (np.arange(out_h, dtype=f) + f(0.5) - f(out_h * 0.5))
    X, Y = np.meshgrid(xs, ys)
    gx = (X * r00 + Y * r10).astype(f)
    gy = (X * r01 + Y * r11).astype(f)
    sx = ((gx + f(1)) * f(out_w) - f(1)) / f(2)
    sy = ((gy + f(1)) * f(out_h) - f(1)) / f(2)
    ix = np.rint(sx).astype(np.int64)
    iy = np.rint(sy).astype(np.int64)
    ok = (ix >= 0) & (ix < out_w) & (iy >= 0) & (iy < out_h)
    return np.where(ok, iy, -1), np.where(ok, ix, -1)


def batch_resize_images(images, scale_factor, angle_deg):
    """dataloader.py:527-533: F.interpolate(bilinear, align_corners=False) to (ceil(H s), ceil(W s)), then rotate
    (nearest, fill -1).  images: torch float32 [B,3,H,W] on the CPU."""
    import torch
    import torch.nn.functional as F
    H, W = images.shape[2], images.shape[3]
    oh, ow = math.ceil(H * scale_factor), math.ceil(W * scale_factor)
    x = F.interpolate(images, size=(oh, ow), mode="bilinear", align_corners=False)
    iy, ix = rotate_source_index(oh, ow, angle_deg)
    out = x[:, :, np.clip(iy, 0, None), np.clip(ix, 0, None)]
    out[:, :, torch.from_numpy(iy < 0)] = -1.0
    return out


def batch_resize_labels(labels, out_h, out_w, angle_deg, fill=-1.0):
    """dataloader.py:549-551: F.interpolate(nearest) to the label grid of the augmented image, rotate (nearest, fill)."""
    import torch
    import torch.nn.functional as F
    x = F.interpolate(labels, size=(out_h, out_w), mode="nearest")
    iy, ix = rotate_source_index(out_h, out_w, angle_deg)
    out = x[:, :, np.clip(iy, 0, None), np.clip(ix, 0, None)]
    out[:, :, torch.from_numpy(iy < 0)] = fill
    return out


def rotate_by_grid_sample(x, angle_deg, fill):
    """What torchvision 0.10's tensor rotate executes: affine_grid-style base grid + grid_sample(nearest, zeros) and a
    mask channel for the fill.  Independent of rotate_source_index (used to cross-check it)."""
    import torch
    import torch.nn.functional as F
    B, C, H, W = x.shape
    th = math.radians(angle_deg)
    theta = torch.tensor([[math.cos(th), -math.sin(th), 0.0], [math.sin(th), math.cos(th), 0.0]], dtype=x.dtype)
    d = 0.5
    base = torch.empty(1, H, W, 3, dtype=x.dtype)
    base[..., 0].copy_(torch.linspace(-W * 0.5 + d, W * 0.5 + d - 1, steps=W, dtype=x.dtype))
    base[..., 1].copy_(torch.linspace(-H * 0.5 + d, H * 0.5 + d - 1, steps=H, dtype=x.dtype).unsqueeze_(-1))
    base[..., 2].fill_(1)
    rescaled = theta.t().unsqueeze(0) / torch.tensor([0.5 * W, 0.5 * H], dtype=x.dtype)
    grid = base.view(1, H * W, 3).bmm(rescaled).view(1, H, W, 2).expand(B, H, W, 2)
    xm = torch.cat([x, torch.ones(B, 1, H, W, dtype=x.dtype)], 1)
    y = F.grid_sample(xm, grid, mode="nearest", padding_mode="zeros", align_corners=False)
    mask = y[:, -1:] < 0.5
    y = y[:, :-1]
    return torch.where(mask.expand_as(y), torch.full_like(y, fill), y)
