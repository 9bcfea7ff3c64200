"""CPU: the data-path oracle (oracle/data_oracle.py) against the third-party arithmetic it restates — the Pillow
installed in the image for the resampler and the uint8 colour jitter (bit for bit), torch's grid_sample for the
nearest-neighbour rotation — and the reference-shaped pipeline composition."""
import numpy as np
import pytest
import torch
from PIL import Image, ImageEnhance

from oracle import data_oracle as do


@pytest.mark.parametrize("H,W,oh,ow", [(480, 720, 480, 720), (960, 1440, 480, 720), (600, 800, 480, 640), (300, 451, 480, 721),
                                       (1080, 1920, 480, 853), (97, 131, 64, 86), (64, 64, 1, 1)])
def test_resampler_equals_pillow(H, W, oh, ow):
    img = np.random.default_rng(H + W).integers(0, 256, (H, W, 3), dtype=np.uint8)
    ref = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BILINEAR))
    assert np.array_equal(do.pil_resize_bilinear(img, oh, ow), ref)


def test_resize_target_is_torchvision_smaller_edge_rule():
    assert do.resize_target(960, 1440, 480) == (480, 720)
    assert do.resize_target(480, 720, 480) == (480, 720)
    assert do.resize_target(1080, 1920, 480) == (480, 853)
    assert do.resize_target(800, 600, 480) == (640, 480)


@pytest.mark.parametrize("b,c", [(0.9, 1.1), (1.1, 0.9), (1.0, 1.0), (0.93, 1.07), (1.099, 0.901), (0.0, 2.0)])
@pytest.mark.parametrize("contrast_first", [0, 1])
def test_colour_jitter_equals_pillow(b, c, contrast_first):
    img = np.random.default_rng(int(b * 100 + c * 10)).integers(0, 256, (120, 160, 3), dtype=np.uint8)
    pil = Image.fromarray(img)
    if contrast_first:
        pil = ImageEnhance.Brightness(ImageEnhance.Contrast(pil).enhance(c)).enhance(b)
    else:
        pil = ImageEnhance.Contrast(ImageEnhance.Brightness(pil).enhance(b)).enhance(c)
    assert np.array_equal(do.color_jitter_u8(img, b, c, contrast_first), np.asarray(pil))


def test_to_tensor_normalize_equals_torch():
    img = np.random.default_rng(3).integers(0, 256, (33, 47, 3), dtype=np.uint8)
    t = torch.from_numpy(img).permute(2, 0, 1).float().div(255)
    assert np.array_equal(do.to_tensor_normalize(img), t.numpy())
    mean, std = (0.4245, 0.4375, 0.3836), (0.1823, 0.1701, 0.1854)
    ref = (t - torch.tensor(mean)[:, None, None]) / torch.tensor(std)[:, None, None]
    assert np.array_equal(do.to_tensor_normalize(img, mean, std), ref.numpy())


@pytest.mark.parametrize("H,W", [(37, 53), (60, 90), (120, 175)])
@pytest.mark.parametrize("angle", [0.0, 30.0, -30.0, 12.345, -7.0, 29.999])
def test_rotation_index_agrees_with_grid_sample(H, W, angle):
    """torchvision's tensor rotate = affine grid + grid_sample(nearest).  The restated index arithmetic must pick the same
    source pixel; a handful of exact rounding ties may differ (bmm accumulation order), none at these sizes."""
    x = torch.randn(2, 3, H, W, generator=torch.Generator().manual_seed(H))
    iy, ix = do.rotate_source_index(H, W, angle)
    a = x[:, :, np.clip(iy, 0, None), np.clip(ix, 0, None)].clone()
    a[:, :, torch.from_numpy(iy < 0)] = -1.0
    b = do.rotate_by_grid_sample(x, angle, -1.0)
    assert (a != b).float().mean().item() < 1e-4
    if angle == 0.0:
        assert torch.equal(a, x)


def test_batch_resize_shapes_and_fill():
    x = torch.rand(2, 3, 48, 72)
    out = do.batch_resize_images(x, 1.25, 20.0)
    assert out.shape == (2, 3, 60, 90)
    assert (out[:, :, 0, 0] == -1.0).all() and out[:, :, 30, 45].min() >= 0.0        # corner outside, centre inside
    lab = torch.rand(2, 3, 6, 9)
    ol = do.batch_resize_labels(lab, 8, 12, 20.0)
    assert ol.shape == (2, 3, 8, 12) and (ol[:, :, 0, 0] == -1.0).all()
    # every label value inside the frame is one of the source values (nearest twice)
    inside = ol[ol != -1.0]
    assert np.isin(inside.numpy(), lab.numpy()).all()


@pytest.mark.parametrize("b,c,contrast_first", [(0.93, 1.07, False), (1.08, 0.91, True), (1.0, 1.0, False)])
def test_grayscale_pipeline_equals_pillow(b, c, contrast_first):
    """Resize -> Grayscale -> ColorJitter(brightness, contrast) on the 'L' image (dataloader.py:359-373) against Pillow."""
    rng = np.random.default_rng(7)
    img = rng.integers(0, 256, (97, 131, 3), dtype=np.uint8)
    pil = Image.fromarray(img).convert("L")
    assert np.array_equal(do.to_gray_u8(img), np.asarray(pil))
    if contrast_first:
        want = ImageEnhance.Brightness(ImageEnhance.Contrast(pil).enhance(c)).enhance(b)
    else:
        want = ImageEnhance.Contrast(ImageEnhance.Brightness(pil).enhance(b)).enhance(c)
    assert np.array_equal(do.color_jitter_gray_u8(do.to_gray_u8(img), b, c, contrast_first), np.asarray(want))
    x = do.prepare_image(img, 97, (b, c, contrast_first), (0.4308,), (0.1724,), grayscale=True)
    assert x.shape == (1, 97, 131)
    assert np.allclose(x[0], (np.asarray(want, np.float32) / 255.0 - 0.4308) / 0.1724, atol=1e-6)
