"""CPU: the CrossLoc on-disk format reader (crossloc_amd/dataset.py) on a scene written by its own writer:
tensor contract of dataloader/dataloader.py (shapes, nodata = -1, un-normalised [0,1] input with raw_image=True,
focal rescaling, file ordering)."""
import numpy as np
import pytest
import torch

from crossloc_amd import dataset, synth


def test_round_trip_scene(tmp_path):
    root = dataset.write_synthetic_scene(str(tmp_path / "scene"), 3, seed=77)
    ds = dataset.CamLocDataset(root, mode=1, sparse=True, coord=True, raw_image=True)
    assert len(ds) == 3
    image, pose, gt, focal, name = ds[1]
    assert image.shape == (3, 480, 720) and image.dtype == torch.float32
    assert 0.0 <= image.min() and image.max() <= 1.0 and image.max() > 0.9          # raw [0,1] RGB
    sc = synth.make_scene(78, noise=0.5, outlier_ratio=0.0)
    assert torch.allclose(pose, torch.from_numpy(sc["pose"]).float(), atol=1e-4)
    assert torch.equal(gt, torch.from_numpy(sc["gt_coords"])) and gt.shape == (3, 60, 90)
    assert focal == pytest.approx(480.0) and name.endswith("frame_00001.png")
    norm = dataset.CamLocDataset(root, raw_image=False)[1][0]
    assert torch.allclose(norm[0], (image[0] - 0.4245) / 0.1823, atol=1e-6)


def test_resize_scales_focal_length(tmp_path):
    from PIL import Image
    root = dataset.write_synthetic_scene(str(tmp_path / "scene"), 1, seed=5)
    p = ds_path = root + "/rgb/frame_00000.png"
    Image.open(p).resize((1440, 960)).save(ds_path)                                   # stored at twice the size
    np.savetxt(root + "/calibration/frame_00000.txt", [960.0])
    image, _, _, focal, _ = dataset.CamLocDataset(root, raw_image=True)[0]
    assert image.shape == (3, 480, 720) and focal == pytest.approx(480.0)


def test_unsupported_modes_raise(tmp_path):
    root = dataset.write_synthetic_scene(str(tmp_path / "scene"), 1)
    with pytest.raises(NotImplementedError):
        dataset.CamLocDataset(root, augment=True, batch=False)
    with pytest.raises(NotImplementedError):
        dataset.CamLocDataset(root, mode=2)
    with pytest.raises(Exception):
        dataset.CamLocDataset(root, coord=False)


def test_training_items_are_decoded_frames_only(tmp_path):
    """augment=True: __getitem__ decodes and hands over the uint8 frame; the transform pipeline runs on the GPU in
    collate_gpu (tests/test_data_gpu.py)."""
    root = dataset.write_synthetic_scene(str(tmp_path / "scene"), 2, seed=9)
    ds = dataset.CamLocDataset(root, augment=True, batch=True)
    frame, pose, gt, focal, name = ds[0]
    assert frame.dtype == torch.uint8 and frame.shape == (480, 720, 3)
    assert pose.shape == (4, 4) and gt.shape == (3, 60, 90) and focal == pytest.approx(480.0)
    # raw_image supersedes augmentation (dataloader.py:217-219)
    assert dataset.CamLocDataset(root, augment=True, raw_image=True)[0][0].dtype == torch.float32


def test_semantics_labels_and_grayscale_evaluation_path(tmp_path):
    """semantics/*.npy: raw class ids trimmed to the 6 training classes, float [1,H,W] (dataloader.py:337-338,
    loss/semantics.py:21-41); grayscale: Resize -> Grayscale -> ToTensor -> Normalize(0.4308, 0.1724) (:171-187)."""
    from PIL import Image
    from crossloc_amd.loss import trim_semantic_label
    root = dataset.write_synthetic_scene(str(tmp_path / "scene"), 2, seed=3, semantics=True)
    raw = np.load(root + "/semantics/frame_00001.npy")
    assert set(np.unique(raw)) <= {0, 1, 2, 3, 6, 9, 17} and raw.max() > 5
    assert np.array_equal(trim_semantic_label(np.array([0, 1, 2, 3, 6, 9, 17])), [0, 1, 1, 2, 3, 4, 5])
    with pytest.raises(AssertionError):
        trim_semantic_label(np.array([0, 8]))
    ds = dataset.CamLocDataset(root, coord=False, semantics=True)
    image, pose, gt, focal, name = ds[1]
    assert gt.shape == (1, 480, 720) and gt.dtype == torch.float32
    assert torch.equal(gt[0], torch.from_numpy(trim_semantic_label(raw)).float())
    both = dataset.CamLocDataset(root, coord=True, semantics=True)[0][2]
    assert set(both) == {"coord", "semantics"}
    gray = dataset.CamLocDataset(root, grayscale=True)[1][0]
    want = np.asarray(Image.open(root + "/rgb/frame_00001.png").convert("L"), np.float32) / 255.0
    assert gray.shape == (1, 480, 720) and torch.allclose(gray[0], (torch.from_numpy(want) - 0.4308) / 0.1724, atol=1e-6)


def test_evaluation_path_resizes_like_torchvision(tmp_path):
    """Resize(image_height): the SMALLER edge becomes image_height and the other one is truncated - 600 x 801 stored at
    height 480 gives 640 columns (not round(640.8) = 641), and a portrait frame scales its width."""
    from PIL import Image
    root = dataset.write_synthetic_scene(str(tmp_path / "scene"), 1, seed=5)
    p = root + "/rgb/frame_00000.png"
    Image.open(p).resize((801, 600)).save(p)
    assert dataset.CamLocDataset(root, raw_image=True)[0][0].shape == (3, 480, 640)
    Image.open(p).resize((600, 801)).save(p)
    assert dataset.CamLocDataset(root, raw_image=True)[0][0].shape == (3, 640, 480)
    assert dataset._resized_shape(480, 720, 480) == (480, 720)


def test_collate_gpu_refuses_to_run_in_a_dataloader_worker(tmp_path, monkeypatch):
    root = dataset.write_synthetic_scene(str(tmp_path / "scene"), 1)
    ds = dataset.CamLocDataset(root, augment=True, batch=True)
    monkeypatch.setattr(torch.utils.data, "get_worker_info", lambda: object())
    with pytest.raises(RuntimeError, match="collate_host"):
        ds.collate_gpu([ds[0]])
    frames, poses, labels, focals, files = dataset.CamLocDataset.collate_host([ds[0], ds[0]])
    assert frames.shape == (2, 480, 720, 3) and frames.dtype == torch.uint8 and not frames.is_cuda and len(focals) == 2
