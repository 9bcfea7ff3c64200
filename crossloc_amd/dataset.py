"""Reader for the CrossLoc on-disk scene format, evaluation path only.

Reference: dataloader/dataloader.py `CamLocDataset` (:60-586).  Directory layout (:232-247):
    <root>/rgb/*.png            RGB(A) frames
    <root>/poses/*.txt          4x4 cam->world matrix (np.loadtxt)
    <root>/calibration/*.txt    focal length in pixels of the stored frame
    <root>/init/*.dat           torch-saved [3,Ho,Wo] scene coordinates, nodata = -1   (coord task)
    <root>/depth/*.dat          torch-saved [Ho,Wo] z-buffer depth                      (depth task)
    <root>/normal/*.dat         torch-saved [3,Ho,Wo] surface normals                   (normal task)
Files of one frame share their sorted position in each directory (:310-338).

Covered: mode=1 / sparse labels, augment=False (what utils/evaluation.py:20-78 builds for testing): frames are
resized to `image_height` when they are not already that tall (PIL bilinear, as torchvision's Resize does on PIL
images, :189-212) with the focal length scaled accordingly (:314-316); raw_image=True returns un-normalised
[0,1] RGB, otherwise the urbanscape mean/std normalisation (:193-196).  Training-time augmentation (rotation,
rescaling, colour jitter, :349-470) is CPU plumbing outside the hot path and is not reproduced.
"""
import os

import numpy as np
import torch
from torch.utils.data import Dataset

MEAN = (0.4245, 0.4375, 0.3836)        # dataloader.py:193-196
STD = (0.1823, 0.1701, 0.1854)


def _sorted_files(d):
    return sorted(os.path.join(d, f) for f in os.listdir(d)) if os.path.isdir(d) else []


class CamLocDataset(Dataset):
    def __init__(self, root_dir, mode=1, sparse=True, coord=True, depth=False, normal=False, semantics=False,
                 augment=False, grayscale=False, batch=True, raw_image=False, image_height=480, **unused):
        if mode != 1 or not sparse:
            raise NotImplementedError("only mode=1 with sparse labels is read (the configuration CrossLoc uses)")
        if augment or grayscale or semantics:
            raise NotImplementedError("augmentation / grayscale / semantics are outside the evaluation hot path")
        if not (coord or depth or normal):
            raise Exception("At least one 3D label should be enabled! Coord: {}, Depth: {}, Normal: {}".format(
                coord, depth, normal))
        self.coord, self.depth, self.normal = coord, depth, normal
        self.raw_image, self.image_height = raw_image, image_height
        roots = root_dir if isinstance(root_dir, list) else [root_dir]
        self.rgb_files, self.pose_files, self.calibration_files = [], [], []
        self.coord_files, self.depth_files, self.normal_files = [], [], []
        for base in roots:
            if not os.path.isdir(base):
                raise ValueError("root_dir type {} is not supported!".format(type(base)))
            self.rgb_files += _sorted_files(os.path.join(base, "rgb"))
            self.pose_files += _sorted_files(os.path.join(base, "poses"))
            self.calibration_files += _sorted_files(os.path.join(base, "calibration"))
            self.coord_files += _sorted_files(os.path.join(base, "init"))
            if depth:
                self.depth_files += _sorted_files(os.path.join(base, "depth"))
            if normal:
                self.normal_files += _sorted_files(os.path.join(base, "normal"))
        n = len(self.rgb_files)
        if not (len(self.pose_files) == n and len(self.calibration_files) == n):
            raise ValueError("rgb / poses / calibration directories hold different numbers of files")

    def __len__(self):
        return len(self.rgb_files)

    def __getitem__(self, idx):
        from PIL import Image
        img = Image.open(self.rgb_files[idx])
        if img.mode != "RGB":
            img = img.convert("RGB")                       # gray -> RGB, RGBA -> RGB (dataloader.py:303-307)
        focal = float(np.loadtxt(self.calibration_files[idx]))
        focal *= self.image_height / img.height            # :314-316
        if img.height != self.image_height:
            w = int(round(img.width * self.image_height / img.height))
            img = img.resize((w, self.image_height), Image.BILINEAR)
        image = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).float() / 255.0
        if not self.raw_image:
            image = (image - torch.tensor(MEAN)[:, None, None]) / torch.tensor(STD)[:, None, None]
        pose = torch.from_numpy(np.loadtxt(self.pose_files[idx])).float()
        labels = {}
        if self.coord:
            labels["coord"] = torch.load(self.coord_files[idx]).float()
        if self.depth:
            labels["depth"] = torch.load(self.depth_files[idx]).float().unsqueeze(0)
        if self.normal:
            labels["normal"] = torch.load(self.normal_files[idx]).float()
        gt = next(iter(labels.values())) if len(labels) == 1 else labels     # dict for several labels (:560-575)
        return image, pose, gt, focal, self.rgb_files[idx]


def write_synthetic_scene(root, count, seed=2021, noise=0.5, outlier_ratio=0.0):
    """Write `count` synthetic frames in the on-disk format above (tests and demos; images are noise)."""
    from PIL import Image
    from . import synth
    for sub in ("rgb", "poses", "calibration", "init"):
        os.makedirs(os.path.join(root, sub), exist_ok=True)
    rng = np.random.default_rng(seed)
    for i in range(count):
        sc = synth.make_scene(seed + i, noise=noise, outlier_ratio=outlier_ratio)
        name = "frame_%05d" % i
        Image.fromarray(rng.integers(0, 256, size=(synth.IMAGE_H, synth.IMAGE_W, 3), dtype=np.uint8)).save(
            os.path.join(root, "rgb", name + ".png"))
        np.savetxt(os.path.join(root, "poses", name + ".txt"), sc["pose"])
        np.savetxt(os.path.join(root, "calibration", name + ".txt"), [sc["focal"]])
        torch.save(torch.from_numpy(sc["gt_coords"]), os.path.join(root, "init", name + ".dat"))
    return root
