"""Reader for the CrossLoc on-disk scene format: evaluation path and training mini-batches.

Reference: dataloader/dataloader.py `CamLocDataset` (:60-586).  Directory layout (:232-247):
    <root>/rgb/*.png            RGB(A) frames
    <root>/poses/*.txt          4x4 cam->world matrix (np.loadtxt)
    <root>/calibration/*.txt    focal length in pixels of the stored frame
    <root>/init/*.dat           torch-saved [3,Ho,Wo] scene coordinates, nodata = -1   (coord task)
    <root>/depth/*.dat          torch-saved [Ho,Wo] z-buffer depth                      (depth task)
    <root>/normal/*.dat         torch-saved [3,Ho,Wo] surface normals                   (normal task)
    <root>/semantics/*.npy      numpy [H,W] raw class ids, full resolution              (semantics task, :337-338)
Files of one frame share their sorted position in each directory (:310-338).

Covered: mode=1 / sparse labels.
  augment=False (what utils/evaluation.py:20-78 builds for testing): frames are resized to `image_height` when they are
    not already that tall (PIL bilinear, as torchvision's Resize does on PIL images, :189-212) with the focal length
    scaled accordingly (:314-316); raw_image=True returns un-normalised [0,1] RGB, otherwise the urbanscape mean/std
    normalisation (:193-196).  Per item on the CPU, like the reference.
  augment=True, batch=True (what utils/learning.py:177-263 builds for training): `__getitem__` only decodes - it
    returns the uint8 HWC frame - and `collate_gpu` (the counterpart of `batch_resize`, :512-586, used as the
    DataLoader's collate_fn) uploads the frames once and runs resize, colour jitter, ToTensor, normalisation and the
    common scale + rotation of the mini-batch on the GPU (crossloc_amd/data.py, csrc/xl_data.hip).  Random draws follow
    the reference: one (brightness, contrast) pair per frame, one (scale, angle) pair per mini-batch from `random`.
Semantics labels (round 3): read with numpy, trimmed to the 6 training classes (loss.trim_semantic_label), float [1,H,W];
  collate_gpu resizes them to the IMAGE size with 'nearest' and rotates them with fill 0 (:540-543).  Grayscale (round 3):
  Resize -> Grayscale -> [ColorJitter] -> ToTensor -> Normalize(0.4308, 0.1724) (:171-187, 359-373), one channel.
Not covered: augment=True with batch=False (per-item scale / rotation with a rotated pose, :349-464: no CrossLoc
training script uses it), dense-depth initialisation (sparse=False), mode 0 / 2.
collate_gpu makes GPU calls: it must run in the main process (DataLoader(num_workers=0), or - to keep the reference's CPU
worker parallelism for PNG decoding, utils/learning.py:238-252 - workers with `collate_host` and `to_gpu` on the fetched
batch in the training loop).
"""
import os

import numpy as np
import torch
from torch.utils.data import Dataset

MEAN = (0.4245, 0.4375, 0.3836)        # dataloader.py:193-196
STD = (0.1823, 0.1701, 0.1854)
MEAN_GRAY, STD_GRAY = 0.4308, 0.1724   # dataloader.py:178-180


def _sorted_files(d):
    return sorted(os.path.join(d, f) for f in os.listdir(d)) if os.path.isdir(d) else []


class CamLocDataset(Dataset):
    def __init__(self, root_dir, mode=1, sparse=True, coord=True, depth=False, normal=False, semantics=False,
                 augment=False, grayscale=False, batch=True, raw_image=False, aug_rotation=30, aug_scale_min=2 / 3,
                 aug_scale_max=3 / 2, aug_contrast=0.1, aug_brightness=0.1, image_height=480, **unused):
        if mode != 1 or not sparse:
            raise NotImplementedError("only mode=1 with sparse labels is read (the configuration CrossLoc uses)")
        if raw_image:
            augment, grayscale = False, False              # dataloader.py:217-219: raw_image supersedes the rest
        if augment and not batch:
            raise NotImplementedError("augment=True needs batch=True (per-item scale / rotation is not reproduced)")
        self.augment, self.batch = augment, batch
        self.aug_rotation, self.aug_scale_min, self.aug_scale_max = aug_rotation, aug_scale_min, aug_scale_max
        self.aug_contrast, self.aug_brightness = aug_contrast, aug_brightness
        if not (coord or depth or normal or semantics):
            raise Exception("At least one 3D label should be enabled! Coord: {}, Depth: {}, Normal: {}".format(
                coord, depth, normal))
        self.coord, self.depth, self.normal, self.semantics, self.grayscale = coord, depth, normal, semantics, grayscale
        self.raw_image, self.image_height = raw_image, image_height
        roots = root_dir if isinstance(root_dir, list) else [root_dir]
        self.rgb_files, self.pose_files, self.calibration_files = [], [], []
        self.coord_files, self.depth_files, self.normal_files, self.semantics_files = [], [], [], []
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
            if semantics:
                self.semantics_files += _sorted_files(os.path.join(base, "semantics"))
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
        if self.augment:
            # training: decode only; collate_gpu does the rest on the GPU
            image = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy())
            return image, *self._labels(idx), focal, self.rgb_files[idx]
        th, tw = _resized_shape(img.height, img.width, self.image_height)   # torchvision Resize(int): smaller edge
        if (th, tw) != (img.height, img.width):
            img = img.resize((tw, th), Image.BILINEAR)
        if self.grayscale:
            img = img.convert("L")                         # transforms.Grayscale() after the resize (:174)
            image = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy())[None].float() / 255.0
            image = (image - MEAN_GRAY) / STD_GRAY
        else:
            image = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).float() / 255.0
            if not self.raw_image:
                image = (image - torch.tensor(MEAN)[:, None, None]) / torch.tensor(STD)[:, None, None]
        pose, gt = self._labels(idx)
        return image, pose, gt, focal, self.rgb_files[idx]

    def _labels(self, idx):
        pose = torch.from_numpy(np.loadtxt(self.pose_files[idx])).float()
        labels = {}
        if self.coord:
            labels["coord"] = torch.load(self.coord_files[idx]).float()
        if self.depth:
            labels["depth"] = torch.load(self.depth_files[idx]).float().unsqueeze(0)
        if self.normal:
            labels["normal"] = torch.load(self.normal_files[idx]).float()
        if self.semantics:
            from .loss import trim_semantic_label
            labels["semantics"] = torch.tensor(trim_semantic_label(np.load(self.semantics_files[idx])),
                                               dtype=torch.float).unsqueeze(0)          # [1, H, W] (:337-338)
        gt = next(iter(labels.values())) if len(labels) == 1 else labels     # dict for several labels (:560-575)
        return pose, gt

    @staticmethod
    def collate_host(batch):
        """A plain stacking collate_fn for DataLoader WORKERS (items produced with augment=True): CPU tensors only - uint8
        frames [B,Hs,Ws,3], poses, labels (tensor or dict), focal lengths, file names.  Pair with pin_memory=True and call
        `to_gpu` on the fetched batch in the training loop."""
        frames = torch.stack([it[0] for it in batch])
        poses = torch.stack([it[1] for it in batch])
        if isinstance(batch[0][2], dict):
            labels = {k: torch.stack([it[2][k] for it in batch]) for k in batch[0][2]}
        else:
            labels = torch.stack([it[2] for it in batch])
        return frames, poses, labels, [it[3] for it in batch], [it[4] for it in batch]

    def to_gpu(self, host_batch, device="cuda", output_subsample=8):
        """The GPU half of `batch_resize` (dataloader.py:512-586) on a batch stacked by `collate_host`: upload once as
        bytes, then resize, (grayscale,) colour jitter, ToTensor, normalisation and the common scale + rotation of the
        mini-batch as HIP kernels.  Main process only."""
        from . import data
        frames, poses, labels, focals, files = host_batch
        frames = frames.to(device, non_blocking=True)
        poses = poses.to(device, non_blocking=True)
        if isinstance(labels, dict):
            labels = {k: v.to(device, non_blocking=True) for k, v in labels.items()}
        else:
            labels = labels.to(device, non_blocking=True)
        jitter = [data.draw_jitter(self.aug_brightness, self.aug_contrast) for _ in files]
        images = data.prepare_images(frames, self.image_height, jitter=jitter, normalize=True, grayscale=self.grayscale)
        import random
        scale_factor = random.uniform(self.aug_scale_min, self.aug_scale_max)      # :525-526, one draw per mini-batch
        angle = random.uniform(-self.aug_rotation, self.aug_rotation)
        images, labels, focals = data.batch_resize(images, labels, list(focals), scale_factor, angle, output_subsample,
                                                   semantics=self.semantics)
        return images, poses, labels, torch.tensor(focals, dtype=torch.float64), files

    def collate_gpu(self, batch, device="cuda", output_subsample=8):
        """Counterpart of `batch_resize` (dataloader.py:512-586) for items produced with augment=True: returns
        (images [B,3|1,H',W'] float32, poses [B,4,4], labels tensor or dict, focal lengths float64 [B], file names) with
        images and labels on `device`.  All frames of a mini-batch must have the same stored size (they are stacked).
        A collate_fn for num_workers=0 ONLY: it makes GPU calls, and a forked DataLoader worker cannot initialise the GPU."""
        if torch.utils.data.get_worker_info() is not None:
            raise RuntimeError("CamLocDataset.collate_gpu runs GPU kernels and cannot be a DataLoader worker's collate_fn: "
                               "use collate_fn=dataset.collate_host in the workers and dataset.to_gpu(batch) in the "
                               "training loop (or num_workers=0)")
        return self.to_gpu(self.collate_host(batch), device, output_subsample)


def _resized_shape(h, w, image_height):
    """torchvision Resize(image_height) on an h x w frame: the smaller edge becomes image_height, the other one is scaled
    and TRUNCATED (int(image_height * long / short)) - the rule of csrc/xl_data.hip::xl_data_resized_shape, restated on
    the host so that the evaluation path needs no GPU library to open a dataset."""
    if (w <= h and w == image_height) or (h <= w and h == image_height):
        return h, w
    if w < h:
        return int(image_height * h / w), image_height
    return image_height, int(image_height * w / h)


def write_synthetic_scene(root, count, seed=2021, noise=0.5, outlier_ratio=0.0, semantics=False):
    """Write `count` synthetic frames in the on-disk format above (tests and demos; images are noise; `semantics`: also a
    full-resolution map of RAW class ids per frame)."""
    from PIL import Image
    from . import synth
    for sub in ("rgb", "poses", "calibration", "init") + (("semantics",) if semantics else ()):
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
        if semantics:
            raw = np.array([0, 1, 2, 3, 6, 9, 17])[rng.integers(0, 7, size=(synth.IMAGE_H // 16, synth.IMAGE_W // 16))]
            np.save(os.path.join(root, "semantics", name + ".npy"), np.kron(raw, np.ones((16, 16), raw.dtype)))
    return root
