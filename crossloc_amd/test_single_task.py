"""Evaluation entry point shaped like the reference's test_single_task.py (main loop :328-366, report
utils/evaluation.py:193-244) for the coord task, on the MI355X path:

    python -m crossloc_amd.test_single_task --synthetic 256 --hypotheses 256 [--network_in model.net]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m crossloc_amd.test_single_task ...

Differences from the reference loop: images are localised in batches (CNN forward + HIP dsacstar per batch)
instead of one at a time with a .cpu() round trip; with WORLD_SIZE > 1 image i goes to rank i % R and one
RCCL all-gather collects the per-image errors; dataset/checkpoint discovery (test_single_task.py:118-256)
is out of scope — frames come from crossloc_amd.synth, weights from --network_in (a reference state_dict) or
the seeded generator.  Without trained weights the network output is not a scene, so `--solver_input
synthetic` (default) feeds the solver the synthetic scene-coordinate maps while the CNN still runs.
"""
import argparse
import os
import random
import time

import numpy as np
import torch

from . import evaluation, networks, synth
from .weights import seeded_state_dict


def set_random_seed(random_seed):
    """utils/learning.py:74-81"""
    torch.manual_seed(random_seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(random_seed)
    random.seed(random_seed)
    np.random.seed(random_seed)


def _parse():
    p = argparse.ArgumentParser(description="CrossLoc coord-task evaluation on MI355X")
    p.add_argument('--hypotheses', '-hyps', type=int, default=64)            # test_single_task.py:76-89 defaults
    p.add_argument('--threshold', '-t', type=float, default=10)
    p.add_argument('--inlieralpha', '-ia', type=float, default=100)
    p.add_argument('--maxpixelerror', '-maxerrr', type=float, default=100)
    p.add_argument('--network_in', type=str, default=None, help='reference-format state_dict (.net)')
    p.add_argument('--synthetic', type=int, default=64, help='number of synthetic frames')
    p.add_argument('--batch', type=int, default=16)
    p.add_argument('--noise', type=float, default=0.5)
    p.add_argument('--outliers', type=float, default=0.3)
    p.add_argument('--solver_input', choices=['synthetic', 'network', 'planted', 'labels'], default='synthetic',
                   help="what the solver consumes: synthetic scene maps (side input), the network output, the network "
                        "output tensor with the synthetic scene written into its coordinate channels after the head "
                        "(`planted`: the CNN -> solver hand-off of `network` with untrained weights), or the ground-truth "
                        "labels (the `predictions = gt_label  # debug only!` switch of test_single_task.py:361)")
    p.add_argument('--scene_dir', type=str, default=None,
                   help='CrossLoc on-disk scene section (rgb/ poses/ calibration/ init/), read by crossloc_amd.dataset')
    p.add_argument('--num_mlr', type=int, default=0, help='3 = CrossLoc three-encoder network')
    p.add_argument('--testing_log', type=str, default=None)
    return p.parse_args()


def main():
    opt = _parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
    set_random_seed(2021)                                                       # test_single_task.py:265

    mean = torch.tensor(synth.SCENE_MEAN, dtype=torch.float32)
    net = networks.TransPoseNet(mean, False, False, 2, 2, 3, 1, 32, opt.num_mlr, 0, False)   # evaluation.py:105-109
    # a frame's result must not depend on the batch (hence on the rank count) it lands in: per-image statistics passes
    net.batch_invariant = True
    if opt.network_in:
        net.load_state_dict(torch.load(opt.network_in, map_location="cpu"), strict=True)     # evaluation.py:113
    else:
        net.load_state_dict(seeded_state_dict(net, seed=2021))
    net = net.to(dev).eval()

    ds = None
    if opt.scene_dir:
        from .dataset import CamLocDataset
        ds = CamLocDataset(opt.scene_dir, mode=1, sparse=True, coord=True, raw_image=True)      # evaluation.py:33-44
        if opt.solver_input == 'synthetic':
            opt.solver_input = 'network'
    K = len(ds) if ds is not None else opt.synthetic
    mine = evaluation.shard_indices(K, rank, world)
    H, W = synth.IMAGE_H, synth.IMAGE_W
    rows, coord_errs = [], []
    t0 = time.time()
    for s in range(0, len(mine), opt.batch):
        idx = mine[s:s + opt.batch]
        if ds is not None:
            items = [ds[i] for i in idx]
            images = torch.stack([it[0] for it in items]).to(dev)
            gt_pose = torch.stack([it[1] for it in items]).to(dev)
            gt_coords = torch.stack([it[2] for it in items]).to(dev)
            # per frame, like the reference's batch-1 loop: sections may mix cameras / stored image heights
            focal = torch.tensor([float(it[3]) for it in items], dtype=torch.float32)
            coords = gt_coords
            H, W = images.shape[2], images.shape[3]
        else:
            scenes = [synth.make_scene(2021 + i, noise=opt.noise, outlier_ratio=opt.outliers) for i in idx]
            # raw_image=True: un-normalised [0,1].  Seeded per GLOBAL image index, so a frame's input (hence its
            # network output) does not depend on the rank count or the batch it lands in
            images = torch.stack([torch.rand((3, H, W), generator=torch.Generator().manual_seed(77000 + i))
                                  for i in idx]).to(dev)
            coords = torch.from_numpy(np.stack([sc["coords"] for sc in scenes])).to(dev)
            gt_pose = torch.from_numpy(np.stack([sc["pose"] for sc in scenes])).to(dev)
            gt_coords = torch.from_numpy(np.stack([sc["gt_coords"] for sc in scenes])).to(dev)
            focal = synth.FOCAL
        # image0/stride reproduce the i % R sharding in the sampler key
        poses, pred = evaluation.localize_batch(net, images, opt.hypotheses, focal, H, W, image0=idx[0],
                                                image_stride=world, threshold=opt.threshold,
                                                inlier_alpha=opt.inlieralpha, max_pixel_error=opt.maxpixelerror,
                                                scene_coords=coords if opt.solver_input in ('synthetic', 'labels') else None,
                                                plant=coords if opt.solver_input == 'planted' else None)
        t_err, r_err = evaluation.pose_errors(gt_pose, poses)
        rows.append(torch.stack([t_err, r_err], 1))
        used = pred[:, :3] if opt.solver_input in ('network', 'planted') else coords
        mask = evaluation.pick_valid_points(gt_coords.flatten(2), synth.NODATA)
        coord_errs.append(torch.norm(gt_coords.flatten(2) - used.flatten(2), dim=1)[mask].cpu())
    torch.cuda.synchronize()
    local = torch.cat(rows, 0) if rows else torch.empty((0, 2), dtype=torch.float64, device=dev)
    allv = evaluation.gather_errors(local, K, rank, world).cpu().numpy()
    elapsed = time.time() - t0
    if rank == 0:
        print("Localised %d frames on %d GPU(s) in %.2f s (%.1f images/s incl. scene generation)" % (
            K, world, elapsed, K / elapsed))
        evaluation.scene_coords_printout(allv[:, 0], allv[:, 1], None, [torch.cat(coord_errs).numpy()],
                                         testing_log=opt.testing_log)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
