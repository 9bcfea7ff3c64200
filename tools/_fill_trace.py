import os, sys, torch, traceback, collections
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from crossloc_amd import loss as xl_loss, networks, synth, optim as xl_optim
from crossloc_amd.weights import seeded_state_dict
B = 4
dev = torch.device("cuda")
net = networks.TransPoseNet(torch.tensor(synth.SCENE_MEAN, dtype=torch.float32), False, False, 2, 2, 3, 1)
net.load_state_dict(seeded_state_dict(net, seed=2021))
net = net.to(dev).train()
images = torch.rand(B, 3, 480, 720, device=dev)
coords, gt, poses = synth.make_batch(1, B, noise=0.5, outlier_ratio=0.0)
gt_t, poses_t = torch.from_numpy(gt).to(dev), torch.from_numpy(poses.astype(np.float32)).to(dev)
grid, cam = xl_loss.get_pixel_grid(8), xl_loss.get_cam_mat(720, 480, 480.0)
opt = xl_optim.Adam(net.parameters(), lr=1e-4)
for _ in range(2):
    xl_optim.train_step(net, opt, images, poses_t, gt_t, grid, cam)
from torch.utils._python_dispatch import TorchDispatchMode
cnt = collections.Counter()
class M(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        st = traceback.extract_stack(limit=8)
        where = [f"{os.path.basename(f.filename)}:{f.lineno}" for f in st if "crossloc_amd" in f.filename][-2:]
        cnt[(name, tuple(where))] += 1
        return func(*args, **(kwargs or {}))
with M():
    xl_optim.train_step(net, opt, images, poses_t, gt_t, grid, cam)
for k, v in sorted(cnt.items(), key=lambda kv: -kv[1])[:60]:
    print(v, k)
