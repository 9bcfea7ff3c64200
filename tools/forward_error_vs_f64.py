import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests", "golden"))
import golden_inputs
from crossloc_amd import networks
from crossloc_amd.weights import seeded_state_dict
F = np.load(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests/golden/full_size.npz"))
MEAN = torch.tensor([-455.934, 417.50, 520.31])
for tag, mlr in (("single", 0), ("mlr3", 3)):
    net = networks.TransPoseNet(MEAN, False, False, 2, 2, 3, 1, 32, mlr, 0, False)
    net.load_state_dict(seeded_state_dict(net, seed=2021))
    net = net.cuda().eval()
    with torch.no_grad():
        y = net(torch.from_numpy(golden_inputs.full_size_image(tag)).cuda()).cpu().double().numpy()
    y64, y32 = F[tag + "_y64"], F[tag + "_y"].astype(np.float64)
    for nm, a in (("hip", y), ("ref32", y32)):
        e = np.abs(a - y64)
        print(tag, nm, "coord max %.3e median %.3e mean %.3e | unc rel max %.3e median %.3e" % (e[:, :3].max(), np.median(e[:, :3]), e[:, :3].mean(),
              (e[:, 3] / np.abs(y64[:, 3])).max(), np.median(e[:, 3] / np.abs(y64[:, 3]))))
