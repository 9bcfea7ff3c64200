"""End-to-end output error of the inference network against the reference golden (tests/golden/net_forward.npz,
generated from the reference's own PyTorch code) for the three forms of the stride-1 3x3 layers: direct implicit GEMM,
Winograd F(2x2,3x3), F(4x4,3x3), F(6x6,3x3) (the default where it needs the fewest multiplies).  Run once per form (the
choice is read when a plan is built):
    XL_NO_WINOGRAD=1 python tools/winograd_error.py ; XL_WINOGRAD=2 python ... ; XL_WINOGRAD=4 python ... ; python ...
Also the 480x720 network against a float64 PyTorch-CPU evaluation of the same graph."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np   # noqa: E402
import torch         # noqa: E402

from crossloc_amd import networks, synth          # noqa: E402
from crossloc_amd.weights import seeded_state_dict  # noqa: E402
from oracle import cnn_oracle                       # noqa: E402

form = "direct" if os.environ.get("XL_NO_WINOGRAD") else "winograd F(%sx%s,3x3)" % ((os.environ.get("XL_WINOGRAD", "6"),) * 2)
MEAN = torch.tensor(synth.SCENE_MEAN, dtype=torch.float32)
gold = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "net_forward.npz"))
for tag, mlr in (("single", 0), ("mlr3", 3)):
    net = networks.TransPoseNet(MEAN, False, False, 2, 2, 3, 1, 32, mlr, 0, False)
    net.load_state_dict(seeded_state_dict(net, seed=2021), strict=True)
    net = net.cuda().eval()
    with torch.no_grad():
        y = net(torch.from_numpy(gold[tag + "_x"]).cuda()).cpu()
    ref = torch.from_numpy(gold[tag + "_y"])
    d = (y[:, :3] - ref[:, :3]).abs().max().item()
    scale = (ref[:, :3] - MEAN[None, :, None, None]).abs().max().item()
    print("%-22s %-6s golden %s: max |dcoord| %.3e m (%.2e of the output range), max rel dsigma %.2e" % (
        form, tag, tuple(y.shape), d, d / scale, ((y[:, 3] - ref[:, 3]).abs() / ref[:, 3].abs()).max().item()))
net = networks.TransPoseNet(MEAN, False, False, 2, 2, 3, 1)
net.load_state_dict(seeded_state_dict(net, seed=3))
x = torch.rand(1, 3, 480, 720, generator=torch.Generator().manual_seed(1))
sd64 = {k: v.double() for k, v in net.state_dict().items()}
ref = cnn_oracle.transposenet_forward(sd64, x.double(), 0, 2, 2)
y = net.cuda().eval()(x.cuda()).cpu().double()
d = (y[:, :3] - ref[:, :3]).abs().max().item()
scale = (ref[:, :3] - MEAN.double()[None, :, None, None]).abs().max().item()
print("%-22s 480x720 vs float64: max |dcoord| %.3e m (%.2e of the output range)" % (form, d, d / scale))
