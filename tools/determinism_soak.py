"""Repeat the 47-frame inference forward and one training step N times and compare every result bitwise with the first
run (fixed-order reductions: any difference is a race).  python tools/determinism_soak.py [N=20]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from crossloc_amd import networks, synth  # noqa: E402
from crossloc_amd.weights import seeded_state_dict  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
mean = torch.tensor(synth.SCENE_MEAN, dtype=torch.float32)
net = networks.TransPoseNet(mean, False, False, 2, 2, 3, 1)
net.load_state_dict(seeded_state_dict(net, seed=2021))
net = net.cuda().eval()
x = torch.rand(47, 3, 480, 720, generator=torch.Generator().manual_seed(3)).cuda()
bad = 0
with torch.no_grad():
    ref = net(x).clone()
    for i in range(N):
        bad += int(not torch.equal(net(x), ref))
print("inference: %d of %d repeats differ" % (bad, N))
net.train()
xt = x[:8]
w = torch.randn(8, 4, 60, 90, generator=torch.Generator().manual_seed(4)).cuda()
grads = None
bad_t = 0
for i in range(max(2, N // 4)):
    net.zero_grad(set_to_none=True)
    (net(xt) * w).sum().backward()
    g = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
    if grads is None:
        grads = g.clone()
    else:
        bad_t += int(not torch.equal(g, grads))
print("training: %d of %d repeats differ" % (bad_t, max(2, N // 4) - 1))
sys.exit(1 if bad or bad_t else 0)
