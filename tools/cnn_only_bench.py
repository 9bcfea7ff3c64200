import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from crossloc_amd import networks
from crossloc_amd.weights import seeded_state_dict
net = networks.TransPoseNet(torch.tensor([-455.934, 417.50, 520.31]), False, False, 2, 2, 3, 1)
net.load_state_dict(seeded_state_dict(net, seed=2021))
net = net.cuda().eval()
x = torch.rand(95, 3, 480, 720, device="cuda")
s = torch.cuda.Stream()
with torch.cuda.stream(s), torch.no_grad():
    for _ in range(3): net(x)
    s.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(20): net(x)
    e1.record(s)
    s.synchronize()
print("cnn only: %.3f ms per 95 frames = %.1f images/s" % (e0.elapsed_time(e1) / 20, 95 * 20 / e0.elapsed_time(e1) * 1e3))
