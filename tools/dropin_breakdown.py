"""Where a frame of the reference's literal loop (test_single_task.py:347-363 + utils/evaluation.py:156-172) spends its time: host
timers around each statement with a device synchronisation behind it (so the parts add up to MORE than the un-instrumented loop,
whose statements overlap a little).  python tools/dropin_breakdown.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import dsacstar  # noqa: E402
from crossloc_amd import networks, synth  # noqa: E402
from crossloc_amd.weights import seeded_state_dict  # noqa: E402

H, W, N = 480, 720, 64
net = networks.TransPoseNet(torch.tensor(synth.SCENE_MEAN, dtype=torch.float32), False, False, 2, 2, 3, 1)
net.load_state_dict(seeded_state_dict(net, seed=2021))
net = net.cuda().eval()
frames = torch.rand((N, 1, 3, H, W), generator=torch.Generator().manual_seed(64))
planted = torch.from_numpy(synth.make_batch(9000, N, noise=0.5, outlier_ratio=0.3)[0]).cuda()
acc = {}


def tick(name, t0):
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    acc[name] = acc.get(name, 0.0) + (t1 - t0)
    return t1


def one_pass(measure):
    with torch.no_grad():
        for i in range(N):
            t = time.perf_counter()
            x = frames[i].cuda()
            if measure: t = tick("image.cuda() (4.1 MB pageable H2D)", t)
            pred = net(x)
            if measure: t = tick("network forward (HIP graph, 88 launches)", t)
            pred, _ = torch.split(pred, [3, 1], dim=1)
            pred = pred.clone()
            pred.copy_(planted[i:i + 1])
            if measure: t = tick("split + clone + plant", t)
            out_pose = torch.zeros((4, 4))
            sc = pred.cpu()
            if measure: t = tick("scene_coords.cpu() (65 KB D2H)", t)
            dsacstar.forward_rgb(sc, out_pose, 256, 10.0, synth.FOCAL, float(W / 2), float(H / 2), 100.0, 100.0, 8)
            if measure: t = tick("dsacstar.forward_rgb (host call: H2D, 2 kernels, D2H)", t)


one_pass(False)
torch.cuda.synchronize()
t0 = time.perf_counter()
one_pass(False)
torch.cuda.synchronize()
print("un-instrumented loop: %.3f ms per frame" % ((time.perf_counter() - t0) / N * 1e3))
one_pass(True)
for k, v in acc.items():
    print("  %-58s %.3f ms" % (k, v / N * 1e3))
print("  sum %.3f ms" % (sum(acc.values()) / N * 1e3))
