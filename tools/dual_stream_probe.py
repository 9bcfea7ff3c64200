"""Experiment: two half-batches of the CNN forward on two HIP streams (MFMA-bound GEMMs of one half overlapping the
HBM-bound transforms / GroupNorm passes of the other) against one full batch on one stream."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crossloc_amd import networks, synth
from crossloc_amd.weights import seeded_state_dict

B = int(sys.argv[1]) if len(sys.argv) > 1 else 24
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 2
mean = torch.tensor(synth.SCENE_MEAN, dtype=torch.float32)
net = networks.TransPoseNet(mean, False, False, 2, 2, 3, 1)
net.load_state_dict(seeded_state_dict(net, seed=2021))
net = net.cuda().eval()
x = torch.rand(B, 3, 480, 720, device="cuda")
streams = [torch.cuda.Stream() for _ in range(NS)]
parts = list(torch.chunk(x, NS, dim=0))
with torch.no_grad():
    ref = net(x)
    for _ in range(2):
        net(x)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10):
        y = net(x)
    torch.cuda.synchronize()
    t1 = (time.perf_counter() - t) / 10
    # per-stream plans are distinct objects only if the batch size differs; same-size halves share one plan (and its
    # buffers!), so give each stream its own network replica sharing the parameters
    nets = [net]
    for _ in range(NS - 1):
        r = networks.TransPoseNet(mean, False, False, 2, 2, 3, 1)
        r.load_state_dict(net.state_dict())
        nets.append(r.cuda().eval())
    def run():
        outs = []
        for s, n, p in zip(streams, nets, parts):
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                outs.append(n(p))
        for s in streams:
            torch.cuda.current_stream().wait_stream(s)
        return outs
    for _ in range(2):
        outs = run()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10):
        outs = run()
    torch.cuda.synchronize()
    t2 = (time.perf_counter() - t) / 10
    y2 = torch.cat(outs, 0)
print("B=%d: one stream %.2f ms (%.0f img/s) | %d streams %.2f ms (%.0f img/s) | max diff %.3g" % (
    B, t1 * 1e3, B / t1, NS, t2 * 1e3, B / t2, (y2 - ref).abs().max().item()))
