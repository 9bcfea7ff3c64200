"""The solver alone: forward_rgb_batch on B synthetic frames, 256 hypotheses, HIP events.  usage: python tools/dsac_bench.py [B]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from crossloc_amd import synth  # noqa: E402
import dsacstar  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 47
coords = torch.from_numpy(synth.make_batch(2021, B, noise=0.5, outlier_ratio=0.3)[0]).cuda()
poses = torch.zeros(B, 4, 4, device="cuda")
for _ in range(3):
    dsacstar.forward_rgb_batch(coords, poses, 256, 10.0, synth.FOCAL, 360.0, 240.0, 100.0, 100.0, 8, image0=0)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    dsacstar.forward_rgb_batch(coords, poses, 256, 10.0, synth.FOCAL, 360.0, 240.0, 100.0, 100.0, 8, image0=0)
e1.record()
torch.cuda.synchronize()
print("solver %d frames x 256 hyps: %.4f ms  (pose checksum %.9e)" % (B, e0.elapsed_time(e1) / 20, poses.double().sum().item()))
