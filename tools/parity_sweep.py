"""Large parity sweep of the HIP solver against the CPU restatement (oracle/): K frames of BASELINE configs[2]
(480x720 -> 60x90 grid, 256 hypotheses) over outlier ratios 0 / 0.3 / 0.6 — every pose bit-compared — and K/4 frames of
backward_rgb (expected loss + gradient bit-compared).  Prints one summary line per stage; exits non-zero on a mismatch.
    python tools/parity_sweep.py [K=1024]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np      # noqa: E402
import torch            # noqa: E402

from crossloc_amd import synth    # noqa: E402
import dsacstar                   # noqa: E402
from oracle import dsac_oracle as xo   # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
xo.set_num_threads(min(64, os.cpu_count() or 1))
bad = 0
B = 64
for rho in (0.0, 0.3, 0.6):
    n_img, worst_t, worst_r, t_gpu, t_cpu = 0, 0.0, 0.0, 0.0, 0.0
    for s in range(0, K // 3, B):
        coords, _, poses = synth.make_batch(9000 + int(rho * 10) * 1000 + s, B, noise=0.5, outlier_ratio=rho)
        co = torch.from_numpy(coords).cuda()
        out = torch.zeros((B, 4, 4), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize(); t0 = time.perf_counter()
        dsacstar.forward_rgb_batch(co, out, 256, 10.0, synth.FOCAL, 360.0, 240.0, 100.0, 100.0, 8, image0=s)
        torch.cuda.synchronize(); t_gpu += time.perf_counter() - t0
        got = out.cpu().numpy()
        t0 = time.perf_counter()
        for b in range(B):
            ref = xo.forward_rgb(coords[b], 256, 10.0, synth.FOCAL, 360.0, 240.0, 100.0, 100.0, 8, image=s + b)
            if not np.array_equal(got[b].view(np.int32), ref.view(np.int32)):
                bad += 1
            te, re = synth.pose_error(poses[b], got[b])
            worst_t, worst_r = max(worst_t, te), max(worst_r, re)
        t_cpu += time.perf_counter() - t0
        n_img += B
    print("forward_rgb rho=%.1f: %d frames, mismatches so far %d, worst error vs ground truth %.1f cm / %.3f deg, "
          "GPU %.2f ms/frame, CPU oracle %.2f ms/frame" % (rho, n_img, bad, worst_t * 100, worst_r, t_gpu / n_img * 1e3,
                                                           t_cpu / n_img * 1e3))
n_img, t_gpu, t_cpu = 0, 0.0, 0.0
for s in range(0, K // 4, 32):
    coords, _, poses = synth.make_batch(20000 + s, 32, noise=0.7, outlier_ratio=0.4)
    co = torch.from_numpy(coords).cuda()
    g = torch.zeros_like(co)
    gt = torch.from_numpy(poses.astype(np.float32)).cuda()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loss = dsacstar.backward_rgb_batch(co, g, gt, 256, 10.0, synth.FOCAL, 360.0, 240.0, 1.0, 1.0, 100.0, 100.0, 100.0, 8, 7,
                                       image0=s)
    torch.cuda.synchronize(); t_gpu += time.perf_counter() - t0
    lg, gg = loss.cpu().numpy(), g.cpu().numpy()
    t0 = time.perf_counter()
    for b in range(32):
        go = np.zeros_like(coords[b])
        lo = xo.backward_rgb(coords[b], go, poses[b], 256, 10.0, synth.FOCAL, 360.0, 240.0, 1.0, 1.0, 100.0, 100.0, 100.0, 8,
                             seed=7, image=s + b)
        if lo != float(lg[b]) or not np.array_equal(go.view(np.int32), gg[b].view(np.int32)):
            bad += 1
    t_cpu += time.perf_counter() - t0
    n_img += 32
print("backward_rgb: %d frames, total mismatches %d, GPU %.2f ms/frame, CPU oracle %.2f ms/frame" % (
    n_img, bad, t_gpu / n_img * 1e3, t_cpu / n_img * 1e3))
sys.exit(1 if bad else 0)
