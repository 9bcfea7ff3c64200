"""Latency legs of bench.py alone (batch 1 and 8 through PipelinedLocalizer, HIP graph), for A/B runs of the tile forms:
XL_TILE_FORM_1X1 / XL_TILE_FORM_WINO = 256 | 192 | 128 force a form; unset = the plan's own makespan estimate."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from crossloc_amd.weights import seeded_state_dict  # noqa: E402
from crossloc_amd import evaluation, networks, synth  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    H, W = 480, 720
    mean = torch.tensor(synth.SCENE_MEAN, dtype=torch.float32)
    net = networks.TransPoseNet(mean, False, False, 2, 2, 3, 1)
    net.load_state_dict(seeded_state_dict(net, seed=2021))
    net = net.to(dev).eval()
    res = {}
    for nb in [int(a) for a in sys.argv[1:]] or [1, 8]:
        imgs = torch.rand((nb, 3, H, W), generator=torch.Generator().manual_seed(nb)).to(dev)
        c_np, _, _ = synth.make_batch(7000, nb, noise=0.5, outlier_ratio=0.3)
        c_t = torch.from_numpy(c_np).to(dev)
        pipe = evaluation.PipelinedLocalizer(net, 256, synth.FOCAL, H, W)
        best = 1e9
        for rep in range(3):
            for _ in range(5):
                pipe.submit(imgs, image0=0, plant=c_t)
            pipe.finish()
            torch.cuda.synchronize()
            n_it = 200 if nb == 1 else 60
            t0 = time.perf_counter()
            for _ in range(n_it):
                pipe.submit(imgs, image0=0, plant=c_t)
            pipe.finish()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / n_it * 1e3)
        res[nb] = best
        del pipe
    print(os.environ.get("XL_TILE_FORM_1X1", "-"), os.environ.get("XL_TILE_FORM_WINO", "-"),
          " ".join("B=%d %.3f ms %.1f img/s" % (nb, ms, nb / ms * 1e3) for nb, ms in res.items()), flush=True)


if __name__ == "__main__":
    main()
