"""How much of a single-frame step is host time: enqueue-only time of 200 submits vs the time until the GPU has finished them."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from crossloc_amd import evaluation, networks, synth  # noqa: E402
from crossloc_amd.weights import seeded_state_dict  # noqa: E402

dev = torch.device("cuda:0")
H, W = 480, 720
net = networks.TransPoseNet(torch.tensor(synth.SCENE_MEAN, dtype=torch.float32), False, False, 2, 2, 3, 1)
net.load_state_dict(seeded_state_dict(net, seed=2021))
net = net.to(dev).eval()
for nb in (1, 8):
    imgs = torch.rand((nb, 3, H, W), generator=torch.Generator().manual_seed(nb)).to(dev)
    c_np, _, _ = synth.make_batch(7000, nb, noise=0.5, outlier_ratio=0.3)
    c_t = torch.from_numpy(c_np).to(dev)
    pipe = evaluation.PipelinedLocalizer(net, 256, synth.FOCAL, H, W)
    for _ in range(10):
        pipe.submit(imgs, image0=0, plant=c_t)
    pipe.finish(); torch.cuda.synchronize()
    for rep in range(3):
        n = 200
        t0 = time.perf_counter()
        for _ in range(n):
            pipe.submit(imgs, image0=0, plant=c_t)
        t1 = time.perf_counter()
        pipe.finish(); torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("B=%d enqueue %.3f ms/step, total %.3f ms/step" % (nb, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3), flush=True)
    # the CNN alone (no solver, no plant copy)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st), torch.no_grad():
        for _ in range(5):
            net(imgs)
        st.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            net(imgs)
        t1 = time.perf_counter()
        st.synchronize()
        t2 = time.perf_counter()
    print("B=%d CNN only: enqueue %.3f ms/step, total %.3f ms/step" % (nb, (t1 - t0) / 200 * 1e3, (t2 - t0) / 200 * 1e3), flush=True)
