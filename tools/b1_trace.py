"""Kernel timeline of one batch-1 step (CNN graph replay + solver): python tools/b1_trace.py run | python tools/b1_trace.py show <kernel_trace.csv>"""
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if sys.argv[1] == "run":
    import torch
    from crossloc_amd import evaluation, networks, synth
    from crossloc_amd.weights import seeded_state_dict
    nb = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    net = networks.TransPoseNet(torch.tensor([-455.934, 417.50, 520.31]), False, False, 2, 2, 3, 1)
    net.load_state_dict(seeded_state_dict(net, seed=2021))
    net = net.cuda().eval()
    imgs = torch.rand((nb, 3, 480, 720), generator=torch.Generator().manual_seed(nb)).cuda()
    c_t = torch.from_numpy(synth.make_batch(7000, nb, noise=0.5, outlier_ratio=0.3)[0]).cuda()
    pipe = evaluation.PipelinedLocalizer(net, 256, synth.FOCAL, 480, 720)
    for _ in range(30):
        pipe.submit(imgs, image0=0, plant=c_t)
    pipe.finish()
    torch.cuda.synchronize()
else:
    rows = sorted(csv.DictReader(open(sys.argv[2])), key=lambda r: int(r["Start_Timestamp"]))
    # the last complete step: from the last conv1 kernel backwards
    starts = [i for i, r in enumerate(rows) if "conv1_mfma" in r["Kernel_Name"]]
    a, b = starts[-2], starts[-1]
    t0 = int(rows[a]["Start_Timestamp"])
    prev_end = t0
    busy = 0
    for r in rows[a:b]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60]
        print("%8.1f us  +%6.1f gap  %7.1f us  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, name))
        busy += e - s
        prev_end = max(prev_end, e)
    print("step %.1f us, kernels %.1f us, %d launches" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3, busy / 1e3, b - a))
