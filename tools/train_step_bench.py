"""BASELINE configs[1]: batch-16 480x720 coord network forward + MLE coord loss + backward on one MI355X —
our HIP path vs PyTorch-ROCm eager (MIOpen) running the same graph (oracle/cnn_oracle.py on GPU tensors).
Not the headline metric; a parity-case timing recorded in DESIGN.md."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np   # noqa: E402
import torch         # noqa: E402

from crossloc_amd import loss as xl_loss, networks, synth   # noqa: E402
from crossloc_amd.weights import seeded_state_dict          # noqa: E402
from oracle import cnn_oracle, loss_oracle                  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--eager", action="store_true", help="also time PyTorch-ROCm eager")
    ap.add_argument("--full", action="store_true", help="the step as the reference runs it: + fused Adam + weight re-pack "
                                                        "(optim.train_step)")
    a = ap.parse_args()
    B = a.batch
    dev = torch.device("cuda")
    mean = torch.tensor(synth.SCENE_MEAN, dtype=torch.float32)
    net = networks.TransPoseNet(mean, False, False, 2, 2, 3, 1)
    net.load_state_dict(seeded_state_dict(net, seed=2021))
    net = net.to(dev).train()
    images = torch.rand(B, 3, 480, 720, device=dev)
    coords, gt, poses = synth.make_batch(1, B, noise=0.5, outlier_ratio=0.0)
    gt_t, poses_t = torch.from_numpy(gt).to(dev), torch.from_numpy(poses.astype(np.float32)).to(dev)
    grid, cam = xl_loss.get_pixel_grid(8), xl_loss.get_cam_mat(720, 480, 480.0)

    def step():
        net.zero_grad(set_to_none=True)
        pred = net(images)
        sc, unc = torch.split(pred, [3, 1], dim=1)                       # train_single_task.py:269
        loss, rate = xl_loss.scene_coords_regression_loss(0.1, 100.0, 1000.0, 50.0, "MLE", grid, -1, cam, sc, unc,
                                                          poses_t, gt_t)
        loss.backward()
        return loss

    if a.full:
        from crossloc_amd import optim as xl_optim
        opt = xl_optim.Adam(net.parameters(), lr=1e-4)
        fwd_bwd = step

        def step():                                                       # noqa: F811
            return xl_optim.train_step(net, opt, images, poses_t, gt_t, grid, cam)[0]
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / a.steps * 1e3
    gflop = 885.64 * B
    if os.environ.get("XL_BENCH_VERBOSE"):
        import ctypes
        L = networks._bind()
        L.xl_cnn_prof_begin.argtypes = [ctypes.c_int]
        L.xl_cnn_prof_end.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        plan = net._plans[(B, 480, 720, 0, True)]
        nf, nb = len(plan.op_array), len(plan.bwd_array)
        L.xl_cnn_prof_begin(nf + nb)
        step()
        torch.cuda.synchronize()
        cap = nf + nb
        idx, typ, tms = (ctypes.c_int32 * cap)(), (ctypes.c_int32 * cap)(), (ctypes.c_float * cap)()
        n = L.xl_cnn_prof_end(idx, typ, tms, cap)
        names = {0: "conv1", 1: "conv", 2: "gn_stats", 3: "gn_apply", 4: "head", 5: "wgrad", 6: "gnb_stats", 7: "gnb_apply",
                 8: "gnb_params", 9: "head_bwd", 10: "conv1_wgrad", 12: "wino_in", 13: "wino_out", 14: "duc_head", 15: "duc_head_bwd", 16: "wino_dy", 17: "wino_wfinal", 18: "gnb_final", 11: "gn_final"}
        tot = {}
        for i in range(n):
            arr = plan.op_array if i < nf else plan.bwd_array
            op = arr[idx[i]]
            key = names[typ[i]] + ("_dgrad" if (typ[i] == 1 and i >= nf) else "") + ("_bwd" if i >= nf and typ[i] != 1 else "")
            tot[key] = tot.get(key, 0.0) + tms[i]
            if i >= nf and tms[i] > float(os.environ.get("XL_BENCH_VERBOSE_MS", "1.0")):
                sys.stderr.write("bwd op %3d %-10s k%d s%d %4d->%4d %3dx%3d  %.3f ms\n" % (
                    idx[i], names[typ[i]], op.ksize, op.stride, op.Cin, op.Cout, op.Hi, op.Wi, tms[i]))
        sys.stderr.write("totals (ms): %s\n" % {k: round(v, 2) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])})
    print("HIP path: %.1f ms/step (B=%d) = %.1f TFLOP/s fwd+bwd, loss %.4f, peak mem %.1f GB" % (
        ms, B, gflop / ms, loss.item(), torch.cuda.max_memory_allocated() / 2**30))
    if a.eager:
        sd = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point and not k.endswith("mean"))
              for k, v in net.state_dict().items()}

        def estep():
            for v in sd.values():
                v.grad = None
            res = cnn_oracle.encoder_forward(sd, images, "encoder", 2, 32)
            pred = cnn_oracle.decoder_forward(sd, res, 2, 3, 1, 32)
            sc, unc = torch.split(pred, [3, 1], dim=1)
            # the reference loss in plain torch on the GPU
            l, _ = _torch_coord_loss(sc, unc, poses_t, gt_t)
            l.backward()
            return l
        for _ in range(2):
            estep()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            l = estep()
        torch.cuda.synchronize()
        ems = (time.perf_counter() - t0) / a.steps * 1e3
        print("PyTorch-ROCm eager: %.1f ms/step = %.1f TFLOP/s, loss %.4f  -> speedup %.2fx" % (
            ems, gflop / ems, l.item(), ems / ms))


def _torch_coord_loss(sc, unc, poses, gt):
    B = sc.shape[0]
    X, G = sc.reshape(B, 3, -1), gt.reshape(B, 3, -1)
    P = torch.linalg.inv(poses)[:, :3, :]
    one = torch.ones(B, 1, X.shape[2], device=sc.device)
    Xc, Gc = torch.bmm(P, torch.cat([X, one], 1)), torch.bmm(P, torch.cat([G, one], 1))
    d = torch.norm(Xc - Gc, dim=1)
    K = torch.tensor([[480.0, 0, 360.0], [0, 480.0, 240.0], [0, 0, 1.0]], device=sc.device)
    p = torch.bmm(K.expand(B, 3, 3), Xc)
    uv = p[:, :2] / torch.clamp(p[:, 2:], min=0.1)
    ys, xs = torch.meshgrid(torch.arange(60, device=sc.device) * 8.0 + 4, torch.arange(90, device=sc.device) * 8.0 + 4, indexing="ij")
    e = (uv - torch.stack([xs, ys]).reshape(1, 2, -1)).norm(dim=1).clamp(min=1e-7)
    g = (G == -1).sum(1) == 0
    m = ~(Xc[:, 2] < 0.1) & ~(e > 1000.0) & ~((d > 50.0) & g)
    ep = e * m
    lr = (ep * (ep <= 100.0)).clamp(min=1e-7) + torch.sqrt(100.0 * (ep * (ep > 100.0)).clamp(min=1e-7) + 1e-7).clamp(min=1e-7)
    s = unc.reshape(B, -1).clamp(min=1e-7)
    lu = 3.0 * torch.log(s) + d.square().clamp(min=1e-7) / (2.0 * s.square().clamp(min=1e-7))
    return (lu * g + lr).sum() / g.numel(), m


if __name__ == "__main__":
    main()
