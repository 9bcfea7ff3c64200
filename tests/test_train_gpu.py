"""GPU: fused Adam against torch.optim.Adam, and the whole training step (forward + fused coord loss + backward +
Adam, train_single_task.py:245-301) on the HIP path: the loss must go down and match a PyTorch-CPU replica."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from crossloc_amd import loss as xl_loss, networks, optim, synth     # noqa: E402
from crossloc_amd.weights import seeded_state_dict                   # noqa: E402
from oracle import cnn_oracle, loss_oracle                           # noqa: E402

pytestmark = pytest.mark.gpu
MEAN = torch.tensor(synth.SCENE_MEAN, dtype=torch.float32)


def test_fused_adam_matches_torch_adam():
    g = torch.Generator().manual_seed(0)
    shapes = [(512, 512, 3, 3), (512,), (70001,), (4, 512, 1, 1), (1,)]
    ours = [torch.randn(s, generator=g).cuda().requires_grad_(True) for s in shapes]
    ref = [p.detach().clone().requires_grad_(True) for p in ours]
    oa = optim.Adam(ours, lr=1e-3)
    ob = torch.optim.Adam(ref, lr=1e-3)
    sched_a = torch.optim.lr_scheduler.MultiStepLR(oa, [2], gamma=0.5)          # utils/learning.py:392-396
    sched_b = torch.optim.lr_scheduler.MultiStepLR(ob, [2], gamma=0.5)
    for step in range(4):
        for a, b in zip(ours, ref):
            gr = torch.randn(a.shape, generator=g).cuda() * (10.0 ** (step - 2))
            a.grad, b.grad = gr.clone(), gr.clone()
        v0 = ours[0]._version
        oa.step(); ob.step(); sched_a.step(); sched_b.step()
        assert ours[0]._version > v0
        for a, b in zip(ours, ref):
            assert torch.allclose(a, b, rtol=2e-6, atol=1e-7)
    st = oa.state[ours[0]]
    assert st["step"] == 4 and torch.allclose(st["exp_avg"], ob.state[ref[0]]["exp_avg"], rtol=1e-5, atol=1e-6)


def test_training_steps_reduce_loss_and_track_cpu_replica():
    B, H, W = 2, 64, 96
    net = networks.TransPoseNet(MEAN, False, False, 1, 1, 3, 1)
    sd0 = seeded_state_dict(net, seed=5)
    net.load_state_dict(sd0)
    coords, gt, poses = synth.make_batch(10, B, noise=0.0, outlier_ratio=0.0, Ho=H // 8, Wo=W // 8, focal=60.0)
    images = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(2))
    gt_t, poses_t = torch.from_numpy(gt), torch.from_numpy(poses.astype(np.float32))

    # PyTorch-CPU replica of the same three steps (reference graph + reference loss + torch Adam)
    sd = {k: v.clone().requires_grad_(v.dtype.is_floating_point and not k.endswith("mean")) for k, v in sd0.items()}
    leaves = [v for v in sd.values() if v.requires_grad]
    oref = torch.optim.Adam(leaves, lr=1e-4)
    ref_losses = []
    for _ in range(3):
        oref.zero_grad()
        pred = cnn_oracle.decoder_forward(sd, cnn_oracle.encoder_forward(sd, images, "encoder", 1, 32), 1, 3, 1, 32)
        l, _ = loss_oracle.coord_loss(pred[:, :3], pred[:, 3:], poses_t, gt_t, 60.0, W / 2, H / 2, 8.0)
        l.backward()
        oref.step()
        ref_losses.append(l.item())

    net = net.cuda().train()
    opt = optim.Adam(net.parameters(), lr=1e-4)                                   # lr of the finetune scripts
    grid, cam = xl_loss.get_pixel_grid(8), xl_loss.get_cam_mat(W, H, 60.0)
    losses = []
    for _ in range(3):
        l, rate = optim.train_step(net, opt, images.cuda(), poses_t.cuda(), gt_t.cuda(), grid, cam)
        losses.append(l.item())
    assert losses[2] < losses[0]
    assert np.allclose(losses, ref_losses, rtol=2e-3), (losses, ref_losses)
    # after the steps the in-place updated weights are the ones the next forward uses
    with torch.no_grad():
        y = net(images.cuda()).cpu()
    yref = cnn_oracle.transposenet_forward({k: v.detach() for k, v in net.state_dict().items()}, images, 0, 1, 1)
    assert (y[:, :3] - yref[:, :3]).abs().max().item() < 1e-2


def test_adam_resume_from_state_dict_uses_the_loaded_moments():
    """load_state_dict replaces the moment tensors: the cached device table of raw pointers must be rebuilt (it used
    to keep updating the freed ones and ignore the loaded moments)."""
    g = torch.Generator().manual_seed(3)
    shapes = [(64, 32, 3, 3), (64,), (70001,)]
    ours = [torch.randn(s, generator=g).cuda().requires_grad_(True) for s in shapes]
    ref = [p.detach().clone().requires_grad_(True) for p in ours]
    oa, ob = optim.Adam(ours, lr=1e-3), torch.optim.Adam(ref, lr=1e-3)

    def step(n):
        for _ in range(n):
            for a, b in zip(ours, ref):
                gr = torch.randn(a.shape, generator=g).cuda()
                a.grad.copy_(gr) if a.grad is not None else setattr(a, "grad", gr.clone())
                b.grad = gr.clone()
            oa.step(); ob.step()
    step(2)
    saved = {"a": oa.state_dict(), "b": ob.state_dict()}
    saved = {k: {"state": {i: {n: (v.clone() if torch.is_tensor(v) else v) for n, v in st.items()}
                           for i, st in sd["state"].items()}, "param_groups": sd["param_groups"]}
             for k, sd in saved.items()}
    step(2)                                                  # moves the moments away from the saved ones
    with torch.no_grad():
        for a, b in zip(ours, ref):
            b.copy_(a)
    oa.load_state_dict(saved["a"]); ob.load_state_dict(saved["b"])
    step(3)
    for a, b in zip(ours, ref):
        assert torch.allclose(a, b, rtol=2e-6, atol=1e-7)
    for a, b in zip(ours, ref):
        assert torch.allclose(oa.state[a]["exp_avg"], ob.state[b]["exp_avg"], rtol=1e-5, atol=1e-7)
        assert torch.allclose(oa.state[a]["exp_avg_sq"], ob.state[b]["exp_avg_sq"], rtol=1e-4, atol=1e-9)   # fma rounding
    assert int(oa.state[ours[0]]["step"]) == int(ob.state[ref[0]]["step"]) == 5


def test_second_training_forward_before_backward_keeps_the_first_graph_intact():
    """Two grad-enabled forwards of one shape before any backward (gradient accumulation over two forwards): the second
    gets a plan of its own; both backward passes see their own activations."""
    net = networks.TransPoseNet(MEAN, False, False, 0, 0, 3, 1)
    net.load_state_dict(seeded_state_dict(net, seed=8))
    net = net.cuda().train()
    xa = torch.rand(1, 3, 64, 96, generator=torch.Generator().manual_seed(1)).cuda()
    xb = torch.rand(1, 3, 64, 96, generator=torch.Generator().manual_seed(2)).cuda()

    def grads(order):
        net.zero_grad(set_to_none=True)
        if order == "interleaved":
            ya, yb = net(xa), net(xb)
            ya.sum().backward(); yb.sum().backward()
        else:
            net(xa).sum().backward(); net(xb).sum().backward()
        return [p.grad.clone() for p in net.parameters()]
    seq, inter = grads("sequential"), grads("interleaved")
    for a, b in zip(seq, inter):
        assert torch.equal(a, b)
    # the graph of a dropped forward releases its plan; inference under no_grad never takes one
    for _ in range(6):
        net(xa)
    with torch.no_grad():
        net(xa)


def test_gradient_tensors_held_by_the_caller_are_never_overwritten():
    """The backward pass hands its gradients out as views of alternating result buffers.  A view that is still alive
    anywhere - kept across zero_grad(set_to_none=True) for logging or gradient surgery, or returned by torch.autograd.grad -
    pins its buffer: later backward passes take another one."""
    net = networks.TransPoseNet(MEAN, False, False, 0, 0, 3, 1)
    net.load_state_dict(seeded_state_dict(net, seed=8))
    net = net.cuda().train()
    xs = [torch.rand(1, 3, 64, 96, generator=torch.Generator().manual_seed(i)).cuda() for i in range(5)]
    net(xs[0]).sum().backward()
    held = [p.grad for p in net.parameters()]                     # references only, no clone
    snap = [g.clone() for g in held]
    ag = torch.autograd.grad(net(xs[1]).sum(), [net.decoder.fc3.weight, net.encoder.conv2.weight])
    ag_snap = [g.clone() for g in ag]
    for x in xs[2:]:                                              # three more backward passes: both buffers had their turn
        net.zero_grad(set_to_none=True)
        net(x).sum().backward()
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(held, snap))
    assert all(torch.equal(a, b) for a, b in zip(ag, ag_snap))
    # ... and with nothing held, the two buffers alternate (stable addresses for the fused optimizer's pointer table)
    del held, ag
    ptrs = []
    for x in xs + xs[:3]:
        net.zero_grad(set_to_none=True)
        net(x).sum().backward()
        ptrs.append(net.decoder.fc3.weight.grad.data_ptr())
    assert len(set(ptrs[-4:])) == 2 and ptrs[-1] == ptrs[-3] and ptrs[-2] == ptrs[-4]


def test_rccl_collectives_on_a_one_rank_group():
    """The two data-path collectives of the N > 1 runs - the flat gradient all-reduce (optim.allreduce_gradients) and the
    NaN-padded all-gather of per-image errors (evaluation.gather_errors) - through the `nccl` backend, which is RCCL on
    ROCm, on device tensors.  A one-GPU box cannot host two ranks (RCCL refuses two ranks on one device), so the group has
    one rank: what is checked is that the RCCL path runs on these tensors and leaves the values it must leave."""
    import os
    import socket
    import torch.distributed as dist
    from crossloc_amd import evaluation, optim
    if dist.is_initialized():
        pytest.skip("a default process group already exists in this process")
    with socket.socket() as sck:
        sck.bind(("127.0.0.1", 0))
        port = sck.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        assert dist.get_backend() == "nccl"
        g = dist.group.WORLD
        params = [torch.nn.Parameter(torch.randn(n, device="cuda")) for n in (7, 1024, 3 * 3 * 512)]
        for p in params:
            p.grad = torch.randn_like(p)
        before = [p.grad.clone() for p in params]
        optim.allreduce_gradients(params, 1, group=g)
        torch.cuda.synchronize()
        assert all(torch.equal(a, p.grad) for a, p in zip(before, params))
        vals = torch.rand(5, 2, device="cuda")
        out = evaluation.gather_errors(vals, 5, 0, 1, group=g)
        assert torch.equal(out, vals)
    finally:
        dist.destroy_process_group()


def test_backward_launch_reductions_do_not_change_a_bit(monkeypatch):
    """Round 5's training-step launch reductions against the forms they replace, same network, same batch: conv1's GroupNorm-backward
    apply folded into its weight-gradient kernel (XL_NO_CONV1_WGRAD_FOLD=1: the apply pass), the GroupNorm parameter gradients of
    the pass in one launch (XL_GNB_PARAMS_PER_LAYER=1: one launch per layer) and the batched re-pack (covered in test_pair_gpu) -
    every parameter gradient (90 tensors of the 1 + 1 extra-block network) is bitwise the same."""
    def grads(env):
        for k in ("XL_NO_CONV1_WGRAD_FOLD", "XL_GNB_PARAMS_PER_LAYER"):
            monkeypatch.delenv(k, raising=False)
        for k in env:
            monkeypatch.setenv(k, "1")
        net = networks.TransPoseNet(torch.tensor(synth.SCENE_MEAN, dtype=torch.float32), False, False, 1, 1, 3, 1)
        net.load_state_dict(seeded_state_dict(net, 31))
        net = net.cuda().train()
        x = torch.rand(2, 3, 128, 192, generator=torch.Generator().manual_seed(9)).cuda()
        pred = net(x)
        w = torch.rand(pred.shape, generator=torch.Generator().manual_seed(10)).cuda()
        (pred * w).sum().backward()
        plan = [p for p in net._plans.values() if p.train][0]
        folded = any(op.type == networks.XL_OP_CONV1_WGRAD and op.aux2 for op in plan.bwd_array)
        listed = any(op.type == networks.XL_OP_GNB_PARAMS_LIST for op in plan.bwd_array)
        return {n: p.grad.clone() for n, p in net.named_parameters()}, folded, listed
    new, folded, listed = grads(())
    old, folded0, listed0 = grads(("XL_NO_CONV1_WGRAD_FOLD", "XL_GNB_PARAMS_PER_LAYER"))
    assert folded and listed and not folded0 and not listed0
    assert len(new) == 90
    for n in new:
        assert torch.equal(new[n], old[n]), n
