"""Fused Adam and data-parallel helpers for the training step.

`Adam` is a torch.optim.Optimizer (so MultiStepLR, utils/learning.py:392-396, keeps working) whose step() is one
HIP launch over all parameter tensors (csrc/xl_optim.hip) instead of ~6 elementwise launches per tensor.
`allreduce_gradients` averages the gradients over the ranks of a node with ONE bucketed all-reduce
(26.8 M fp32 = 107 MB; ring over xGMI is per-link bound: ~2*(R-1)/R*107 MB / 153 GB/s = 1.2 ms at R = 8).
"""
import ctypes

import torch

from . import _lib

CHUNK = 65536


class _Chunk(ctypes.Structure):
    _fields_ = [("param", ctypes.c_void_p), ("grad", ctypes.c_void_p), ("exp_avg", ctypes.c_void_p),
                ("exp_avg_sq", ctypes.c_void_p), ("n", ctypes.c_int32), ("pad", ctypes.c_int32)]


def _bind():
    L = _lib.lib()
    if not hasattr(L, "_optim_bound"):
        L.xl_adam_step.restype = ctypes.c_int
        L.xl_adam_step.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_float] * 7 + [ctypes.c_void_p]
        L._optim_bound = True
    return L


class Adam(torch.optim.Optimizer):
    """torch.optim.Adam(params, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0) semantics, fused."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._tables = {}

    def _table(self, gi, group):
        """Device table of chunks for one param group; rebuilt when any pointer it holds changes - parameters,
        gradients, or the moment tensors (load_state_dict replaces the state tensors: a table keyed on parameters and
        gradients alone would keep updating the freed ones)."""
        ps = [p for p in group["params"] if p.grad is not None]

        def moments(p):
            st = self.state.get(p)
            if not st or "exp_avg" not in st:
                return (0, 0)
            return (st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr())
        sig = tuple((p.data_ptr(), p.grad.data_ptr()) + moments(p) for p in ps)
        cached = self._tables.setdefault(gi, {})      # a few tables per group: the network hands its gradients out in
        if sig in cached:                              # alternating buffers (networks._Plan.run_backward)
            return cached[sig]
        while len(cached) >= 4:                         # evict the oldest table (dicts keep insertion order)
            cached.pop(next(iter(cached)))
        rows = []
        for p in ps:
            if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous():
                raise RuntimeError("crossloc_amd.optim.Adam needs contiguous float32 GPU parameters and gradients")
            st = self.state[p]
            if len(st) == 0:
                st["step"] = 0
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            for k in ("exp_avg", "exp_avg_sq"):         # a loaded state may be strided / on another device or dtype
                m = st[k]
                if m.device != p.device or m.dtype != torch.float32 or not m.is_contiguous():
                    st[k] = m.to(device=p.device, dtype=torch.float32).contiguous()
            n = p.numel()
            for o in range(0, n, CHUNK):
                rows.append(_Chunk(p.data_ptr() + 4 * o, p.grad.data_ptr() + 4 * o, st["exp_avg"].data_ptr() + 4 * o,
                                   st["exp_avg_sq"].data_ptr() + 4 * o, min(CHUNK, n - o), 0))
        host = (_Chunk * len(rows))(*rows)
        dev = torch.empty(ctypes.sizeof(host), dtype=torch.uint8, device=ps[0].device)
        dev.copy_(torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8))
        sig = tuple((p.data_ptr(), p.grad.data_ptr()) + moments(p) for p in ps)      # (the state may just have been made: the
        cached[sig] = (dev, len(rows))                                               #  pre-state signature is never a key)
        return dev, len(rows)

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._tables = {}

    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        self._tables = {}

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        L = _bind()
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            dev, n = self._table(gi, group)
            for p in ps:
                self.state[p]["step"] += 1
            t = int(self.state[ps[0]]["step"])
            b1, b2 = group["betas"]
            with torch.cuda.device(ps[0].device):
                stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
                _lib.check(L.xl_adam_step(ctypes.c_void_p(dev.data_ptr()), n, float(group["lr"]), float(b1), float(b2),
                                          float(group["eps"]), float(group["weight_decay"]), 1.0 - b1 ** t,
                                          1.0 - b2 ** t, stream))
            # the parameters were modified through raw pointers: bump their autograd version counters so that
            # consumers tracking `_version` (TransPoseNet re-packs its conv operands) notice, without a launch
            torch.autograd.graph.increment_version(ps)
        return loss


def allreduce_gradients(params, world_size, group=None):
    """Average `.grad` over the ranks with one flat all-reduce (NCCL=RCCL on GPUs, gloo on CPU tensors).  A single rank
    returns at once unless a process group is passed explicitly (then the collective runs: a one-rank RCCL group is how
    the path is exercised on a one-GPU box)."""
    if world_size <= 1 and group is None:
        return
    import torch.distributed as dist
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat.div_(world_size)
    o = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[o:o + n].view_as(g))
        o += n


def train_step(network, optimizer, images, gt_poses, gt_coords, pixel_grid, cam_mat, world_size=1,
               min_depth=0.1, soft_clamp=100.0, hard_clamp=1000.0, init_tolerance=50.0, uncertainty='MLE',
               nodata_value=-1):
    """One iteration of train_single_task.py:245-301 for the coord task on the HIP path: forward, split of the
    uncertainty channel (:269), fused coordinate loss, backward, (gradient all-reduce), fused Adam."""
    from . import loss as xl_loss
    optimizer.zero_grad(set_to_none=True)             # (no fill kernels; backward hands fresh gradient tensors out)
    pred = network(images)
    nt = network.num_task_channel
    sc, unc = torch.split(pred, [nt, network.num_pos_channel], dim=1)
    loss, rate = xl_loss.scene_coords_regression_loss(min_depth, soft_clamp, hard_clamp, init_tolerance, uncertainty,
                                                      pixel_grid, nodata_value, cam_mat, sc, unc, gt_poses, gt_coords)
    loss.backward()
    allreduce_gradients([p for p in network.parameters()], world_size)
    optimizer.step()
    return loss.detach(), rate
