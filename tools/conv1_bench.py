"""Timing of the two forms of the fused first layer (statistics pass / normalise + write pass) at the bench batch:
python tools/conv1_bench.py [B=47]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from crossloc_amd import networks  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 47
H, W = 480, 720
L = networks._bind()
g = torch.Generator().manual_seed(1)
x = torch.rand(B, 3, H, W, generator=g).cuda()
wt = torch.randn(32, 3, 3, 3, generator=g) * 0.2
wd = wt.permute(2, 3, 1, 0).contiguous().cuda()
wfr = networks._Plan.conv1_fragments(wt.cuda())
bd = torch.randn(32, generator=g).cuda()
coeff = torch.rand(B, 32, 2, generator=g).cuda()
out = torch.empty(B, H, W, 32, device="cuda")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for ppt in (5, 0):
    nch = -(-(H * W) // (256 * ppt)) if ppt else -(-H // 16) * -(-W // 64)
    stats = torch.zeros(B, nch, 32, 2, dtype=torch.float64, device="cuda")
    for which in ("stats", "write"):
        op = networks.XlOp()
        op.type = networks.XL_OP_CONV1
        op.B, op.Hi, op.Wi, op.Cin, op.Ho, op.Wo, op.Cout, op.ld_out = B, H, W, 3, H, W, 32, 32
        op.groups, op.nchunks, op.reserved_i = 32, nch, ppt
        op.in_, op.w, op.bias = x.data_ptr(), (wd if ppt else wfr).data_ptr(), bd.data_ptr()
        if which == "stats":
            op.stats = stats.data_ptr()
        else:
            op.aux2, op.out, op.flags = coeff.data_ptr(), out.data_ptr(), networks.GN_RELU_IN
        arr = (networks.XlOp * 1)(op)
        for _ in range(3):
            networks._check(L.xl_cnn_run(arr, 1, st))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            L.xl_cnn_run(arr, 1, st)
        e1.record()
        torch.cuda.synchronize()
        print("%s form, %s pass: %.3f ms" % ("packed-VALU" if ppt else "matrix-pipe", which, e0.elapsed_time(e1) / 20))
