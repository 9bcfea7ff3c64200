"""One stem layer (3x3 stride 2 on the split pipe) alone, at the size of the 47-frame plan: launch time by events on the
launch stream.  usage: stem_bench.py [cin] [frames];  XL_STEM_FORM selects a measurement form of the kernel."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from crossloc_amd import networks  # noqa: E402

cin = int(sys.argv[1]) if len(sys.argv) > 1 else 32
B = int(sys.argv[2]) if len(sys.argv) > 2 else 47
cout = 2 * cin
H, W = {32: (480, 720), 64: (240, 360), 128: (120, 180)}[cin]
Ho, Wo = H // 2, W // 2
g = torch.Generator().manual_seed(1)
x = (torch.randn(B, H, W, cin, generator=g) * 2 + 0.5).cuda()
coef = torch.stack([torch.rand(B, cin, generator=g) + 0.5, torch.randn(B, cin, generator=g)], 2).contiguous().cuda()
wt = (torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5).cuda()
wd = networks._Plan.split_bf16_interleaved(networks._Plan._stem_rows(wt), 9 * cin)
bd = torch.randn(cout, generator=g).cuda()
out = torch.empty(B, Ho, Wo, cout, device="cuda")
op = networks.XlOp()
op.type = networks.XL_OP_CONV
op.B, op.Hi, op.Wi, op.Cin, op.Ho, op.Wo, op.Cout = B, H, W, cin, Ho, Wo, cout
op.ksize, op.stride, op.ld_in, op.ld_out = 3, 2, cin, cout
op.flags = networks.CONV_SPLIT_BF16 | networks.CONV_SPLIT_IL | networks.CONV_NORM_IN | networks.CONV_NORM_RELU
op.aux2 = coef.data_ptr()
op.in_, op.w, op.bias, op.out = x.data_ptr(), wd.data_ptr(), bd.data_ptr(), out.data_ptr()
L = networks._bind()
arr = (networks.XlOp * 1)(op)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(3):
    networks._check(L.xl_cnn_run(arr, 1, st))
torch.cuda.synchronize()
n = 20
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    networks._check(L.xl_cnn_run(arr, 1, st))
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
flop = 2.0 * B * Ho * Wo * cout * 9 * cin * 6
print("stem %d->%d x%d form=%s: %.1f us  %.0f TFLOP/s bf16  checksum %.6e" % (cin, cout, B, os.environ.get("XL_STEM_FORM", "-"), ms * 1e3, flop / ms / 1e9,
                                                                  out.double().sum().item()))
