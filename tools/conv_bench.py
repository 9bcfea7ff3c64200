"""Micro-benchmark of one implicit-GEMM conv op (default: the dominant 3x3 512->512 @ 60x90 x16) with HIP
events; used standalone and under rocprofv3 --pmc to diagnose the kernel.  Not part of the product path."""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from crossloc_amd import networks  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cin", type=int, default=512)
    ap.add_argument("--cout", type=int, default=512)
    ap.add_argument("--k", type=int, default=3)
    ap.add_argument("--s", type=int, default=1)
    ap.add_argument("--B", type=int, default=16)
    ap.add_argument("--H", type=int, default=60)
    ap.add_argument("--W", type=int, default=90)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--z", type=int, default=1, help="batched GEMMs per launch (16 = the Winograd F(2x2,3x3) GEMMs)")
    ap.add_argument("--stats", action="store_true", help="fused GroupNorm statistics epilogue (32 groups), as the inference plans run 1x1 layers")
    ap.add_argument("--norm", action="store_true", help="normalise-on-load form of the 1x1 conv (XL_CONV_NORM_IN + ReLU)")
    ap.add_argument("--split", action="store_true", help="1x1 layer on the bf16 pipe (split_conv1x1_kernel)")
    a = ap.parse_args()
    L = networks._bind()
    torch.manual_seed(0)
    Ho = (a.H + 2 * (a.k // 2) - a.k) // a.s + 1
    Wo = (a.W + 2 * (a.k // 2) - a.k) // a.s + 1
    x = torch.randn(a.z, a.B, a.H, a.W, a.cin, device="cuda")
    w = torch.randn(a.z, a.cout, a.k, a.k, a.cin, device="cuda") * 0.02
    b = torch.randn(a.cout, device="cuda")
    out = torch.empty(a.z, a.B, Ho, Wo, a.cout, device="cuda")
    op = networks.XlOp()
    op.type = networks.XL_OP_CONV
    op.B, op.Hi, op.Wi, op.Cin, op.Ho, op.Wo, op.Cout = a.B, a.H, a.W, a.cin, Ho, Wo, a.cout
    op.ksize, op.stride, op.ld_in, op.ld_out = a.k, a.s, a.cin, a.cout
    op.in_, op.w, op.bias, op.out = x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr()
    if a.z > 1:
        op.nchunks2, op.bias = a.z, None
    if a.norm:
        coef = torch.stack([torch.rand(a.B, a.cin, device="cuda") + 0.5, torch.randn(a.B, a.cin, device="cuda")], 2).contiguous()
        op.flags, op.aux2 = networks.CONV_NORM_IN | networks.CONV_NORM_RELU, coef.data_ptr()
    if a.split:
        wsp = networks._Plan.split_bf16_interleaved(w.reshape(a.cout, a.cin), a.cin)
        op.w = wsp.data_ptr()
        op.flags |= networks.CONV_SPLIT_BF16 | networks.CONV_SPLIT_IL
    if a.stats:
        nch = (Ho * Wo + (255 if a.split else 127)) // (256 if a.split else 128) + 1
        st_buf = torch.zeros(a.B * nch * 32 * 2, dtype=torch.float64, device="cuda")
        op.stats, op.groups, op.nchunks = st_buf.data_ptr(), 32, nch
    arr = (networks.XlOp * 1)(op)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        networks._check(L.xl_cnn_run(arr, 1, st))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        L.xl_cnn_run(arr, 1, st)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    od = out.double()
    print("checksum sum=%.9e abs=%.9e" % (od.sum().item(), od.abs().sum().item()))
    flop = 2.0 * a.z * a.B * Ho * Wo * a.cout * a.k * a.k * a.cin
    print("conv %dx%d s%d %d->%d B%d %dx%d: %.4f ms  %.1f TFLOP/s (%.1f%% of 157.3)" % (
        a.k, a.k, a.s, a.cin, a.cout, a.B, a.H, a.W, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3 * 100))


if __name__ == "__main__":
    main()
