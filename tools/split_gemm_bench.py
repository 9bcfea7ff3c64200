"""Timing of the batched Winograd GEMM of a 512->512 layer at 44 frames: fp32 MFMA kernel vs the split-bf16 kernel
(csrc/xl_gemm_split.hip), same V / U values.  python tools/split_gemm_bench.py [B=44] [C=512]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crossloc_amd import networks

B = int(sys.argv[1]) if len(sys.argv) > 1 else 44
C = int(sys.argv[2]) if len(sys.argv) > 2 else 512
nf, T, N = 64, B * 150, 512
L = networks._bind()
V = torch.randn(nf * T * C, device="cuda")
U = torch.randn(nf * N * C, device="cuda") * 0.05
Vs, Us = networks._Plan.split_bf16(V), networks._Plan.split_bf16(U)
M0 = torch.empty(nf * T * N, device="cuda"); M1 = torch.empty_like(M0)


def op(split):
    g = networks.XlOp()
    g.type = networks.XL_OP_CONV
    g.B, g.Hi, g.Wi, g.Cin, g.Ho, g.Wo, g.Cout = B, 10, 15, C, 10, 15, N
    g.ksize, g.stride, g.ld_in, g.ld_out, g.nchunks2 = 1, 1, C, N, nf
    g.in_, g.w, g.out = (Vs if split else V).data_ptr(), (Us if split else U).data_ptr(), (M1 if split else M0).data_ptr()
    g.flags = networks.CONV_SPLIT_BF16 if split else 0
    return (networks.XlOp * 1)(g)


st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
Vil = networks._Plan.split_bf16_interleaved(V.view(nf, T, C), C)
Uil = networks._Plan.split_bf16_interleaved(U.view(nf, N, C), C)
M2 = torch.empty_like(M0)


def op_il():
    g = op(1)[0]
    g.in_, g.w, g.out = Vil.data_ptr(), Uil.data_ptr(), M2.data_ptr()
    g.flags = networks.CONV_SPLIT_BF16 | networks.CONV_SPLIT_IL
    return (networks.XlOp * 1)(g)


for split in (0, 1, 2):
    arr = op_il() if split == 2 else op(split)
    for _ in range(3):
        networks._check(L.xl_cnn_run(arr, 1, st))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        networks._check(L.xl_cnn_run(arr, 1, st))
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("%s: %.3f ms  %.1f TFLOP/s (fp32-equivalent)" % (("fp32 MFMA ", "split-bf16", "split 256x256 interleaved")[split], ms, 2.0 * nf * T * N * C / ms / 1e9))
ref = M0.double()
print("max |split - fp32| / max|M| = %.2e" % ((M1.double() - ref).abs().max().item() / ref.abs().max().item()))
print("max |split256 - fp32| / max|M| = %.2e, bitwise equal to the 128x128 split form: %s" % (
    (M2.double() - ref).abs().max().item() / ref.abs().max().item(), torch.equal(M1, M2)))
