"""Timing of the batched Winograd GEMM of a 512->512 layer (64 products of [150 B x 512] x [512 x 512]): the fp16-pair kernels of
csrc/xl_gemm_pair.hip (both operands by DMA / V as fp32, pairs formed in the kernel) beside the six-pass bf16 kernel, same values.
python tools/pair_gemm_bench.py [B=95] [C=512]        XL_PAIR_CLK=1: per-phase shader ticks of the DMA form"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crossloc_amd import networks

B = int(sys.argv[1]) if len(sys.argv) > 1 else 95
C = int(sys.argv[2]) if len(sys.argv) > 2 else 512
nf, T, N = 64, B * 150, 512
L = networks._bind()
g = torch.Generator(device="cuda").manual_seed(1)
V = torch.randn(nf * T * C, device="cuda", generator=g) * 8.0
Wt = torch.randn(N, C, 3, 3, device="cuda", generator=g) * (2.0 / (9 * C)) ** 0.5
scale = torch.tensor([2.0 ** 5, 2.0 ** -5], device="cuda")
Vp = torch.zeros(nf * T * C * 2, dtype=torch.int16, device="cuda")
networks._check(L.xl_cnn_pair_activation(V.data_ptr(), Vp.data_ptr(), nf * T, C, scale.data_ptr(), None))
Up = torch.zeros(2 * nf * N * C + 4 * nf, dtype=torch.int16, device="cuda")
networks._check(L.xl_cnn_pack_wino_weight_pair(Wt.data_ptr(), Up.data_ptr(), N, C, 6, 0, None))
U6 = torch.zeros(3 * nf * N * C, dtype=torch.int16, device="cuda")
networks._check(L.xl_cnn_pack_wino_weight(Wt.data_ptr(), U6.data_ptr(), N, C, 6, 0, 2, None))
U32 = torch.zeros(nf * N * C, device="cuda")
networks._check(L.xl_cnn_pack_wino_weight(Wt.data_ptr(), U32.data_ptr(), N, C, 6, 0, 0, None))
# XL_PAIR_LO_MASK=k (experiment, round 6): clear the k low mantissa bits of every LOW term (lo' of V, lo of U) - does the matrix
# pipe draw less power, and so clock higher, on operands with fewer set bits?  (error vs float64 printed below)
_k = int(os.environ.get("XL_PAIR_LO_MASK", "0"))
if _k:
    m = torch.tensor(-(1 << _k), dtype=torch.int16, device="cuda")
    Vp.view(-1, 2, 8)[:, 1, :] &= m
    Up[: 2 * nf * N * C].view(-1, 2, 16)[:, 1, :] &= m
outs = {}


def op(kind):
    g = networks.XlOp()
    g.type = networks.XL_OP_CONV
    g.B, g.Hi, g.Wi, g.Cin, g.Ho, g.Wo, g.Cout = B, 10, 15, C, 10, 15, N
    g.ksize, g.stride, g.ld_in, g.ld_out, g.nchunks2, g.reserved_i = 1, 1, C, N, nf, 256
    outs[kind] = torch.empty(nf * T * N, device="cuda")
    g.out, g.scale = outs[kind].data_ptr(), scale.data_ptr()
    il = networks.CONV_SPLIT_BF16 | networks.CONV_SPLIT_IL
    if kind == "pair_dma":
        g.in_, g.w, g.flags = Vp.data_ptr(), Up.data_ptr(), il | networks.CONV_PAIR_F16
    elif kind == "pair_act":
        g.in_, g.w, g.flags = V.data_ptr(), Up.data_ptr(), il | networks.CONV_PAIR_F16 | networks.CONV_SPLIT_ACT
    elif kind == "six_act":
        g.in_, g.w, g.flags = V.data_ptr(), U6.data_ptr(), il | networks.CONV_SPLIT_ACT
    else:
        g.in_, g.w, g.flags = V.data_ptr(), U32.data_ptr(), 0
    return (networks.XlOp * 1)(g)


st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
kinds = ["pair_dma"] if os.environ.get("XL_PAIR_ONLY_DMA") else ["pair_dma", "pair_act", "six_act"] + (["f32"] if B <= 47 else [])
if os.environ.get("XL_PAIR_CLK"):               # (the phase clocks synchronise after every launch: their own run)
    for _ in range(3):
        networks._check(L.xl_cnn_run(op("pair_dma"), 1, st))
    torch.cuda.synchronize()
    sys.exit(0)
for kind in kinds:
    arr = op(kind)
    for _ in range(3):
        networks._check(L.xl_cnn_run(arr, 1, st))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        networks._check(L.xl_cnn_run(arr, 1, st))
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    passes = 3 if kind.startswith("pair") else 6 if kind == "six_act" else 1
    fl = 2.0 * nf * T * N * C
    print("%-9s %.3f ms  %.1f TFLOP/s fp32-equivalent, %.0f TFLOP/s on the pipe (%.3f of %s)" % (
        kind, ms, fl / ms / 1e9, passes * fl / ms / 1e9, passes * fl / ms / 1e9 / (2500.0 if passes > 1 else 157.3), "2500" if passes > 1 else "157.3"))
ref = torch.matmul((V.view(nf, T, C)[:4]).double(), U32.view(nf, N, C)[:4].double().transpose(1, 2))
for kind in kinds:
    got = outs[kind].view(nf, T, N)[:4].double()
    print("%-9s max err vs float64 / max|M| = %.3e" % (kind, ((got - ref).abs().amax((1, 2)) / ref.abs().amax((1, 2))).max().item()))
