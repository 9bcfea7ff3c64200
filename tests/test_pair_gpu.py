"""GPU parity of the fp16 pair GEMMs (round 5, XL_CONV_PAIR_F16, csrc/xl_gemm_pair.hip + csrc/xl_pack.hip): the same
products as the split-bf16 kernels with three matrix-pipe passes instead of six.  Every GEMM is held against a float64 product
of the same fp32 operands AND against the fp32-MFMA kernel's own error on that product (the bar: fp32-class, <= 1.5x), over
operand magnitudes from 1e-8 to 1e4 and with activation scales far looser than necessary - the layout keeps the low term in
fp16's normal range, so a loose (safe) scale must not cost accuracy."""
import ctypes
import math
import os

import pytest

_PAIRS_OFF = os.environ.get("XL_GEMM_PAIR") in ("", "0") or os.environ.get("XL_GEMM_SPLIT_BF16") in ("0", "1")
pairs_only = pytest.mark.skipif(_PAIRS_OFF, reason="the fp16-pair forms are switched off (measurement switch set)")

torch = pytest.importorskip("torch")
import torch.nn as nn                      # noqa: E402
import torch.nn.functional as F            # noqa: E402

from crossloc_amd import networks          # noqa: E402

pytestmark = pytest.mark.gpu

PAIR = networks.CONV_SPLIT_BF16 | networks.CONV_SPLIT_IL | networks.CONV_PAIR_F16


def _run(ops):
    L = networks._bind()
    arr = (networks.XlOp * len(ops))(*ops)
    networks._check(L.xl_cnn_run(arr, len(ops), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()


def _scale_for(maxabs, slack=0):
    """{s, 1/s}: the largest power of two with s * maxabs <= 2^14, divided by 2^slack (a bound looser than the data)."""
    e = 14 - math.frexp(max(maxabs, 1e-30))[1] - slack
    s = math.ldexp(1.0, e)
    return torch.tensor([s, 1.0 / s], dtype=torch.float32, device="cuda")


def _pair_weight(w2d):
    """[rows][K] fp32 (device) -> pairs {hi, lo} + 2 floats, as the plan packs a 1x1 layer."""
    rows, K = w2d.shape
    dst = torch.zeros(2 * rows * K + 4, dtype=torch.int16, device="cuda")
    src = w2d.contiguous()
    networks._check(networks._bind().xl_cnn_pair_weight(src.data_ptr(), dst.data_ptr(), rows, K, 1, None))
    torch.cuda.synchronize()
    return dst


def _pairs(dst, Z, rows, K):
    body = dst[:2 * Z * rows * K].view(torch.float16).view(Z, rows, K // 16, 2, 16).float()
    tail = dst[2 * Z * rows * K:].view(torch.float32)
    hi, lo = (body[:, :, :, p].reshape(Z, rows, K) for p in range(2))
    return hi, lo, tail[Z:2 * Z]


@pytest.mark.parametrize("mag", [1.0, 3e-6, 2e3])
def test_pair_weight_is_the_scaled_pair(mag):
    g = torch.Generator().manual_seed(7)
    w = (torch.randn(256, 96, generator=g) * mag).cuda()
    w[3, 5] = 0.0
    hi, lo, inv = _pairs(_pair_weight(w), 1, 256, 96)
    s = 1.0 / inv.item()
    assert math.frexp(s)[0] == 0.5                                  # a power of two
    x = w.double() * s
    assert 2.0 ** 14 <= x.abs().max().item() < 2.0 ** 15
    assert torch.equal(hi[0], (w * s).half().float())
    assert torch.equal(lo[0], ((w * s) - hi[0]).half().float())
    # 22 significand bits: |x - hi - lo| <= 2^-22 |x| (or the fp16 subnormal spacing for the smallest elements)
    err = (x - hi[0].double() - lo[0].double()).abs()
    assert (err <= x.abs() * 2.0 ** -22 + 2.0 ** -25).all()


@pytest.mark.parametrize("m", [6, 4])
def test_winograd_pair_pack_vs_float64(m):
    g = torch.Generator().manual_seed(m)
    cout, cin = 256, 128
    w = (torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5).cuda()
    nf = (m + 2) ** 2
    dst = torch.zeros(2 * nf * cout * cin + 4 * nf, dtype=torch.int16, device="cuda")
    networks._check(networks._bind().xl_cnn_pack_wino_weight_pair(w.data_ptr(), dst.data_ptr(), cout, cin, m, 0, None))
    torch.cuda.synchronize()
    hi, lo, inv = _pairs(dst, nf, cout, cin)
    G = torch.tensor(networks._Plan._WINO_G[m], dtype=torch.float64, device="cuda")
    U = torch.einsum("xa,ocab,yb->xyoc", G, w.double(), G).reshape(nf, cout, cin)
    for z in range(nf):
        s = 1.0 / inv[z].item()
        assert math.frexp(s)[0] == 0.5
        x = U[z] * s
        assert 2.0 ** 14 <= x.abs().max().item() < 2.0 ** 15, (z, x.abs().max().item())
        # U rounded to fp32 once, then 22 bits of it
        assert ((hi[z].double() + lo[z].double()) - x).abs().max().item() <= 2.0 ** 15 * (2.0 ** -22 + 2.0 ** -24)


def _fp32_mfma_gemm(V, U, Z, T, cin, cout):
    """The same batched product on the fp32-MFMA kernel (the accuracy yardstick)."""
    Mb = torch.full((Z * T * cout,), float("nan"), device="cuda")
    gm = networks.XlOp()
    gm.type = networks.XL_OP_CONV
    gm.B, gm.Hi, gm.Wi, gm.Cin, gm.Ho, gm.Wo, gm.Cout = 1, T, 1, cin, T, 1, cout
    gm.ksize, gm.stride, gm.ld_in, gm.ld_out, gm.nchunks2 = 1, 1, cin, cout, Z
    gm.in_, gm.w, gm.out = V.data_ptr(), U.data_ptr(), Mb.data_ptr()
    _run([gm])
    return Mb.view(Z, T, cout)


@pytest.mark.parametrize("Z,T,cin,cout,vmag,umag,slack", [
    (64, 1200, 512, 512, 10.0, 0.02, 0),       # the dominant launch's shape (a few frames)
    (64, 1200, 512, 512, 10.0, 0.02, 9),       # ... with a bound 512 times looser than the data
    (3, 2900, 256, 512, 1e4, 1e-3, 0),         # |x| up to 1e4 and beyond
    (3, 2900, 256, 512, 1e4, 1e-3, 6),
    (5, 700, 128, 256, 1e-8, 50.0, 0),         # |x| down to 1e-8: the scale brings it into range
    (5, 700, 128, 256, 1e-8, 50.0, 12),
    (2, 257, 1536, 1024, 1.0, 1.0, 3),         # one ragged second row tile, four column tiles, 96 K-steps
    (2, 9000, 32, 256, 1.0, 1.0, 0)])          # two K-steps per tile: the operand stream runs a tile ahead
def test_pair_gemm_vs_float64_and_the_fp32_mfma_kernel(Z, T, cin, cout, vmag, umag, slack):
    """pair_gemm_kernel: V as activation pairs, U as weight pairs, both by LDS-DMA; M against float64."""
    g = torch.Generator().manual_seed(Z * 31 + T + cin)
    V = (torch.randn(Z, T, cin, generator=g) * vmag).cuda()
    V[:, :, ::7] *= 1e-3                                        # a wide spread inside every row
    V[0, 0, :16] = 0.0
    per_z = torch.logspace(-2, 2, Z)             # every frequency its own weight magnitude (as G g G^T has)
    U = (torch.randn(Z, cout, cin, generator=g) * umag * per_z[:, None, None]).cuda()
    scale = _scale_for(V.abs().max().item(), slack)
    L = networks._bind()
    Vp = torch.zeros(Z * T * cin * 2, dtype=torch.int16, device="cuda")
    networks._check(L.xl_cnn_pair_activation(V.data_ptr(), Vp.data_ptr(), Z * T, cin, scale.data_ptr(), None))
    Up = torch.zeros(2 * Z * cout * cin + 4 * Z, dtype=torch.int16, device="cuda")
    for z in range(Z):                                          # (one scale per matrix: pack them one by one, gather the tails)
        one = _pair_weight(U[z])
        Up[2 * z * cout * cin:2 * (z + 1) * cout * cin] = one[:2 * cout * cin]
        Up[2 * Z * cout * cin:].view(torch.float32)[Z + z] = one[2 * cout * cin:].view(torch.float32)[1]
    Mb = torch.full((Z * T * cout,), float("nan"), device="cuda")
    gm = networks.XlOp()
    gm.type = networks.XL_OP_CONV
    gm.B, gm.Hi, gm.Wi, gm.Cin, gm.Ho, gm.Wo, gm.Cout = 1, T, 1, cin, T, 1, cout
    gm.ksize, gm.stride, gm.ld_in, gm.ld_out, gm.nchunks2 = 1, 1, cin, cout, Z
    gm.flags = PAIR
    gm.in_, gm.w, gm.out, gm.scale = Vp.data_ptr(), Up.data_ptr(), Mb.data_ptr(), scale.data_ptr()
    _run([gm, gm])                                              # twice: the second launch overwrites
    got = Mb.view(Z, T, cout).double()
    assert torch.isfinite(got).all()
    ref = torch.matmul(V.double(), U.double().transpose(1, 2))
    m32 = _fp32_mfma_gemm(V, U, Z, T, cin, cout).double()
    for z in range(Z):
        sc = ref[z].abs().max().item()
        esp = (got[z] - ref[z]).abs().max().item() / sc
        e32 = (m32[z] - ref[z]).abs().max().item() / sc
        assert esp <= 1.5 * e32 + 1e-7, (z, esp, e32)


@pytest.mark.parametrize("cin,cout,B,H,W,norm,relu,stats,pad,form", [
    (512, 512, 3, 24, 37, True, True, True, 0, 256),      # tiles straddle images, last tile ragged
    (512, 512, 3, 24, 37, True, True, True, 0, -256),     # tiles that start at image boundaries (batch-invariant plans)
    (512, 512, 2, 60, 90, True, False, True, 0, 256),
    (256, 512, 3, 20, 31, False, False, True, 0, 256),
    (512, 512, 5, 16, 16, False, False, False, 32, 256),  # operands inside wider tensors
    (64, 1024, 1, 40, 52, False, False, False, 0, 256),
    (512, 512, 8, 60, 90, True, True, True, 0, 256),      # 338 tiles on 256 workgroups: second tiles (stores in flight)
    (32, 512, 7, 64, 80, True, True, True, 0, 256),       # two K-steps per tile
    (512, 256, 1, 60, 90, True, True, True, 0, 128),      # single-frame forms
    (512, 512, 1, 60, 90, True, True, True, 0, 192),
    (512, 512, 2, 60, 90, False, False, True, 0, 384)])
def test_pair_conv1x1_vs_float64_and_the_split_bf16_kernel(cin, cout, B, H, W, norm, relu, stats, pad, form):
    """pair_conv1x1_kernel: fp32 activations, (optionally) normalised and turned into pairs on their way into LDS; bias, the
    GroupNorm partial sums of the output, every tile form - against float64 and beside the six-pass bf16 kernel."""
    g = torch.Generator().manual_seed(cin + cout + B + H)
    x = torch.randn(B, cin, H, W, generator=g) * 3.0 + 1.0
    coef = torch.stack([torch.rand(B, cin, generator=g) + 0.5, torch.randn(B, cin, generator=g)], 2)
    conv = nn.Conv2d(cin, cout, 1)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * (2.0 / cin) ** 0.5)
        conv.bias.copy_(torch.randn(cout, generator=g))
        xn = x.double()
        if norm:
            xn = xn * coef[:, :, 0, None, None].double() + coef[:, :, 1, None, None].double()
            if relu:
                xn = xn.clamp(min=0)
        ref = F.conv2d(xn, conv.weight.double(), conv.bias.double())
    wide_in = torch.full((B, H, W, cin + pad), float("nan"))
    wide_in[..., pad:] = x.permute(0, 2, 3, 1)
    xd = wide_in.cuda()
    w2 = conv.weight.detach().reshape(cout, cin).cuda()
    wp, wb = _pair_weight(w2), networks._Plan.split_bf16_interleaved(w2, cin)
    bd, cd = conv.bias.detach().cuda(), coef.contiguous().cuda()
    scale = _scale_for(xn.abs().max().item(), 5)
    G, rows = cout // 16, 128 if abs(form) == 128 else 256
    nchunks = (H * W + rows - 1) // rows + 1

    def run(pair):
        out = torch.full((B, H, W, cout + pad), float("nan"), device="cuda")
        st = torch.zeros(B * nchunks * G * 2, dtype=torch.float64, device="cuda")
        op = networks.XlOp()
        op.type = networks.XL_OP_CONV
        op.B, op.Hi, op.Wi, op.Cin, op.Ho, op.Wo, op.Cout = B, H, W, cin, H, W, cout
        op.ksize, op.stride, op.ld_in, op.ld_out, op.reserved_i = 1, 1, cin + pad, cout + pad, form
        op.flags = PAIR if pair else (networks.CONV_SPLIT_BF16 | networks.CONV_SPLIT_IL)
        if norm:
            op.flags |= networks.CONV_NORM_IN | (networks.CONV_NORM_RELU if relu else 0)
            op.aux2 = cd.data_ptr()
        op.in_, op.w, op.bias, op.out = xd.data_ptr() + 4 * pad, (wp if pair else wb).data_ptr(), bd.data_ptr(), out.data_ptr()
        op.scale = scale.data_ptr()
        if stats:
            op.stats, op.groups, op.nchunks = st.data_ptr(), G, nchunks
        _run([op, op])
        o = out.cpu()
        if pad:
            assert torch.isnan(o[..., cout:]).all()
        return o[..., :cout].permute(0, 3, 1, 2).double(), st.cpu()
    got, st = run(True)
    six, _ = run(False)
    sc = ref.abs().max().item()
    esp, e6 = (got - ref).abs().max().item() / sc, (six - ref).abs().max().item() / sc
    assert esp < 2e-6 and esp <= 1.5 * e6 + 1e-7, (esp, e6)
    if stats:
        sums = st.view(B, nchunks, G, 2)
        if form < 0:
            assert (sums[:, (H * W + 255) // 256:] == 0).all()
        sums = sums.sum(1)
        grp = got.reshape(B, G, -1)
        assert torch.allclose(sums[:, :, 0], grp.sum(2), rtol=1e-5, atol=1e-3)
        assert torch.allclose(sums[:, :, 1], (grp * grp).sum(2), rtol=1e-5)


@pytest.mark.parametrize("cin,cout,B,H,W,pad,form", [(512, 512, 2, 60, 90, 0, 256), (512, 512, 3, 24, 37, 32, 256),
                                                     (256, 512, 9, 60, 90, 0, 256), (512, 256, 1, 60, 90, 0, 128)])
def test_pair_conv1x1_residual_on_load(cin, cout, B, H, W, pad, form):
    """XL_CONV_NORM_ADD on the pair kernel: v = max(max(fmaf(x, scale, shift), 0) + r, 0) formed while loading - BITWISE the plain
    pair kernel fed with v computed in fp32 in the same order (the scale is a power of two: it commutes with every rounding)."""
    g = torch.Generator().manual_seed(cin + cout + B + H)
    x = torch.randn(B, cin, H, W, generator=g) * 2.0 + 0.5
    r = torch.randn(B, cin, H, W, generator=g).clamp(min=0) * 1.5
    coef = torch.stack([torch.rand(B, cin, generator=g) + 0.5, torch.randn(B, cin, generator=g)], 2)
    w = torch.randn(cout, cin, generator=g) * (2.0 / cin) ** 0.5
    b = torch.randn(cout, generator=g)
    sc, sh = coef[:, :, 0, None, None], coef[:, :, 1, None, None]
    v64 = ((x.double() * sc.double() + sh.double()).clamp(min=0) + r.double()).clamp(min=0)
    ref = F.conv2d(v64, w.double()[:, :, None, None], b.double())
    v32 = ((x.double() * sc.double() + sh.double()).float().clamp(min=0) + r).clamp(min=0)

    def wide(t):
        o = torch.full((B, H, W, cin + pad), float("nan"))
        o[..., pad:] = t.permute(0, 2, 3, 1)
        return o.cuda()
    xd, rd, vd = wide(x), wide(r), wide(v32)
    wp, bd, cd = _pair_weight(w.cuda()), b.cuda(), coef.contiguous().cuda()
    scale = _scale_for(v64.abs().max().item(), 4)
    G, rows = cout // 16, 128 if form == 128 else 256
    nchunks = (H * W + rows - 1) // rows + 1

    def run(res_form):
        out = torch.full((B, H, W, cout), float("nan"), device="cuda")
        st = torch.zeros(B * nchunks * G * 2, dtype=torch.float64, device="cuda")
        op = networks.XlOp()
        op.type = networks.XL_OP_CONV
        op.B, op.Hi, op.Wi, op.Cin, op.Ho, op.Wo, op.Cout = B, H, W, cin, H, W, cout
        op.ksize, op.stride, op.ld_in, op.ld_out, op.reserved_i = 1, 1, cin + pad, cout, form
        op.flags = PAIR
        op.w, op.bias, op.out, op.scale = wp.data_ptr(), bd.data_ptr(), out.data_ptr(), scale.data_ptr()
        if res_form:
            op.flags |= networks.CONV_NORM_IN | networks.CONV_NORM_RELU | networks.CONV_NORM_ADD
            op.in_, op.aux, op.ld_aux, op.aux2 = xd.data_ptr() + 4 * pad, rd.data_ptr() + 4 * pad, cin + pad, cd.data_ptr()
        else:
            op.in_ = vd.data_ptr() + 4 * pad
        op.stats, op.groups, op.nchunks = st.data_ptr(), G, nchunks
        _run([op, op])
        return out.cpu(), st.cpu()
    got, st_res = run(True)
    plain, st_plain = run(False)
    assert torch.equal(got, plain), (got - plain).abs().max()
    assert torch.equal(st_res, st_plain)
    esp = (got.permute(0, 3, 1, 2).double() - ref).abs().max().item() / ref.abs().max().item()
    assert esp < 2e-6, esp


@pytest.mark.parametrize("cin,cout,B,H,W,norm,relu,pad", [
    (64, 128, 2, 60, 90, True, True, 0), (128, 256, 2, 60, 90, True, True, 0), (32, 64, 2, 40, 64, True, True, 0),
    (64, 128, 3, 37, 53, True, False, 32), (128, 256, 5, 33, 41, False, False, 0), (128, 256, 1, 120, 180, True, True, 0),
    (32, 64, 1, 97, 131, False, False, 0)])
def test_stride2_stem_conv_as_fp16_pairs(cin, cout, B, H, W, norm, relu, pad):
    """pair_conv3x3s2_kernel (csrc/xl_stem_pair.hip): the stride-2 3x3 stem layers with three fp16 passes - against a float64
    convolution, beside the six-pass bf16 kernel, with the GroupNorm partial sums of the epilogue."""
    g = torch.Generator().manual_seed(cin + cout + B + H)
    x = torch.randn(B, cin, H, W, generator=g) * 3.0 + 1.0
    coef = torch.stack([torch.rand(B, cin, generator=g) + 0.5, torch.randn(B, cin, generator=g)], 2)
    conv = nn.Conv2d(cin, cout, 3, 2, 1)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * (2.0 / (9 * cin)) ** 0.5)
        conv.bias.copy_(torch.randn(cout, generator=g))
        xn = x.double()
        if norm:
            xn = xn * coef[:, :, 0, None, None].double() + coef[:, :, 1, None, None].double()
            if relu:
                xn = xn.clamp(min=0)
        ref = F.conv2d(xn, conv.weight.double(), conv.bias.double(), stride=2, padding=1)
    Ho, Wo = ref.shape[2], ref.shape[3]
    wide_in = torch.full((B, H, W, cin + pad), float("nan"))
    wide_in[..., pad:] = x.permute(0, 2, 3, 1)
    xd = wide_in.cuda()
    wsrc = conv.weight.detach().cuda().contiguous()
    wp = torch.zeros(2 * wsrc.numel() + 4, dtype=torch.int16, device="cuda")
    networks._check(networks._bind().xl_cnn_pair_weight(wsrc.data_ptr(), wp.data_ptr(), cout, 9 * cin, 9, None))
    wb = networks._Plan.split_bf16_interleaved(networks._Plan._stem_rows(wsrc), 9 * cin)
    bd, cd = conv.bias.detach().cuda(), coef.contiguous().cuda()
    scale = _scale_for(xn.abs().max().item(), 5)
    bm, wm = {64: (128, 4), 128: (128, 2), 256: (256, 2)}[cout]
    HWo = Ho * Wo
    nchunks = (-(-HWo // bm) + 1) * wm

    def run(pair):
        out = torch.full((B, Ho, Wo, cout + pad), float("nan"), device="cuda")
        stats = torch.full((B, nchunks, 32, 2), float("nan"), dtype=torch.float64, device="cuda")
        op = networks.XlOp()
        op.type = networks.XL_OP_CONV
        op.B, op.Hi, op.Wi, op.Cin, op.Ho, op.Wo, op.Cout = B, H, W, cin, Ho, Wo, cout
        op.ksize, op.stride, op.ld_in, op.ld_out = 3, 2, cin + pad, cout + pad
        op.flags = PAIR if pair else (networks.CONV_SPLIT_BF16 | networks.CONV_SPLIT_IL)
        if norm:
            op.flags |= networks.CONV_NORM_IN | (networks.CONV_NORM_RELU if relu else 0)
            op.aux2 = cd.data_ptr()
        op.in_, op.w, op.bias, op.out = xd.data_ptr() + 4 * pad, (wp if pair else wb).data_ptr(), bd.data_ptr(), out.data_ptr()
        op.scale = scale.data_ptr()
        op.stats, op.groups, op.nchunks = stats.data_ptr(), 32, nchunks
        _run([op, op])
        o = out.cpu()
        if pad:
            assert torch.isnan(o[..., cout:]).all()
        return o[..., :cout].permute(0, 3, 1, 2).double(), stats.cpu()
    got, st = run(True)
    six, _ = run(False)
    sc = ref.abs().max().item()
    esp, e6 = (got - ref).abs().max().item() / sc, (six - ref).abs().max().item() / sc
    assert esp < 2e-6 and esp <= 1.5 * e6 + 1e-7, (esp, e6)
    grouped = got.reshape(B, 32, cout // 32, HWo)
    for n in range(B):
        valid = (((n + 1) * HWo - 1) // bm - (n * HWo) // bm + 1) * wm
        assert torch.isfinite(st[n, :valid]).all() and torch.isnan(st[n, valid:]).all(), (n, valid)
        s1, s2 = st[n, :valid, :, 0].sum(0), st[n, :valid, :, 1].sum(0)
        r1, r2 = grouped[n].sum((1, 2)), (grouped[n] ** 2).sum((1, 2))
        assert ((s1 - r1).abs() <= 2e-6 * grouped[n].abs().sum((1, 2))).all(), (n, (s1 - r1).abs().max())
        assert torch.allclose(s2, r2, rtol=2e-6, atol=0), (n, ((s2 - r2) / r2).abs().max())


@pytest.mark.parametrize("defer", [0, 1, 2])
def test_winograd_input_transform_writes_the_pairs_of_its_fp32_result(defer):
    """XL_OP_WINO_IN with XL_CONV_PAIR_F16: V as activation pairs = the split of the fp32 transform's V, to the bit."""
    B, H, W, C = 2, 13, 20, 128
    g = torch.Generator().manual_seed(defer)
    x = (torch.randn(B, H, W, C, generator=g) * 2.0).cuda()
    coef = torch.stack([torch.rand(B, C, generator=g) + 0.5, torch.randn(B, C, generator=g)], 2).contiguous().cuda()
    Th, Tw = -(-H // 6), -(-W // 6)
    T = B * Th * Tw
    scale = _scale_for(225 * 30.0, 2)
    outs = []
    for pair in (False, True):
        V = torch.zeros(64 * T * C, device="cuda")
        a = networks.XlOp()
        a.type, a.ksize = networks.XL_OP_WINO_IN, 6
        a.B, a.Hi, a.Wi, a.Cin, a.Ho, a.Wo, a.ld_in = B, H, W, C, Th, Tw, C
        a.in_, a.out = x.data_ptr(), V.data_ptr()
        if defer:
            a.aux2 = coef.data_ptr()
            a.flags = networks.GN_RELU_IN if defer == 2 else 0
        if pair:
            a.flags |= networks.CONV_PAIR_F16
            a.scale = scale.data_ptr()
        _run([a])
        outs.append(V)
    want = torch.zeros(64 * T * C * 2, dtype=torch.int16, device="cuda")
    networks._check(networks._bind().xl_cnn_pair_activation(outs[0].data_ptr(), want.data_ptr(), 64 * T, C, scale.data_ptr(), None))
    torch.cuda.synchronize()
    assert torch.equal(outs[1].view(torch.int16), want)


def test_pair_scales_follow_the_groupnorm_bound():
    """xl_cnn_pair_scales: s = the largest power of two with s * sum_i (sqrtN_i max|gamma_i| + max|beta_i|) <= 2^14."""
    g = torch.Generator().manual_seed(3)
    gam = [torch.randn(c, generator=g).cuda() for c in (32, 64, 512)]
    bet = [torch.randn(c, generator=g).cuda() for c in (32, 64, 512)]
    sq = [588.0, 415.7, 293.9]
    n = 3
    out = torch.zeros(8, device="cuda")
    networks._check(networks._bind().xl_cnn_pair_scales((ctypes.c_void_p * n)(*[t.data_ptr() for t in gam]),
                                                        (ctypes.c_void_p * n)(*[t.data_ptr() for t in bet]),
                                                        (ctypes.c_int * n)(32, 64, 512), (ctypes.c_float * n)(*sq), n, out.data_ptr(), None))
    torch.cuda.synchronize()
    bound = sum(r * a.abs().max().item() + b.abs().max().item() for r, a, b in zip(sq, gam, bet))
    s, inv, sv, svinv = out[:4].tolist()
    assert math.frexp(s)[0] == 0.5 and s * inv == 1.0 and sv * 256 == s and sv * svinv == 1.0
    assert s * bound <= 2.0 ** 14 < 2 * s * bound


@pytest.mark.skipif(os.environ.get("XL_GEMM_PAIR") == "0" or os.environ.get("XL_GEMM_SPLIT_BF16") in ("0", "1"), reason="pair path switched off")
def test_inference_plans_run_their_gemms_as_pairs_and_agree_with_the_six_pass_plan(monkeypatch):
    """The default inference plan takes the pair kernels for every split GEMM; its output agrees with the six-pass bf16 plan
    (XL_GEMM_PAIR=0) to fp32 rounding noise; re-scaling every GroupNorm's gamma and beta by 2^7 after the plan was built
    moves the activation scale with them (refresh through the weight version) and the two plans still agree."""
    from crossloc_amd.weights import seeded_state_dict
    mean = torch.tensor([-455.934, 417.50, 520.31])
    x = torch.rand(2, 3, 256, 384, generator=torch.Generator().manual_seed(5)).cuda()     # 32 x 48 maps: F(6x6,3x3), split 1x1 layers

    def build():
        net = networks.TransPoseNet(mean, False, False, 2, 2, 3, 1)
        net.load_state_dict(seeded_state_dict(net, 1234))
        return net.cuda().eval()
    outs = {}
    for tag in ("pair", "six"):
        if tag == "six":
            monkeypatch.setenv("XL_GEMM_PAIR", "0")
        net = build()
        with torch.no_grad():
            y0 = net(x).clone()
            plan = next(iter(net._plans.values())) if hasattr(net, "_plans") else None
            for mod in net.modules():
                if isinstance(mod, nn.GroupNorm):
                    mod.weight.mul_(2.0 ** 7)
                    mod.bias.mul_(2.0 ** 7)
            y1 = net(x).clone()
        outs[tag] = (y0, y1, plan)
    p = outs["pair"][2]
    if p is not None:
        npair = sum(1 for op in p.ops if op.type == networks.XL_OP_CONV and (op.flags & networks.CONV_PAIR_F16))
        nsix = sum(1 for op in p.ops if op.type == networks.XL_OP_CONV and (op.flags & networks.CONV_SPLIT_BF16) and op.ksize == 1
                   and not (op.flags & networks.CONV_PAIR_F16))
        assert npair > 0 and nsix == 0, (npair, nsix)
    for i in (0, 1):
        a, b = outs["pair"][i], outs["six"][i]
        assert torch.isfinite(a).all()
        d = (a[:, :3] - b[:, :3]).abs().max().item()
        assert d <= 2e-4 * max(1.0, b[:, :3].abs().max().item()), (i, d)


def _amax_slot(t):
    """What the GroupNorm-backward apply pass leaves for the pair GEMMs that read its result: max |t| as float bits."""
    return torch.tensor([t.abs().max().item()], dtype=torch.float32, device="cuda").view(torch.int32)


@pytest.mark.parametrize("cin,cout,M,Z,splits,dmag,xmag,slack", [
    (512, 512, 2 * 20 * 31, 1, 3, 1.0, 1.0, 5), (256, 512, 1500, 1, 1, 1e-7, 3.0, 9), (512, 256, 777, 4, 2, 2e3, 0.05, 3),
    (512, 512, 2400, 36, 1, 1e-4, 10.0, 6), (256, 256, 300, 2, 5, 1.0, 1.0, 0)])
def test_weight_gradient_products_as_fp16_pairs(cin, cout, M, Z, splits, dmag, xmag, slack):
    """XL_OP_WGRAD with XL_CONV_SPLIT_BF16 | XL_CONV_PAIR_F16 (csrc/xl_wgrad_pair.hip): P_z = dY_z^T X_z, the pixels / Winograd tiles as
    the K dimension, both operands converted in the kernel - dY at the scale of its recorded maximum (any magnitude: 1e-7 ... 2e3),
    X at the plan's scale (looser than its data by 2^slack) - against float64, beside the six-pass bf16 and the fp32-MFMA kernels."""
    g = torch.Generator().manual_seed(cin + cout + M + Z)
    x = torch.randn(Z, M, cin, generator=g) * xmag
    dy = torch.randn(Z, M, cout, generator=g) * dmag
    dy[:, ::5] *= 1e-3                                                # gradients are heavy-tailed: most rows far below the maximum
    ref = torch.einsum("zto,ztc->zoc", dy.double(), x.double())
    xd, dyd = x.cuda(), dy.cuda()
    xs, slot = _scale_for(x.abs().max().item(), slack), _amax_slot(dyd)

    def run(flags, sp):
        out = torch.full((Z, cout, cin), float("nan"), device="cuda")
        scratch = torch.full((Z * sp * cout * cin,), float("nan"), device="cuda")
        op = networks.XlOp()
        op.type, op.flags = networks.XL_OP_WGRAD, flags
        op.B, op.Hi, op.Wi, op.Cin, op.Ho, op.Wo, op.Cout = 1, M, 1, cin, M, 1, cout
        op.ksize, op.stride, op.ld_in, op.ld_aux, op.groups, op.nchunks2 = 1, 1, cin, cout, Z, sp
        op.in_, op.aux, op.out, op.stats2 = xd.data_ptr(), dyd.data_ptr(), out.data_ptr(), scratch.data_ptr()
        op.scale, op.out2 = xs.data_ptr(), slot.data_ptr()
        _run([op])
        return out.cpu().double()
    got = run(networks.CONV_SPLIT_BF16 | networks.CONV_PAIR_F16, splits)
    scale = ref.abs().max().item()
    esp = (got - ref).abs().max().item() / scale
    e6 = (run(networks.CONV_SPLIT_BF16, splits) - ref).abs().max().item() / scale
    e32 = (run(0, splits) - ref).abs().max().item() / scale
    assert esp < 2e-6 and esp <= 1.5 * max(e6, e32) + 1e-7, (esp, e6, e32)


@pytest.mark.parametrize("cin,cout,B,H,W,dmag", [(512, 512, 2, 20, 31, 1.0), (256, 512, 3, 16, 16, 3e-8), (512, 1024, 1, 24, 37, 4e3)])
def test_conv1x1_data_gradient_as_fp16_pairs_with_the_scale_from_the_recorded_maximum(cin, cout, B, H, W, dmag):
    """Training plans (round 5): dX = dY W of a 1x1 layer through pair_conv1x1_kernel with XL_CONV_PAIR_AMAX - the kernel derives its
    power-of-two scale from max |dY| (float bits in a slot, as the GroupNorm-backward apply pass records it), the transposed weights
    are pair-packed by xl_cnn_pair_weight(taps = 0); second producers accumulate.  Against float64, gradients from 3e-8 to 4e3."""
    L = networks._bind()
    g = torch.Generator().manual_seed(cin + cout + H)
    w = torch.randn(cout, cin, generator=g) * (2.0 / cin) ** 0.5
    dy = torch.randn(B, H, W, cout, generator=g) * dmag
    dy[:, ::3] *= 1e-4
    ref = torch.matmul(dy.double(), w.double())                      # [B,H,W,cin]
    ws = w.cuda().contiguous()
    wp = torch.zeros(2 * ws.numel() + 4, dtype=torch.int16, device="cuda")
    networks._check(L.xl_cnn_pair_weight(ws.data_ptr(), wp.data_ptr(), cin, cout, 0, None))
    dyd = dy.cuda()
    slot = _amax_slot(dyd)
    gx = torch.full((B, H, W, cin), float("nan"), device="cuda")
    op = networks.XlOp()
    op.type = networks.XL_OP_CONV
    op.flags = PAIR | networks.CONV_PAIR_AMAX
    op.B, op.Hi, op.Wi, op.Cin, op.Ho, op.Wo, op.Cout = B, H, W, cout, H, W, cin
    op.ksize, op.stride, op.ld_in, op.ld_out, op.reserved_i = 1, 1, cout, cin, 256
    op.in_, op.w, op.out, op.scale = dyd.data_ptr(), wp.data_ptr(), gx.data_ptr(), slot.data_ptr()
    _run([op])
    sc = ref.abs().max().item()
    assert ((gx.cpu().double() - ref).abs().max() / sc).item() < 1e-6
    op.flags |= networks.CONV_ACCUMULATE
    _run([op, op])
    assert ((gx.cpu().double() - 3 * ref).abs().max() / sc).item() < 3e-6


@pairs_only
def test_gradient_maxima_are_recorded_by_the_groupnorm_backward_pass():
    """A training step leaves max |dx| of every GroupNorm-backward apply pass in the plan's slots (zeroed by XL_OP_FILL0 at the head of
    the backward list): after a backward pass every slot a pair GEMM reads holds a positive finite float, and a second pass with
    gradients 1000 times larger moves every one of them by that factor (to fp32 rounding) - the slots are re-derived, not accumulated."""
    from crossloc_amd.weights import seeded_state_dict
    net = networks.TransPoseNet(torch.tensor([-455.934, 417.50, 520.31]), False, False, 1, 1, 3, 1)
    net.load_state_dict(seeded_state_dict(net, 77))
    net = net.cuda().train()
    x = torch.rand(2, 3, 256, 384, generator=torch.Generator().manual_seed(3)).cuda()
    vals = []
    for k in (1.0, 1000.0):
        net.zero_grad(set_to_none=True)
        (net(x) * k).sum().backward()
        torch.cuda.synchronize()
        plan = [p for p in net._plans.values() if p.train][0]
        used = sorted({op.scale for op in plan.bwd_array if op.type == networks.XL_OP_CONV and (op.flags & networks.CONV_PAIR_AMAX)} |
                      {op.out2 for op in plan.bwd_array if op.type == networks.XL_OP_WGRAD and (op.flags & networks.CONV_PAIR_F16)})
        assert len(used) > 5
        idx = [(a - plan.bwd_amax.data_ptr()) // 4 for a in used]
        v = plan.bwd_amax.view(torch.float32)[idx].cpu()
        assert torch.isfinite(v).all() and (v > 0).all()
        vals.append(v)
    assert torch.allclose(vals[1], vals[0] * 1000.0, rtol=1e-3)


def test_batched_repack_is_bitwise_the_per_matrix_packs():
    """xl_cnn_repack_pairs (what a training loop calls after every optimizer step): a list of F(6x6,3x3) layers (forward and data-
    gradient operands, different sizes), a list of F(4x4,3x3) ones and a list of plain matrices (1x1, transposed, 3x3 tap-major) in
    three launches per list - every byte of every destination, the scratch maxima and the inverse scales included, equals what
    xl_cnn_pack_wino_weight_pair / xl_cnn_pair_weight write for that matrix alone."""
    import numpy as np
    L = networks._bind()
    g = torch.Generator().manual_seed(11)
    dt = networks.PAIR_ITEM_DTYPE

    def check(m, specs):
        srcs, single, batch, items = [], [], [], []
        for cout, cin, kind, mag in specs:
            if m:
                w = (torch.randn(cout, cin, 3, 3, generator=g) * mag).cuda()
                n16 = (m + 2) ** 2 * cout * cin * 2 + 4 * (m + 2) ** 2
                rows, K = cout, cin
            elif kind == 9:
                w = (torch.randn(cout, cin, 3, 3, generator=g) * mag).cuda()
                n16, rows, K = cout * cin * 9 * 2 + 4, cout, cin * 9
            elif kind == 0:
                w = (torch.randn(cin, cout, generator=g) * mag).cuda()            # [K][rows]: the transpose is packed
                n16, rows, K = cout * cin * 2 + 4, cout, cin
            else:
                w = (torch.randn(cout, cin, generator=g) * mag).cuda()
                n16, rows, K = cout * cin * 2 + 4, cout, cin
            a = torch.full((n16,), 0x5555, dtype=torch.int16, device="cuda")
            b = torch.full((n16,), 0x2222, dtype=torch.int16, device="cuda")
            if m:
                networks._check(L.xl_cnn_pack_wino_weight_pair(w.data_ptr(), a.data_ptr(), cout, cin, m, kind, None))
            else:
                networks._check(L.xl_cnn_pair_weight(w.data_ptr(), a.data_ptr(), rows, K, kind, None))
            srcs.append(w); single.append(a); batch.append(b)
            items.append((w.data_ptr(), b.data_ptr(), rows, K, kind, 0))
        table = torch.from_numpy(np.array(items, dtype=dt).view(np.uint8).copy()).cuda()
        networks._check(L.xl_cnn_repack_pairs(table.data_ptr(), len(items), m, max(i[2] * i[3] for i in items), None))
        torch.cuda.synchronize()
        for a, b in zip(single, batch):
            assert torch.equal(a, b)
    check(6, [(512, 512, 0, 0.02), (512, 512, 1, 0.02), (256, 128, 0, 3.0), (128, 256, 1, 1e-6), (64, 64, 0, 0.0)])
    check(4, [(128, 128, 0, 0.05), (256, 128, 1, 0.5)])
    check(0, [(512, 512, 1, 0.03), (256, 512, 0, 0.03), (1024, 512, 1, 2.0), (128, 64, 9, 0.1), (64, 32, 9, 1e-5), (256, 256, 0, 0.0)])
    assert L.xl_cnn_repack_pairs(None, 0, 6, 1, None) == 0                      # an empty list is not an error
    assert L.xl_cnn_repack_pairs(None, 3, 6, 1, None) != 0 and L.xl_cnn_repack_pairs(1, 3, 5, 1, None) != 0


@pairs_only
def test_refresh_weights_batched_and_per_matrix_leave_identical_operands(monkeypatch):
    """A training plan's refresh_weights after an in-place parameter update: the batched re-pack (default) and the per-matrix one
    (XL_NO_BATCHED_REPACK=1) write identical pair operands, and both differ from the operands of the old weights."""
    from crossloc_amd.weights import seeded_state_dict
    net = networks.TransPoseNet(torch.tensor([-455.934, 417.50, 520.31]), False, False, 1, 1, 3, 1)
    net.load_state_dict(seeded_state_dict(net, 5))
    net = net.cuda().train()
    x = torch.rand(2, 3, 256, 384, generator=torch.Generator().manual_seed(4)).cuda()
    net(x).sum().backward()
    plan = [p for p in net._plans.values() if p.train][0]
    assert len(plan.packed_pair) > 10
    before = [e[0].clone() for e in plan.packed_pair.values()]
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(1.0 + 0.01 * torch.rand_like(p))
    plan.refresh_weights()
    torch.cuda.synchronize()
    batched = [e[0].clone() for e in plan.packed_pair.values()]
    assert any(not torch.equal(a, b) for a, b in zip(before, batched))
    for e in plan.packed_pair.values():
        e[0].fill_(0x1111)
    monkeypatch.setenv("XL_NO_BATCHED_REPACK", "1")
    plan.refresh_weights()
    torch.cuda.synchronize()
    for a, e in zip(batched, plan.packed_pair.values()):
        assert torch.equal(a, e[0])


@pairs_only
def test_training_plans_run_the_stride2_stem_layers_as_pairs():
    """conv3 / conv4 of a TRAINING plan under XL_TRAIN_PAIR_STEM=1 (inputs: materialised GroupNorm + ReLU outputs - the bound that
    makes the static scale safe holds for them as for the on-load form of inference plans) carry XL_CONV_PAIR_F16 (the default, 0:
    the six-pass kernels), and the forward of the two plans agrees to the pair GEMMs' accuracy."""
    from crossloc_amd.weights import seeded_state_dict
    outs = []
    for env in ("1", "0"):
        os.environ["XL_TRAIN_PAIR_STEM"] = env
        try:
            net = networks.TransPoseNet(torch.tensor([-455.934, 417.50, 520.31]), False, False, 1, 1, 3, 1)
            net.load_state_dict(seeded_state_dict(net, 5))
            net = net.cuda().train()
            x = torch.rand(2, 3, 256, 384, generator=torch.Generator().manual_seed(4)).cuda()
            y = net(x)
            plan = [p for p in net._plans.values() if p.train][0]
            stem = [op for op in plan.ops if op.type == networks.XL_OP_CONV and op.ksize == 3 and op.stride == 2]
            assert len(stem) >= 2
            assert all(bool(op.flags & networks.CONV_PAIR_F16) == (env == "1") for op in stem)
            outs.append(y.detach().cpu().double())
        finally:
            os.environ.pop("XL_TRAIN_PAIR_STEM", None)
    ref = outs[1]
    rng = (ref[:, :3] - torch.tensor([-455.934, 417.50, 520.31]).double()[None, :, None, None]).abs().max().item()
    assert (outs[0][:, :3] - ref[:, :3]).abs().max().item() <= 2e-4 * max(1.0, rng)
    assert torch.allclose(outs[0][:, 3], ref[:, 3], rtol=2e-3)
