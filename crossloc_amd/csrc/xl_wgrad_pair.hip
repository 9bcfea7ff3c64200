// crossloc_hip: weight gradients of training plans as fp16 PAIRS, three matrix-pipe passes (round 5) - the loop of
// csrc/xl_wgrad_split.hip with the arithmetic of csrc/xl_gemm_pair.hip.
//
//   P_z,s[o][c] = sum_{t in split s} dY_z[t][o] * X_z[t][c]            ("pixels as the GEMM K dimension")
//
// Neither operand is packed ahead of time here, and both are converted in the kernel, but they are not symmetric:
//   * X - the layer's (normalised) input, or V = B^T x B for a Winograd layer - is a GroupNorm output: it takes the PLAN's
//     power-of-two scale (a bound, loose by construction) and the form that does not mind: {hi, lo' = (x - hi) 2^11};
//   * dY - a gradient (or dM = A dY A^T) - has no static bound: its scale comes from the DATA, the maximum magnitude the pass that
//     wrote it recorded (xl_amax_commit in csrc/xl_cnn_bwd.hip; max |dY| 2^e in [2^14, 2^15)), which makes it tight - so dY takes
//     the weight-like form {hi, lo = dY - hi} and its hs = hi 2^-11 is derived in registers.
// Products hs x lo', lo x hi, hi x hi on v_mfma_f32_32x32x16_f16, fp32 accumulation, exact un-scaling of the partial tile.
// A thread owns ONE channel (row of an operand tile) and 8 consecutive t, as in the bf16 kernel: the transpose costs nothing.
// Tile 256 (o) x 256 (c), 8 waves of 128 x 64, K-step = 16 t, two LDS stages of 2 x 16 KB, one barrier per step.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/crossloc_cnn.h"
#include "../../include/crossloc_dsac.h"   // status codes
#include "xl_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr int kUnit = 64;                               // bytes per row and K-step: 2 planes x 16 fp16
constexpr int kOperand = 256 * kUnit;                   // one operand of one stage: 16 KB
__device__ __forceinline__ int swz(int row) { return ((row >> 2) & 3) ^ (((row >> 1) & 1) << 1); }      // (csrc/xl_gemm_pair.hip)

struct WgPairArgs {
    const float *x, *dy; float *partial;
    int M, Cin, Cout, ldX, ldY, splits, mPerSplit, zCount, nbo, nbc;
    long long zX, zY;
    unsigned xBytes, dyBytes;
    // NORM: `x` is the RAW output of the producing convolution; its GroupNorm (+ReLU) is applied while the operand is loaded
    // (round 4, training plans whose GroupNorm applies are left to the consumers): coef = [B][Cin][2] {scale, shift}, HW pixels
    // per image (a multiple of 8: the 8 consecutive t of a thread belong to one image), normLo = 0 (ReLU) or -inf
    const float *coef; int HW; float normLo;
    const float *xScale;             // {s, 1 / s} of X (the plan's activation scale; its Winograd pair for V)
    const unsigned *dyAmax;          // max |dY| (or |dM|) of the launch's dY operand, as float bits
};

template <bool NORM>
__global__ __launch_bounds__(512)
void wgrad_pair_kernel(WgPairArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
    constexpr int kStage = 2 * kOperand;                              // dY side (rows = o) then x side (rows = c)
    constexpr unsigned OOB = 0x80000000u;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;                          // 2 x 4 waves of 128 (o) x 64 (c)

    // block -> (z, o-tile, c-tile, split); the splits of a tile are neighbours (their operand columns share cache lines)
    int b = blockIdx.x;
    const int s = b % a.splits; b /= a.splits;
    const int ct = b % a.nbc; b /= a.nbc;
    const int ot = b % a.nbo;
    const int z = b / a.nbo;
    const int o0 = ot * 256, c0 = ct * 256;
    const int mBeg = s * a.mPerSplit;
    int mEnd = mBeg + a.mPerSplit; if (mEnd > a.M) mEnd = a.M;
    const int nk = mEnd > mBeg ? (mEnd - mBeg + 15) >> 4 : 0;

    const __amdgpu_buffer_rsrc_t srdY = __builtin_amdgcn_make_buffer_rsrc((void *)(a.dy + z * a.zY), 0, (int)a.dyBytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t srdX = __builtin_amdgcn_make_buffer_rsrc((void *)(a.x + z * a.zX), 0, (int)a.xBytes, 0x00020000);
    const int row = tid & 255, th = tid >> 8;                         // my channel of both tiles; my half of the 16 t of a step
    const bool okO = o0 + row < a.Cout, okC = c0 + row < a.Cin;
    const float sX = a.xScale[0];
    float sY = 1.f, invAll = a.xScale[1];
    {
        const unsigned bits = a.dyAmax[0];
        int e = 0;
        if (bits >> 23) e = 14 - ((int)(bits >> 23) - 127);            // max |dY| 2^e in [2^14, 2^15)
        e = e > 120 ? 120 : (e < -120 ? -120 : e);
        sY = __builtin_bit_cast(float, (unsigned)(e + 127) << 23);
        invAll *= __builtin_bit_cast(float, (unsigned)(127 - e) << 23);
    }
    float ra[8], rb[8];
    float nSc = 1.f, nSh = 0.f;                                        // NORM: {scale, shift} of (image of the loaded step, my channel)
    int nImg = -1, nLive = 0;                                          // ... that image; how many of my 8 t are inside the split
    auto load_regs = [&](int kk) {                                    // K-step kk -> registers
        const int t0 = mBeg + 16 * kk + 8 * th;
        if constexpr (NORM) {
            nLive = mEnd - t0; nLive = nLive < 0 ? 0 : (nLive > 8 ? 8 : nLive);
            const int n = t0 / a.HW;
            if (okC && nLive > 0 && n != nImg) {
                nImg = n;
                const f32x2 c2 = *reinterpret_cast<const f32x2 *>(a.coef + ((long long)n * a.Cin + c0 + row) * 2);
                nSc = c2[0] * sX; nSh = c2[1] * sX;                   // (fmaf(x, scale s, shift s) = s fmaf(x, scale, shift) to the bit)
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int t = t0 + e;
            const bool live = t < mEnd;
            const unsigned va = (live && okO) ? (unsigned)(((long long)t * a.ldY + o0 + row) * 4) : OOB;
            const unsigned vb = (live && okC) ? (unsigned)(((long long)t * a.ldX + c0 + row) * 4) : OOB;
            ra[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srdY, (int)va, 0, 0));
            rb[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srdX, (int)vb, 0, 0));
        }
    };
    unsigned wOff[2];                                                 // my 16-byte slot of plane p in a row
#pragma unroll
    for (int p = 0; p < 2; ++p) wOff[p] = (unsigned)(row * kUnit + (((2 * p + th) ^ swz(row)) * 16));
    auto convert = [&](int stage) {                                   // registers -> stage (both operands)
        unsigned char *sb = dsm + stage * kStage;
        unsigned w[2][4];
#pragma unroll
        for (int h = 0; h < 4; ++h) {                                  // dY: {hi, lo} of dY sY (the tight, data-derived scale)
            const f32x2 v = f32x2{ ra[2 * h], ra[2 * h + 1] } * sY;
            const f16x2 vh = __builtin_convertvector(v, f16x2);
            w[0][h] = __builtin_bit_cast(unsigned, vh);
            w[1][h] = __builtin_bit_cast(unsigned, __builtin_convertvector(v - __builtin_convertvector(vh, f32x2), f16x2));
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) *reinterpret_cast<u32x4 *>(sb + wOff[p]) = u32x4{ w[p][0], w[p][1], w[p][2], w[p][3] };
        if constexpr (NORM) {                                          // one fmaf, one max: the arithmetic of every apply site, times sX
#pragma unroll
            for (int e = 0; e < 8; ++e) rb[e] = (okC && e < nLive) ? fmaxf(fmaf(rb[e], nSc, nSh), a.normLo) : 0.f;
        }
#pragma unroll
        for (int h = 0; h < 4; ++h) {                                  // X: {hi, lo' = (x - hi) 2^11} of x sX (the plan's scale)
            const f32x2 v = NORM ? f32x2{ rb[2 * h], rb[2 * h + 1] } : f32x2{ rb[2 * h], rb[2 * h + 1] } * sX;
            const f16x2 vh = __builtin_convertvector(v, f16x2);
            w[0][h] = __builtin_bit_cast(unsigned, vh);
            w[1][h] = __builtin_bit_cast(unsigned, __builtin_convertvector((v - __builtin_convertvector(vh, f32x2)) * 2048.f, f16x2));
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) *reinterpret_cast<u32x4 *>(sb + kOperand + wOff[p]) = u32x4{ w[p][0], w[p][1], w[p][2], w[p][3] };
    };

    const int fr = lane & 31, kh = lane >> 5;
    unsigned slot[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) slot[p] = (unsigned)(((2 * p + kh) ^ swz(fr)) * 16);
    const unsigned frA = (unsigned)((wm * 128 + fr) * kUnit), frB = (unsigned)(kOperand + (wn * 64 + fr) * kUnit);
    f16x8 fa[2][4], fas[4], fb[2][2];                                 // dY side {hi, lo} + hs; X side {hi, lo'}
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto ldA = [&](int stage, int p, int i) { return *reinterpret_cast<const f16x8 *>(dsm + stage * kStage + frA + i * 32 * kUnit + slot[p]); };
    auto ldB = [&](int stage, int p, int j) { return *reinterpret_cast<const f16x8 *>(dsm + stage * kStage + frB + j * 32 * kUnit + slot[p]); };
    auto mma = [&](const f16x8 (&x)[2], const f16x8 (&d)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x[j], d[i], acc[i][j], 0, 0, 0);
    };

    if (nk > 0) {
        load_regs(0);
        convert(0);
        if (nk > 1) load_regs(1);
        __syncthreads();
        for (int kk = 0; kk < nk; ++kk) {
            const int st = kk & 1;
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[0][i] = ldA(st, 0, i);
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[1][j] = ldB(st, 1, j);
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[1][i] = ldA(st, 1, i);
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[0][j] = ldB(st, 0, j);
#pragma unroll
            for (int i = 0; i < 4; ++i) fas[i] = fa[0][i] * (_Float16)0.00048828125f;     // hs = hi 2^-11
            mma(fb[1], fas);                                           // lo'(X) x hs(dY)
            __builtin_amdgcn_sched_barrier(0);
            // lo(dY) x hi(X) with the conversion of step kk + 1 threaded through it (the other stage: every wave finished
            // reading it before the barrier that ended step kk - 1)
            mma(fb[0], fa[1]);
            if (kk + 1 < nk) convert(st ^ 1);
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 10, 0);
                if (g == 3 || g == 7) __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (kk + 2 < nk) load_regs(kk + 2);                        // (the registers are free again)
            // my LDS writes are done, every wave has read this step's stage - but NOT vmcnt(0) (see csrc/xl_wgrad_split.hip)
            __builtin_amdgcn_s_waitcnt(0x0070 | 0xC00F);               // lgkmcnt(0)
            __builtin_amdgcn_sched_barrier(0);                         // (nothing of the next step may move above the barrier, nothing of
            __builtin_amdgcn_s_barrier();                              //  this one below it: the bare s_barrier carries no fence - ADVICE r4)
            __builtin_amdgcn_sched_barrier(0);
            mma(fb[0], fa[0]);                                         // hi x hi
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- store the partial tile: row o = lane & 31 (+ 32 i), 4 consecutive c per quad
    float *P = a.partial + ((long long)z * a.splits + s) * (long long)a.Cout * a.Cin;
    const int rhalf = kh * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int o = o0 + wm * 128 + i * 32 + fr;
        if (o >= a.Cout) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = c0 + wn * 64 + j * 32 + rhalf + 8 * q;
                if (c < a.Cin)
                    *reinterpret_cast<f32x4 *>(P + (long long)o * a.Cin + c) =
                        f32x4{ acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3] } * invAll;
            }
    }
}

}  // namespace

// XL_OP_WGRAD with ksize 1 and XL_CONV_SPLIT_BF16 | XL_CONV_PAIR_F16: scale = {s, 1 / s} of `in` (the plan's activation scale), out2 = the
// slot holding max |aux| as float bits (written by the pass that produced `aux`).  Otherwise the same operands and scratch layout as the fp32 kernel - in = x [M][Cin]
// (ld_in), aux = dY [M][Cout] (ld_aux), stats2 = partial [z][splits][Cout][Cin], groups = Z batched GEMMs (dense operands),
// nchunks2 = splits.  Cin % 4 == 0.  The caller runs wgrad_reduce_kernel afterwards.  XL_CONV_NORM_IN (1x1 layers, Z = 1): `in` is
// the raw output of the producing convolution, aux2 = its {scale, shift} pairs [B][Cin][2], XL_CONV_NORM_RELU; Ho*Wo % 8 == 0.
int xl_run_wgrad_pair(const xl_op &op, hipStream_t st)
{
    if (op.ksize != 1 || op.stride != 1 || op.nchunks2 < 1 || op.ld_in % 4 != 0 || op.ld_aux % 4 != 0 || op.Cin % 4 != 0 ||
        !op.in || !op.aux || !op.stats2 || !op.scale || !op.out2) return XL_ERR_ARG;
    const bool norm = (op.flags & XL_CONV_NORM_IN) != 0;
    if (norm && (!op.aux2 || op.groups > 1 || (op.Ho * op.Wo) % 8 != 0)) return XL_ERR_ARG;
    WgPairArgs a;
    a.xScale = (const float *)op.scale; a.dyAmax = (const unsigned *)op.out2;
    a.x = (const float *)op.in; a.dy = (const float *)op.aux; a.partial = (float *)op.stats2;
    a.M = op.B * op.Ho * op.Wo; a.Cin = op.Cin; a.Cout = op.Cout; a.ldX = op.ld_in; a.ldY = op.ld_aux;
    a.splits = op.nchunks2;
    a.mPerSplit = ((a.M + a.splits - 1) / a.splits + 15) / 16 * 16;
    a.nbo = (op.Cout + 255) / 256; a.nbc = (op.Cin + 255) / 256;
    a.zCount = op.groups > 1 ? op.groups : 1;
    if (a.zCount > 1 && (op.ld_in != op.Cin || op.ld_aux != op.Cout)) return XL_ERR_ARG;
    a.zX = (long long)a.M * op.Cin; a.zY = (long long)a.M * op.Cout;
    const long long xb = (((long long)a.M - 1) * op.ld_in + op.Cin) * 4, yb = (((long long)a.M - 1) * op.ld_aux + op.Cout) * 4;
    if (xb >= 0x7fffffffLL || yb >= 0x7fffffffLL) return XL_ERR_ARG;
    a.xBytes = (unsigned)xb; a.dyBytes = (unsigned)yb;
    a.coef = (const float *)op.aux2; a.HW = op.Ho * op.Wo;
    a.normLo = (op.flags & XL_CONV_NORM_RELU) ? 0.f : -__builtin_inff();
    const size_t lds = 4 * (size_t)kOperand;                          // two stages of two operands: 4 x 16 KB = 64 KB
    static XlLdsLimit configured[2];
    int cfgDev;
    if (configured[norm].needs(lds, &cfgDev)) {
        const void *fn = norm ? reinterpret_cast<const void *>(wgrad_pair_kernel<true>) : reinterpret_cast<const void *>(wgrad_pair_kernel<false>);
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return XL_ERR_HIP;
        configured[norm].done(lds, cfgDev);
    }
    if (norm) hipLaunchKernelGGL(wgrad_pair_kernel<true>, dim3(a.zCount * a.nbo * a.nbc * a.splits), dim3(512), lds, st, a);
    else hipLaunchKernelGGL(wgrad_pair_kernel<false>, dim3(a.zCount * a.nbo * a.nbc * a.splits), dim3(512), lds, st, a);
    return XL_OK;
}
