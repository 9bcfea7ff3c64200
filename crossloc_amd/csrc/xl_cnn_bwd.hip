// xl_cnn_bwd.hip — backward kernels of the scene-coordinate CNN (what autograd + cuDNN dgrad/wgrad + native
// GroupNorm backward did for `loss.backward()`, /root/reference/train_single_task.py:298).
//
//   data gradient      the forward implicit-GEMM kernel in MODE 1 (xl_cnn.hip): same MFMA core, mirrored tap offsets,
//                      transposed weight operand, optional accumulate epilogue for gradients with two producers
//   wgrad_kernel       dW[o][tap][c] = sum_m dY[m][o] * X[m @ tap][c]: implicit GEMM with the PIXEL dimension as K,
//                      split-K over pixel ranges, fp32 MFMA 32x32x2 with operands read transposed from LDS
//                      ([pixel][channel] tiles, ds_read_b32 rows are conflict-free), fixed-order reduce -> OIHW
//   gnb_*              GroupNorm backward fused with the forward epilogue (ReLU / residual add / ReLU):
//                      pass 1 per-(image, chunk, channel) sums of dv, dv*xhat, xhat; pass 2 dx (+ d residual);
//                      then d gamma, d beta and the conv-bias gradient in closed form from the sums (no extra pass)
//   head_bwd_kernel    backward of fc3 + mean + exp(hardtanh)
//   conv1_wgrad_kernel weight/bias gradient of the 3-channel first conv
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/crossloc_cnn.h"
#include "../../include/crossloc_dsac.h"
#include "xl_common.h"

int xl_run_wgrad_split(const xl_op &op, hipStream_t st);   // xl_wgrad_split.hip
int xl_run_wgrad_pair(const xl_op &op, hipStream_t st);    // xl_wgrad_pair.hip

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t rsrc, unsigned voff)
{
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, 0, 0);
    return __builtin_bit_cast(f32x4, v);
}

// ---------------------------------------------------------------------------------------------- wgrad

// bijective XCD remap: block b runs on XCD b%8; give each XCD a contiguous run of work items
__device__ __forceinline__ int xcd_remap(int b, int nwg)
{
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = b & 7, local = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
}

struct WgradArgs {
    const float *x; const float *dy; float *partial;
    int B, Hi, Wi, Cin, Ho, Wo, Cout, ldX, ldY, ksize, stride;
    int M, splits, mPerSplit, nbo, nbc;
    unsigned xBytes, dyBytes;
    // batched launch (Winograd weight gradient: one [Cout x Cin] GEMM per frequency): zCount consecutive blocks of
    // x / dy (zX / zY elements apart) and of the partial buffer
    int zCount; long long zX, zY;
};

// tile BO (output channels) x BC (input channels) of ONE tap; K = output pixels of this split.
// waves WO x WC, wave tile (BO/WO) x (BC/WC) = TI x TJ MFMA tiles of 32x32.
// Both operand tiles are [pixel][channel] slices of dY / X, i.e. already K-major with the channel contiguous: they go
// global -> LDS by DMA (buffer_load ... lds, lane-linear = the natural row-major tile; out-of-range pixels and padding
// taps are zero-filled by the bounds check).  MFMA 32x32x2 takes ONE k per lane, so a lane cannot vectorise its
// fragment along k; it reads TI (TJ) CONSECUTIVE channels instead and feeds them to TI (TJ) different MFMAs: MFMA
// tile i of the wave holds the interleaved rows {TI*r + i} (columns {TJ*c + j}) — one ds_read_b64 per operand and
// k-pair instead of two ds_read_b32, and float2 stores in the epilogue.  The K-loop has the same shape as the
// forward kernel's: double-buffered, one barrier per K-step placed before its last chunk, the first fragments of
// the next step prefetched across the step boundary, DMA issue in the shadow of the last chunk's MFMAs.
template <int N> struct FragT { typedef float type; };
template <> struct FragT<2> { typedef float type __attribute__((ext_vector_type(2))); };
template <int N> __device__ __forceinline__ float frag_elem(const typename FragT<N>::type &v, int i);
template <> __device__ __forceinline__ float frag_elem<1>(const float &v, int) { return v; }
template <> __device__ __forceinline__ float frag_elem<2>(const FragT<2>::type &v, int i) { return v[i]; }

template <int BO, int BC, int WO, int WC>
__global__ __launch_bounds__(64 * WO * WC)
void wgrad_kernel(WgradArgs a)
{
    constexpr int NW = WO * WC, NT = 64 * NW;
    constexpr int TI = BO / WO / 32, TJ = BC / WC / 32;
    static_assert(TI <= 2 && TJ <= 2, "fragment width");
    constexpr int KB = 32;                                  // pixels per K-step
    constexpr int LA = (KB * BO / 4) / NT, LB = (KB * BC / 4) / NT;   // DMA instructions per wave and K-step
    constexpr int kWaitVm0 = 0x0F70;                        // s_waitcnt vmcnt(0)
    // ONE LDS object: with a second one hipcc waits vmcnt(0) before every ds_read that follows an LDS-DMA
    __shared__ __attribute__((aligned(16))) float smem[2 * KB * (BO + BC)];
    float (*sA)[KB * BO] = reinterpret_cast<float (*)[KB * BO]>(smem);                  // dY tile  [pixel][o]
    float (*sB)[KB * BC] = reinterpret_cast<float (*)[KB * BC]>(smem + 2 * KB * BO);    // X tile   [pixel][c]
    typedef __attribute__((address_space(3))) void lds_void;
    typedef typename FragT<TI>::type fragA_t;
    typedef typename FragT<TJ>::type fragB_t;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wo = wave / WC, wc = wave - wo * WC;
    const int PAD = (a.ksize == 3) ? 1 : 0;

    // Workgroup b runs on XCD b%8 (own L2).  Each XCD gets a contiguous run of the (split, tap, ob, cb) order, so
    // the ~64 workgroups resident on an XCD stream the SAME pixel range (one split) through its L2 — the dY / X rows
    // are fetched once per XCD and tap group instead of once per workgroup.
    const int taps = a.ksize * a.ksize;
    int t = xcd_remap(blockIdx.x, a.zCount * a.splits * taps * a.nbo * a.nbc);
    const int cb = t % a.nbc; t /= a.nbc;
    const int ob = t % a.nbo; t /= a.nbo;
    const int tap = t % taps; t /= taps;
    const int split = t % a.splits;
    const int z = t / a.splits;
    a.x += z * a.zX; a.dy += z * a.zY;
    a.partial += (long long)z * a.splits * taps * a.Cout * a.Cin;
    const int ky = tap / a.ksize, kx = tap - ky * a.ksize;
    const int o0 = ob * BO, c0 = cb * BC;
    const int mBeg = split * a.mPerSplit;
    int mEnd = mBeg + a.mPerSplit; if (mEnd > a.M) mEnd = a.M;

    const __amdgpu_buffer_rsrc_t srdX = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, (int)a.xBytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t srdY = __builtin_amdgcn_make_buffer_rsrc((void *)a.dy, 0, (int)a.dyBytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    const int HoWo = a.Ho * a.Wo;

    // DMA slot q of this wave covers float4s (q*NW + wave)*64 .. +63 of the [KB][B?/4] tile
    int rowA[LA], colA[LA], rowB[LB], colB[LB];
#pragma unroll
    for (int q = 0; q < LA; ++q) { const int f = (q * NW + wave) * 64 + lane; rowA[q] = f / (BO / 4); colA[q] = (f - rowA[q] * (BO / 4)) * 4 + o0; }
#pragma unroll
    for (int q = 0; q < LB; ++q) { const int f = (q * NW + wave) * 64 + lane; rowB[q] = f / (BC / 4); colB[q] = (f - rowB[q] * (BC / 4)) * 4 + c0; }
    // Pixel coordinates of the X rows this lane fetches, advanced by KB pixels per K-step WITHOUT divisions
    // (straight-line selects only: the address arithmetic must stay in the MFMA basic block to overlap with it).
    // Requires Ho*Wo >= KB (one image wrap per step at most); checked by the launcher.
    int pn[LB], py[LB], px[LB];
    const int dOy = KB / a.Wo, dOx = KB - dOy * a.Wo;
#pragma unroll
    for (int q = 0; q < LB; ++q) {
        const int m = mBeg + rowB[q];
        pn[q] = m / HoWo;
        const int rem = m - pn[q] * HoWo;
        py[q] = rem / a.Wo; px[q] = rem - py[q] * a.Wo;
    }
    int mCur = mBeg;
    // offsets of the K-step starting at mCur, then advance to the next one
    auto offsets = [&](unsigned (&va)[LA], unsigned (&vb)[LB]) {
#pragma unroll
        for (int q = 0; q < LA; ++q) {
            const int m = mCur + rowA[q];
            va[q] = m < mEnd ? (unsigned)(m * a.ldY + colA[q]) * 4u : OOB;
        }
#pragma unroll
        for (int q = 0; q < LB; ++q) {
            const int m = mCur + rowB[q];
            const int iy = py[q] * a.stride - PAD + ky, ix = px[q] * a.stride - PAD + kx;
            const bool ok = m < mEnd && (unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi;
            // mask arithmetic, not a select: hipcc turns the select into an exec-masked branch around the multiplies
            const unsigned msk = 0u - (unsigned)ok;
            vb[q] = (((unsigned)(((pn[q] * a.Hi + iy) * a.Wi + ix) * a.ldX + colB[q]) * 4u) & msk) | (OOB & ~msk);
            px[q] += dOx; py[q] += dOy;
            const bool cx = px[q] >= a.Wo;
            px[q] -= cx ? a.Wo : 0; py[q] += cx ? 1 : 0;
            const bool cy = py[q] >= a.Ho;
            py[q] -= cy ? a.Ho : 0; pn[q] += cy ? 1 : 0;
        }
        mCur += KB;
    };
    auto issue_a = [&](const unsigned (&va)[LA], int buf) {
#pragma unroll
        for (int q = 0; q < LA; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srdY, (lds_void *)&sA[buf][(q * NW + wave) * 256], 16, (int)va[q], 0, 0, 0);
    };
    auto issue_b = [&](const unsigned (&vb)[LB], int buf) {
#pragma unroll
        for (int q = 0; q < LB; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srdX, (lds_void *)&sB[buf][(q * NW + wave) * 256], 16, (int)vb[q], 0, 0, 0);
    };

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    int nk = (mEnd - mBeg + KB - 1) / KB;
    if (nk < 0) nk = 0;
    {
        unsigned va[LA], vb[LB];
        offsets(va, vb);
        issue_a(va, 0); issue_b(vb, 0);
        offsets(va, vb);
        issue_a(va, 1); issue_b(vb, 1);
    }
    __builtin_amdgcn_s_waitcnt(kWaitVm0);                   // the fence of __syncthreads() does not wait for LDS-DMA
    __syncthreads();

    const int fr = lane & 31, fk = lane >> 5;
    const float *Afrag = &sA[0][fk * BO + wo * (BO / WO) + TI * fr];
    const float *Bfrag = &sB[0][fk * BC + wc * (BC / WC) + TJ * fr];
    // fragments of a chunk of 4 k-pairs, ping-ponging between two register sets
    fragA_t fa[2][4];
    fragB_t fb[2][4];
    auto read_frags = [&](int set, int buf, int c) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            fa[set][u] = *reinterpret_cast<const fragA_t *>(Afrag + buf * (KB * BO) + 2 * (4 * c + u) * BO);
            fb[set][u] = *reinterpret_cast<const fragB_t *>(Bfrag + buf * (KB * BC) + 2 * (4 * c + u) * BC);
        }
    };
    auto multiply_u = [&](int set, int u) {
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(frag_elem<TI>(fa[set][u], i), frag_elem<TJ>(fb[set][u], j),
                                                                 acc[i][j], 0, 0, 0);
    };
    auto multiply = [&](int set) {
#pragma unroll
        for (int u = 0; u < 4; ++u) multiply_u(set, u);
    };
    constexpr int NM = 4 * TI * TJ, ND = 8;
    read_frags(0, 0, 0);
    for (int kk = 0; kk < nk; ++kk) {
        const int buf = kk & 1;
        unsigned vaN[LA], vbN[LB];                          // offsets of step kk+2, computed under this step's MFMAs
        offsets(vaN, vbN);
        read_frags(1, buf, 1);
        multiply(0);
        read_frags(0, buf, 2);
        multiply(1);
        read_frags(1, buf, 3);
        multiply(0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, ND, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, ND, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, ND, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, NM - 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(kWaitVm0);               // tile kk+1 has landed
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        multiply_u(1, 0);
        __builtin_amdgcn_sched_barrier(0x6);
        issue_a(vaN, buf);
        __builtin_amdgcn_sched_barrier(0x6);
        multiply_u(1, 1);
        __builtin_amdgcn_sched_barrier(0x6);
        issue_b(vbN, buf);
        read_frags(0, buf ^ 1, 0);
        __builtin_amdgcn_sched_barrier(0x6);
        multiply_u(1, 2);
        multiply_u(1, 3);
    }
    __builtin_amdgcn_s_waitcnt(kWaitVm0);                   // trailing DMAs must not outlive the workgroup's LDS

    // partial[split][tap][o][c]; MFMA tile (i, j): rows TI*row + i, columns TJ*col + j
    const int col = lane & 31, rhalf = (lane >> 5) * 4;
    float *P = a.partial + ((long long)split * a.ksize * a.ksize + tap) * a.Cout * a.Cin;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = o0 + wo * (BO / WO) + TI * ((r & 3) + 8 * (r >> 2) + rhalf) + i;
            const int c = c0 + wc * (BC / WC) + TJ * col;
            float *dst = P + (long long)o * a.Cin + c;
            if constexpr (TJ == 2) {
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                *reinterpret_cast<f32x2 *>(dst) = f32x2{ acc[i][0][r], acc[i][1][r] };
            } else {
                *dst = acc[i][0][r];
            }
        }
}

// dW[o][c][ky][kx] = sum over splits (fixed order) of partial[s][tap][o][c]
__global__ void wgrad_reduce_kernel(const float *__restrict__ partial, float *__restrict__ dw, int splits, int taps,
                                    int Cout, int Cin)
{
    const long long total = (long long)taps * Cout * Cin;
    partial += (long long)blockIdx.y * splits * total;              // batched launch: one block of partials per GEMM
    dw += (long long)blockIdx.y * total;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < splits; ++k) s += partial[(long long)k * total + i];
        const int c = (int)(i % Cin);
        long long t = i / Cin;
        const int o = (int)(t % Cout);
        const int tap = (int)(t / Cout);
        dw[((long long)o * Cin + c) * taps + tap] = s;
    }
}

template <int BO, int BC, int WO, int WC>
int launch_wgrad(const xl_op &op, hipStream_t st)
{
    WgradArgs a;
    a.x = (const float *)op.in; a.dy = (const float *)op.aux; a.partial = (float *)op.stats2;
    a.B = op.B; a.Hi = op.Hi; a.Wi = op.Wi; a.Cin = op.Cin; a.Ho = op.Ho; a.Wo = op.Wo; a.Cout = op.Cout;
    a.ldX = op.ld_in; a.ldY = op.ld_aux; a.ksize = op.ksize; a.stride = op.stride;
    a.M = op.B * op.Ho * op.Wo; a.splits = op.nchunks2;
    a.mPerSplit = ((a.M + a.splits - 1) / a.splits + 31) / 32 * 32;
    a.nbo = op.Cout / BO; a.nbc = op.Cin / BC;
    a.zCount = op.groups > 1 ? op.groups : 1;                     // WGRAD: groups = GEMMs per launch
    if (a.zCount > 1 && (op.ksize != 1 || op.ld_in != op.Cin || op.ld_aux != op.Cout)) return XL_ERR_ARG;
    a.zX = (long long)a.M * op.Cin; a.zY = (long long)a.M * op.Cout;
    const long long xb = (((long long)op.B * op.Hi * op.Wi - 1) * op.ld_in + op.Cin) * 4;
    const long long yb = (((long long)a.M - 1) * op.ld_aux + op.Cout) * 4;
    if (xb >= 0x7fffffffLL || yb >= 0x7fffffffLL) return XL_ERR_ARG;
    if (op.Ho * op.Wo < 32) return XL_ERR_UNSUPPORTED;          // the kernel advances pixel coordinates by one K-step (32)
    a.xBytes = (unsigned)xb; a.dyBytes = (unsigned)yb;
    const int taps = op.ksize * op.ksize;
    hipLaunchKernelGGL((wgrad_kernel<BO, BC, WO, WC>), dim3(a.zCount * taps * a.nbo * a.nbc * a.splits), dim3(64 * WO * WC), 0, st, a);
    const long long total = (long long)taps * op.Cout * op.Cin;
    long long blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)blocks, a.zCount), dim3(256), 0, st, (const float *)op.stats2,
                       (float *)op.out, a.splits, taps, op.Cout, op.Cin);
    return XL_OK;
}

// ---------------------------------------------------------------------------------------------- Winograd weight gradient

// Weight gradient of a stride-1 3x3 layer through F(4x4,3x3): with V = B^T x B (wino4_in_kernel) and dM = A dY A^T per
// 4x4 output tile, dU[xi][co][ci] = sum_tiles dM[xi][t][co] V[xi][t][ci] is 36 GEMMs with the TILES as the K dimension
// (the batched wgrad_kernel above, 4x fewer multiplies than the 9-tap form), and dg = G^T dU G.
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <typename V>
__device__ __forceinline__ void wino4_a(const V (&d)[4], V (&o)[6])      // A = (A^T)^T, 6x4
{
    o[0] = d[0];
    o[1] = d[0] + d[1] + d[2] + d[3];
    o[2] = d[0] - d[1] + d[2] - d[3];
    o[3] = d[0] + 2.f * d[1] + 4.f * d[2] + 8.f * d[3];
    o[4] = d[0] - 2.f * d[1] + 4.f * d[2] - 8.f * d[3];
    o[5] = d[3];
}

// dY [B,H,W,C] (pixel stride ld) -> dM [36][B*Th*Tw][C]; one tile x 2 channels per thread; pixels past H / W are zero
__global__ __launch_bounds__(256)
void wino4_dy_kernel(const float *__restrict__ dy, float *__restrict__ dM, int B, int H, int W, int C, int ld, int Th, int Tw)
{
    const int C2 = C >> 1;
    const long long T = (long long)B * Th * Tw;
    const long long items = T * C2;
    const long long zs = T * C;
    for (long long it = (long long)blockIdx.x * 256 + threadIdx.x; it < items; it += (long long)gridDim.x * 256) {
        const int c2 = (int)(it % C2);
        const long long t = it / C2;
        const int tx = (int)(t % Tw);
        const int ty = (int)((t / Tw) % Th);
        const int n = (int)(t / ((long long)Tw * Th));
        f32x2 w[6][4];                               // w[i][q] = (A dY)[i][q]
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int x = 4 * tx + q;
            f32x2 col[4], o[6];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int y = 4 * ty + p;
                if (y < H && x < W) col[p] = *reinterpret_cast<const f32x2 *>(dy + (((long long)n * H + y) * W + x) * ld + 2 * c2);
                else col[p] = f32x2{ 0.f, 0.f };
            }
            wino4_a(col, o);
#pragma unroll
            for (int i = 0; i < 6; ++i) w[i][q] = o[i];
        }
        float *op = dM + t * C + 2 * c2;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            f32x2 o[6];
            wino4_a(w[i], o);
#pragma unroll
            for (int j = 0; j < 6; ++j) *reinterpret_cast<f32x2 *>(op + (6 * i + j) * zs) = o[j];
        }
    }
}

// dg[o][c][a][b] = sum_ij G[i][a] dU[6i+j][o][c] G[j][b]  (G of F(4x4,3x3)); one (o, c) per thread, OIHW output
__global__ __launch_bounds__(256)
void wino4_wfinal_kernel(const float *__restrict__ dU, float *__restrict__ dw, int Cout, int Cin)
{
    const long long total = (long long)Cout * Cin;
    const float G[6][3] = { { 0.25f, 0.f, 0.f }, { -1.f / 6, -1.f / 6, -1.f / 6 }, { -1.f / 6, 1.f / 6, -1.f / 6 },
                            { 1.f / 24, 1.f / 12, 1.f / 6 }, { 1.f / 24, -1.f / 12, 1.f / 6 }, { 0.f, 0.f, 1.f } };
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        float u[6][6];
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b = 0; b < 6; ++b) u[a][b] = dU[(long long)(6 * a + b) * total + i];
        float t[3][6];                               // t = G^T u
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 6; ++b) {
                float v = 0.f;
#pragma unroll
                for (int k = 0; k < 6; ++k) v = fmaf(G[k][a], u[k][b], v);
                t[a][b] = v;
            }
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                float v = 0.f;
#pragma unroll
                for (int k = 0; k < 6; ++k) v = fmaf(t[a][k], G[k][b], v);
                dw[i * 9 + a * 3 + b] = v;
            }
    }
}

// The same through F(6x6,3x3): dM = A dY A^T per 6x6 output tile (A = (A^T)^T, 8x6), 64 GEMMs, dg = G^T dU G with the
// 8x3 G.  Measured against float64 on a 64-channel layer: 9e-6 of max|ref| (F(4x4,3x3): 6e-6).
template <typename V>
__device__ __forceinline__ void wino6_a(const V (&d)[6], V (&o)[8])
{
    const V e = d[0] + d[2] + d[4], f = d[1] + d[3] + d[5];
    const V g = d[0] + 4.f * d[2] + 16.f * d[4], h = 2.f * d[1] + 8.f * d[3] + 32.f * d[5];
    const V k = 32.f * d[0] + 8.f * d[2] + 2.f * d[4], l = 16.f * d[1] + 4.f * d[3] + d[5];
    o[0] = d[0];
    o[1] = e + f; o[2] = e - f;
    o[3] = g + h; o[4] = g - h;
    o[5] = k + l; o[6] = k - l;
    o[7] = d[5];
}

// Round 5: the largest magnitude a gradient pass writes, for the fp16-pair GEMMs that consume its result (csrc/xl_gemm_pair.hip,
// csrc/xl_wgrad_pair.hip: a gradient has no static bound, so its power-of-two scale comes from the data): every thread keeps a
// running maximum, a wave combines its 64 with a DPP tree, at most one atomicMax per wave on the float's bits (xl_wave_max_commit,
// csrc/xl_common.h).  The slot is zeroed by an XL_OP_FILL0 at the head of the backward op list.
__device__ __forceinline__ void xl_amax_commit(float m, unsigned *slot) { xl_wave_max_commit(m, slot); }

// dY [B,H,W,C] (pixel stride ld) -> dM [64][B*Th*Tw][C]; one tile x 2 channels per thread; pixels past H / W are zero
__global__ __launch_bounds__(256)
void wino6_dy_kernel(const float *__restrict__ dy, float *__restrict__ dM, int B, int H, int W, int C, int ld, int Th, int Tw,
                     unsigned *__restrict__ amax)
{
    float mx = 0.f;
    const int C2 = C >> 1;
    const long long T = (long long)B * Th * Tw;
    const long long items = T * C2;
    const long long zs = T * C;
    for (long long it = (long long)blockIdx.x * 256 + threadIdx.x; it < items; it += (long long)gridDim.x * 256) {
        const int c2 = (int)(it % C2);
        const long long t = it / C2;
        const int tx = (int)(t % Tw);
        const int ty = (int)((t / Tw) % Th);
        const int n = (int)(t / ((long long)Tw * Th));
        f32x2 w[8][6];                               // w[i][q] = (A dY)[i][q]
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const int x = 6 * tx + q;
            f32x2 col[6], o[8];
#pragma unroll
            for (int p = 0; p < 6; ++p) {
                const int y = 6 * ty + p;
                if (y < H && x < W) col[p] = *reinterpret_cast<const f32x2 *>(dy + (((long long)n * H + y) * W + x) * ld + 2 * c2);
                else col[p] = f32x2{ 0.f, 0.f };
            }
            wino6_a(col, o);
#pragma unroll
            for (int i = 0; i < 8; ++i) w[i][q] = o[i];
        }
        float *op = dM + t * C + 2 * c2;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            f32x2 o[8];
            wino6_a(w[i], o);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                *reinterpret_cast<f32x2 *>(op + (8 * i + j) * zs) = o[j];
                mx = fmaxf(mx, fmaxf(fabsf(o[j][0]), fabsf(o[j][1])));
            }
        }
    }
    xl_amax_commit(mx, amax);
}

// dg[o][c][a][b] = sum_ij G[i][a] dU[8i+j][o][c] G[j][b]  (G of F(6x6,3x3)); one (o, c) per thread, OIHW output
__global__ __launch_bounds__(256)
void wino6_wfinal_kernel(const float *__restrict__ dU, float *__restrict__ dw, int Cout, int Cin)
{
    const long long total = (long long)Cout * Cin;
    const float G[8][3] = { { 1.f, 0.f, 0.f }, { -2.f / 9, -2.f / 9, -2.f / 9 }, { -2.f / 9, 2.f / 9, -2.f / 9 },
                            { 1.f / 90, 1.f / 45, 2.f / 45 }, { 1.f / 90, -1.f / 45, 2.f / 45 },
                            { 1.f / 45, 1.f / 90, 1.f / 180 }, { 1.f / 45, -1.f / 90, 1.f / 180 }, { 0.f, 0.f, 1.f } };
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        float t[3][8];                               // t = G^T u, built row by row of u
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 8; ++b) t[a][b] = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const float u = dU[(long long)(8 * k + b) * total + i];
#pragma unroll
                for (int a = 0; a < 3; ++a) t[a][b] = fmaf(G[k][a], u, t[a][b]);
            }
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                float v = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) v = fmaf(t[a][k], G[k][b], v);
                dw[i * 9 + a * 3 + b] = v;
            }
    }
}

// ---------------------------------------------------------------------------------------------- GroupNorm backward

struct GnbArgs {
    const float *x, *dout, *outAct, *gamma, *beta;
    const float *fco;        // forward coefficients from GN_FINAL: [B][C][2] {scale, shift}, then [B][C][2] {mean, rstd}
    double *bstats;          // [B][nchunks2][C][3]  (sum dv, sum dv*xhat, sum xhat)
    float *dx, *daux;
    double *ncsums;          // [B][C][6]: A, Bc, Xh, S1, S2, rstd (gnb_final_kernel)
    float *bco;              // [B][C][3]: rstd*gamma, rstd*S1/m, rstd*S2/m (gnb_final_kernel)
    int B, HW, C, ldX, ldD, ldO, ldDx, ldAux, G, nchunks2, flags;
    unsigned *amax;          // gnb_apply (optional): max |dx| of the pass, as float bits (xl_amax_commit)
    int rev;                 // gnb_stats: walk the images in descending order
};

// dv (gradient w.r.t. v = gn(x)) of one element
__device__ __forceinline__ float gnb_dv(float dout, float outAct, float v, int flags)
{
    float t = dout;
    if ((flags & XL_GN_RELU_OUT) && !(outAct > 0.f)) t = 0.f;
    if ((flags & XL_GN_RELU_IN) && !(v > 0.f)) t = 0.f;
    return t;
}

// grid (nchunks2, B), T threads, T % (C/4) == 0
__global__ void gnb_stats_kernel(GnbArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double smemD[];
    const int T = blockDim.x, tid = threadIdx.x;
    const int C4 = a.C >> 2;
    float *sCo = reinterpret_cast<float *>(smemD + (size_t)T * 12);          // mean, rstd, sc, sh per channel
    // images in DESCENDING order (a.rev; XL_GNB_REVERSE=0: ascending): the apply pass walks ascending and so starts with what this
    // pass read last - its first reads are still on the die (apply 118.9 -> 114.7 us per launch at batch 16; this pass itself does
    // not gain from meeting its producer's last writes: 85.7 -> 85.2)
    const int n = a.rev ? (int)gridDim.y - 1 - (int)blockIdx.y : (int)blockIdx.y, chunk = blockIdx.x;
    {
        const float *scsh = a.fco + (long long)n * a.C * 2;
        const float *murs = a.fco + ((long long)a.B + n) * a.C * 2;
        for (int c = tid; c < a.C; c += T) {
            sCo[c] = murs[2 * c]; sCo[a.C + c] = murs[2 * c + 1]; sCo[2 * a.C + c] = scsh[2 * c]; sCo[3 * a.C + c] = scsh[2 * c + 1];
        }
    }
    __syncthreads();
    const int c4 = tid % C4, prow = tid / C4, rows = T / C4;
    const int per = (a.HW + a.nchunks2 - 1) / a.nchunks2;
    const int p0 = chunk * per;
    int p1 = p0 + per; if (p1 > a.HW) p1 = a.HW;
    double sA[4] = { 0, 0, 0, 0 }, sB[4] = { 0, 0, 0, 0 }, sX[4] = { 0, 0, 0, 0 };
    const int c = 4 * c4;
    // the thread's channel quad is fixed: its coefficients live in registers; four pixels per trip with all their loads issued
    // before the first use (round 4: one pixel per trip left 2 - 3 loads in flight per thread, 3.8 TB/s).  The sums are
    // accumulated pixel by pixel in the same order as before.
    f32x4 cMu, cRs, cSc, cSh;
#pragma unroll
    for (int j = 0; j < 4; ++j) { cMu[j] = sCo[c + j]; cRs[j] = sCo[a.C + c + j]; cSc[j] = sCo[2 * a.C + c + j]; cSh[j] = sCo[3 * a.C + c + j]; }
    const bool hasOut = (a.flags & XL_GN_RELU_OUT) != 0;
    auto accumulate = [&](const f32x4 &xv, const f32x4 &dv4, const f32x4 &ov) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float v = xv[j] * cSc[j] + cSh[j];
            const float xh = (xv[j] - cMu[j]) * cRs[j];
            const float dv = gnb_dv(dv4[j], ov[j], v, a.flags);
            sA[j] += (double)dv; sB[j] += (double)dv * (double)xh; sX[j] += (double)xh;
        }
    };
    int p = p0 + prow;
    for (; p + 3 * rows < p1; p += 4 * rows) {
        f32x4 xv[4], dv4[4], ov[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long long pix = (long long)n * a.HW + p + k * rows;
            xv[k] = *reinterpret_cast<const f32x4 *>(a.x + pix * a.ldX + c);
            dv4[k] = *reinterpret_cast<const f32x4 *>(a.dout + pix * a.ldD + c);
            ov[k] = f32x4{ 1.f, 1.f, 1.f, 1.f };
            if (hasOut) ov[k] = *reinterpret_cast<const f32x4 *>(a.outAct + pix * a.ldO + c);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) accumulate(xv[k], dv4[k], ov[k]);
    }
    for (; p < p1; p += rows) {
        const long long pix = (long long)n * a.HW + p;
        const f32x4 xv = *reinterpret_cast<const f32x4 *>(a.x + pix * a.ldX + c);
        const f32x4 dv4 = *reinterpret_cast<const f32x4 *>(a.dout + pix * a.ldD + c);
        f32x4 ov = { 1.f, 1.f, 1.f, 1.f };
        if (hasOut) ov = *reinterpret_cast<const f32x4 *>(a.outAct + pix * a.ldO + c);
        accumulate(xv, dv4, ov);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { smemD[tid * 12 + j] = sA[j]; smemD[tid * 12 + 4 + j] = sB[j]; smemD[tid * 12 + 8 + j] = sX[j]; }
    __syncthreads();
    for (int ch = tid; ch < a.C; ch += T) {
        double A = 0.0, Bc = 0.0, X = 0.0;
        for (int r = 0; r < rows; ++r) {
            const int th = r * C4 + (ch >> 2), sl = ch & 3;
            A += smemD[th * 12 + sl]; Bc += smemD[th * 12 + 4 + sl]; X += smemD[th * 12 + 8 + sl];
        }
        double *o = a.bstats + (((long long)n * a.nchunks2 + chunk) * a.C + ch) * 3;
        o[0] = A; o[1] = Bc; o[2] = X;
    }
}

// Totals of the per-chunk sums and everything the streaming apply pass needs per (image, channel), computed once per
// image instead of in the prologue of every apply workgroup.  grid (B, C / CB), 1024 threads; a workgroup owns CB
// channels (whole groups); 4 threads per channel each sum a contiguous quarter of the chunks, the quarters are added in
// a fixed order.
__global__ __launch_bounds__(1024)
void gnb_final_kernel(GnbArgs a, int CB)
{
    extern __shared__ __attribute__((aligned(16))) double dA[];      // A, Bc, Xh totals per channel, then S1, S2 per group
    const int tid = threadIdx.x, n = blockIdx.x;
    const int C = a.C, cpg = C / a.G;
    const int c0 = blockIdx.y * CB, g0 = c0 / cpg, GB = CB / cpg;
    double *sS = dA + 3 * (size_t)CB;
    const int part = tid & 3;
    const int per = (a.nchunks2 + 3) >> 2;
    const int k0 = part * per;
    int k1 = k0 + per; if (k1 > a.nchunks2) k1 = a.nchunks2;
    for (int cl = tid >> 2; cl < CB; cl += 256) {
        double A = 0.0, Bc = 0.0, X = 0.0;
        const double *bs = a.bstats + ((long long)n * a.nchunks2 * C + c0 + cl) * 3;
#pragma unroll 4
        for (int k = k0; k < k1; ++k) { A += bs[(long long)k * C * 3]; Bc += bs[(long long)k * C * 3 + 1]; X += bs[(long long)k * C * 3 + 2]; }
        // quarters 0+1 and 2+3, then the halves: the same order for every channel
        A += __shfl_xor(A, 1); Bc += __shfl_xor(Bc, 1); X += __shfl_xor(X, 1);
        A += __shfl_xor(A, 2); Bc += __shfl_xor(Bc, 2); X += __shfl_xor(X, 2);
        if (part == 0) { dA[3 * cl] = A; dA[3 * cl + 1] = Bc; dA[3 * cl + 2] = X; }
    }
    __syncthreads();
    for (int gl = tid; gl < GB; gl += 1024) {
        double S1 = 0.0, S2 = 0.0;
        for (int cl = gl * cpg; cl < (gl + 1) * cpg; ++cl) {
            S1 += (double)a.gamma[c0 + cl] * dA[3 * cl]; S2 += (double)a.gamma[c0 + cl] * dA[3 * cl + 1];
        }
        sS[2 * gl] = S1; sS[2 * gl + 1] = S2;
    }
    __syncthreads();
    const double m = (double)cpg * (double)a.HW;
    const float *murs = a.fco + ((long long)a.B + n) * C * 2;
    for (int cl = tid; cl < CB; cl += 1024) {
        const int c = c0 + cl, gl = cl / cpg;
        const double rs = (double)murs[2 * c + 1];
        float *bo = a.bco + ((long long)n * C + c) * 3;
        bo[0] = (float)(rs * (double)a.gamma[c]);
        bo[1] = (float)(rs * sS[2 * gl] / m);
        bo[2] = (float)(rs * sS[2 * gl + 1] / m);
        double *o = a.ncsums + ((long long)n * C + c) * 6;
        o[0] = dA[3 * cl]; o[1] = dA[3 * cl + 1]; o[2] = dA[3 * cl + 2]; o[3] = sS[2 * gl]; o[4] = sS[2 * gl + 1]; o[5] = rs;
    }
    (void)g0;
}

// grid (achunks, B), 256 threads
__global__ __launch_bounds__(256)
void gnb_apply_kernel(GnbArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float sK[];      // per channel: mean, rstd, sc, sh, k1, k2, k3
    const int tid = threadIdx.x, n = blockIdx.y;
    const int C = a.C;
    {
        const float *scsh = a.fco + (long long)n * C * 2;
        const float *murs = a.fco + ((long long)a.B + n) * C * 2;
        const float *bo = a.bco + (long long)n * C * 3;
        for (int c = tid; c < C; c += 256) {
            sK[c] = murs[2 * c]; sK[C + c] = murs[2 * c + 1]; sK[2 * C + c] = scsh[2 * c]; sK[3 * C + c] = scsh[2 * c + 1];
            sK[4 * C + c] = bo[3 * c]; sK[5 * C + c] = bo[3 * c + 1]; sK[6 * C + c] = bo[3 * c + 2];
        }
    }
    __syncthreads();
    const int C4 = C >> 2;
    const int achunks = gridDim.x;
    const int per = (a.HW + achunks - 1) / achunks;
    const int p0 = blockIdx.x * per;
    int p1 = p0 + per; if (p1 > a.HW) p1 = a.HW;
    const long long nElem4 = (long long)(p1 - p0) * C4;
    if (C4 <= 256 && 256 % C4 == 0) {
        // the usual case (C = 32 ... 1024, a power of two): a thread keeps ONE channel quad - its seven coefficients in
        // registers, no 64-bit division per element - and walks the pixels, four per trip with all loads issued first
        // (round 4: 4.5 -> TB/s).  Element by element the same arithmetic as the general loop below.
        const int c = 4 * (tid % C4), rows = 256 / C4;
        f32x4 kMu, kRs, kSc, kSh, k1, k2, k3;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            kMu[j] = sK[c + j]; kRs[j] = sK[C + c + j]; kSc[j] = sK[2 * C + c + j]; kSh[j] = sK[3 * C + c + j];
            k1[j] = sK[4 * C + c + j]; k2[j] = sK[5 * C + c + j]; k3[j] = sK[6 * C + c + j];
        }
        const bool hasOut = (a.flags & XL_GN_RELU_OUT) != 0, reluIn = (a.flags & XL_GN_RELU_IN) != 0;
        const bool add = (a.flags & XL_GN_ADD) != 0, accAux = (a.flags & XL_GN_ACC_AUX) != 0;
        float mx = 0.f;
        auto one = [&](long long pix, const f32x4 &xv, const f32x4 &d4, const f32x4 &ov, const f32x4 &old) {
            f32x4 dx, t4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = xv[j] * kSc[j] + kSh[j];
                const float xh = (xv[j] - kMu[j]) * kRs[j];
                float t = d4[j];
                if (hasOut && !(ov[j] > 0.f)) t = 0.f;
                t4[j] = t;
                const float dv = (reluIn && !(v > 0.f)) ? 0.f : t;
                dx[j] = k1[j] * dv - k2[j] - xh * k3[j];
            }
            if (add) {
                if (accAux) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) t4[j] += old[j];
                }
                *reinterpret_cast<f32x4 *>(a.daux + pix * a.ldAux + c) = t4;
            }
            *reinterpret_cast<f32x4 *>(a.dx + pix * a.ldDx + c) = dx;
            mx = fmaxf(fmaxf(mx, fmaxf(fabsf(dx[0]), fabsf(dx[1]))), fmaxf(fabsf(dx[2]), fabsf(dx[3])));
        };
        const f32x4 ones = { 1.f, 1.f, 1.f, 1.f };
        int p = p0 + tid / C4;
        for (; p + 3 * rows < p1; p += 4 * rows) {
            f32x4 xv[4], d4[4], ov[4], old[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const long long pix = (long long)n * a.HW + p + k * rows;
                xv[k] = *reinterpret_cast<const f32x4 *>(a.x + pix * a.ldX + c);
                d4[k] = *reinterpret_cast<const f32x4 *>(a.dout + pix * a.ldD + c);
                ov[k] = hasOut ? *reinterpret_cast<const f32x4 *>(a.outAct + pix * a.ldO + c) : ones;
                old[k] = (add && accAux) ? *reinterpret_cast<const f32x4 *>(a.daux + pix * a.ldAux + c) : ones;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) one((long long)n * a.HW + p + k * rows, xv[k], d4[k], ov[k], old[k]);
        }
        for (; p < p1; p += rows) {
            const long long pix = (long long)n * a.HW + p;
            const f32x4 xv = *reinterpret_cast<const f32x4 *>(a.x + pix * a.ldX + c);
            const f32x4 d4 = *reinterpret_cast<const f32x4 *>(a.dout + pix * a.ldD + c);
            const f32x4 ov = hasOut ? *reinterpret_cast<const f32x4 *>(a.outAct + pix * a.ldO + c) : ones;
            const f32x4 old = (add && accAux) ? *reinterpret_cast<const f32x4 *>(a.daux + pix * a.ldAux + c) : ones;
            one(pix, xv, d4, ov, old);
        }
        xl_amax_commit(mx, a.amax);
        return;
    }
    float mxs = 0.f;
    for (long long f = tid; f < nElem4; f += 256) {
        const int p = p0 + (int)(f / C4);
        const int c = (int)(f - (long long)(p - p0) * C4) * 4;
        const long long pix = (long long)n * a.HW + p;
        const f32x4 xv = *reinterpret_cast<const f32x4 *>(a.x + pix * a.ldX + c);
        const f32x4 d4 = *reinterpret_cast<const f32x4 *>(a.dout + pix * a.ldD + c);
        f32x4 ov = { 1.f, 1.f, 1.f, 1.f };
        if (a.flags & XL_GN_RELU_OUT) ov = *reinterpret_cast<const f32x4 *>(a.outAct + pix * a.ldO + c);
        f32x4 dx, t4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float v = xv[j] * sK[2 * C + c + j] + sK[3 * C + c + j];
            const float xh = (xv[j] - sK[c + j]) * sK[C + c + j];
            float t = d4[j];
            if ((a.flags & XL_GN_RELU_OUT) && !(ov[j] > 0.f)) t = 0.f;
            t4[j] = t;
            const float dv = ((a.flags & XL_GN_RELU_IN) && !(v > 0.f)) ? 0.f : t;
            dx[j] = sK[4 * C + c + j] * dv - sK[5 * C + c + j] - xh * sK[6 * C + c + j];
        }
        if (a.flags & XL_GN_ADD) {
            float *q = a.daux + pix * a.ldAux + c;
            if (a.flags & XL_GN_ACC_AUX) {
                const f32x4 old = *reinterpret_cast<const f32x4 *>(q);
#pragma unroll
                for (int j = 0; j < 4; ++j) t4[j] += old[j];
            }
            *reinterpret_cast<f32x4 *>(q) = t4;
        }
        *reinterpret_cast<f32x4 *>(a.dx + pix * a.ldDx + c) = dx;
        mxs = fmaxf(fmaxf(mxs, fmaxf(fabsf(dx[0]), fabsf(dx[1]))), fmaxf(fabsf(dx[2]), fabsf(dx[3])));
    }
    xl_amax_commit(mxs, a.amax);
}

// d gamma, d beta, d conv-bias from the per-(image, channel) sums; one thread per channel
__global__ void gnb_params_kernel(const double *ncsums, const float *gamma, int B, int C, int G, int HW,
                                  float *dgamma, float *dbeta, float *dbias)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const int cpg = C / G;
    const double m = (double)cpg * (double)HW;
    double dg = 0.0, db = 0.0, dbi = 0.0;
    for (int n = 0; n < B; ++n) {
        const double *o = ncsums + ((long long)n * C + c) * 6;
        dg += o[1]; db += o[0];
        // sum over pixels of dx = rstd*(gamma*A - (HW*S1 + S2*sum_xhat)/m)
        dbi += o[5] * ((double)gamma[c] * o[0] - ((double)HW * o[3] + o[4] * o[2]) / m);
    }
    dgamma[c] = (float)dg; dbeta[c] = (float)db;
    // one channel per group = instance norm: a per-channel bias cancels exactly, its gradient is identically 0
    if (dbias) dbias[c] = (cpg == 1) ? 0.f : (float)dbi;
}

// the same for a list of layers: blockIdx.y = layer (XL_OP_GNB_PARAMS_LIST)
__global__ void gnb_params_list_kernel(const xl_gnb_params_item *__restrict__ items)
{
    const xl_gnb_params_item it = items[blockIdx.y];
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= it.C) return;
    const int cpg = it.C / it.G;
    const double m = (double)cpg * (double)it.HW;
    const double gam = (double)it.gamma[c];
    double dg = 0.0, db = 0.0, dbi = 0.0;
    for (int n = 0; n < it.B; ++n) {
        const double *o = it.sums + ((long long)n * it.C + c) * 6;
        dg += o[1]; db += o[0];
        dbi += o[5] * (gam * o[0] - ((double)it.HW * o[3] + o[4] * o[2]) / m);
    }
    it.dgamma[c] = (float)dg; it.dbeta[c] = (float)db;
    if (it.dbias) it.dbias[c] = (cpg == 1) ? 0.f : (float)dbi;
}

// ---------------------------------------------------------------------------------------------- head backward

// forward: s_o = w_o . x + b_o; out_o = s_o + mean (o < nTask); out_o = exp(clamp(s_o, lo, hi)) otherwise.
// dout/fout NCHW [B][Cout][HW]; x NHWC; dx NHWC; partial dW [blocks*4 waves][Cout][Cin], partial db [..][Cout]
template <int COUT_MAX, int NQ_MAX>
__global__ __launch_bounds__(256)
void head_bwd_kernel(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ dout,
                     const float *__restrict__ fout, float *__restrict__ dx, float *__restrict__ pW,
                     float *__restrict__ pB, int B, int HW, int Cin, int ldX, int ldDx, int Cout, int nTask,
                     float lo, float hi)
{
    const int lane = threadIdx.x & 63;
    const int waveGlobal = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int nWaves = (gridDim.x * 256) >> 6;
    const int nq = (Cin + 255) >> 8;                          // (trips of 256 channels; lanes past Cin hold nothing)
    const long long total = (long long)B * HW;
    const float elo = expf(lo), ehi = expf(hi);
    f32x4 aw[COUT_MAX][NQ_MAX];
    float ab[COUT_MAX];
#pragma unroll
    for (int o = 0; o < COUT_MAX; ++o) {
        ab[o] = 0.f;
#pragma unroll
        for (int q = 0; q < NQ_MAX; ++q) aw[o][q] = f32x4{ 0.f, 0.f, 0.f, 0.f };
    }
    for (long long p = waveGlobal; p < total; p += nWaves) {
        const int n = (int)(p / HW);
        const int pix = (int)(p - (long long)n * HW);
        float ds[COUT_MAX];
#pragma unroll
        for (int o = 0; o < COUT_MAX; ++o) {
            ds[o] = 0.f;
            if (o < Cout) {
                const long long idx = ((long long)n * Cout + o) * HW + pix;
                float g = dout[idx];
                if (o >= nTask) {
                    const float fo = fout[idx];
                    g = (fo > elo && fo < ehi) ? g * fo : 0.f;            // d exp(clamp(s)) / ds
                }
                ds[o] = g;
                ab[o] += g;
            }
        }
#pragma unroll
        for (int q = 0; q < NQ_MAX; ++q) {
            if (q < nq && q * 256 + lane * 4 < Cin) {
                const int c = q * 256 + lane * 4;
                const f32x4 xv = *reinterpret_cast<const f32x4 *>(x + p * ldX + c);
                f32x4 d = { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
                for (int o = 0; o < COUT_MAX; ++o) {
                    if (o < Cout) {
                        const f32x4 wv = *reinterpret_cast<const f32x4 *>(w + (long long)o * Cin + c);
#pragma unroll
                        for (int j = 0; j < 4; ++j) { d[j] = fmaf(ds[o], wv[j], d[j]); aw[o][q][j] = fmaf(ds[o], xv[j], aw[o][q][j]); }
                    }
                }
                *reinterpret_cast<f32x4 *>(dx + p * ldDx + c) = d;
            }
        }
    }
#pragma unroll
    for (int o = 0; o < COUT_MAX; ++o) {
        if (o < Cout) {
#pragma unroll
            for (int q = 0; q < NQ_MAX; ++q)
                if (q < nq && q * 256 + lane * 4 < Cin)
                    *reinterpret_cast<f32x4 *>(pW + ((long long)waveGlobal * Cout + o) * Cin + q * 256 + lane * 4) = aw[o][q];
            if (lane == 0) pB[(long long)waveGlobal * Cout + o] = ab[o];
        }
    }
}

// Backward of the full-size (semantics) head for the pure pixel-shuffle case (H = 8 Hs, W = 8 Ws; duc_head_kernel in
// xl_cnn.hip): one output pixel per thread.  dz = dout (x fout inside the clamp for exp channels); the gradient of the
// DUC activation goes back through the shuffle index map (every input element has exactly one output pixel: plain
// stores, no atomics); d fc3.weight / d fc3.bias are accumulated per thread, reduced per block in a fixed order and
// written as one partial vector [C*C + C] per block.
// TRIM (H x W differs from the shuffled 8*Hs x 8*Ws grid, networks.py:344-349: F.interpolate(bilinear) trims it): the
// thread recomputes the interpolated activation of its output pixel for d fc3.weight and writes d(interpolated
// activation) [B][C][H][W] to `dv`; duc_trim_bwd_kernel gathers that back onto the shuffled grid.
template <int CMAX, int TRIM>
__global__ __launch_bounds__(256)
void duc_head_bwd_kernel(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ dout,
                         const float *__restrict__ fout, float *__restrict__ dx, float *__restrict__ partial,
                         int B, int Hs, int Ws, int C, int ldX, int ldDx, int nTask, float lo, float hi,
                         int Ho, int Wo, float *__restrict__ dv)
{
    __shared__ float sRed[4][CMAX * CMAX + CMAX];
    const int H = TRIM ? Ho : 8 * Hs, W = TRIM ? Wo : 8 * Ws;
    const int Hu = 8 * Hs, Wu = 8 * Ws;
    const float sy = (float)Hu / (float)H, sx = (float)Wu / (float)W;
    const float elo = expf(lo), ehi = expf(hi);
    const long long HW = (long long)H * W, total = (long long)B * HW;
    float aW[CMAX][CMAX], aB[CMAX];
#pragma unroll
    for (int o = 0; o < CMAX; ++o) {
        aB[o] = 0.f;
#pragma unroll
        for (int c = 0; c < CMAX; ++c) aW[o][c] = 0.f;
    }
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < total; p += (long long)gridDim.x * 256) {
        const int xx = (int)(p % W);
        const int y = (int)((p / W) % H);
        const int n = (int)(p / HW);
        const long long pix = (long long)y * W + xx;
        const long long src = (((long long)n * Hs + (y >> 3)) * Ws + (xx >> 3));
        const int sub = (y & 7) * 8 + (xx & 7);
        // TRIM: the four taps of the forward interpolation (same float arithmetic as duc_head_kernel)
        long long a00 = 0, a01 = 0, a10 = 0, a11 = 0;
        float ly0 = 1.f, ly1 = 0.f, lx0 = 1.f, lx1 = 0.f;
        if (TRIM) {
            float fy = sy * ((float)y + 0.5f) - 0.5f; if (fy < 0.f) fy = 0.f;
            float fx = sx * ((float)xx + 0.5f) - 0.5f; if (fx < 0.f) fx = 0.f;
            const int y0 = (int)fy, x0 = (int)fx;
            const int y1 = y0 + (y0 < Hu - 1 ? 1 : 0), x1 = x0 + (x0 < Wu - 1 ? 1 : 0);
            ly1 = fy - (float)y0; lx1 = fx - (float)x0; ly0 = 1.f - ly1; lx0 = 1.f - lx1;
            const long long img = (long long)n * Hs * Ws * ldX;
            a00 = img + ((long long)(y0 >> 3) * Ws + (x0 >> 3)) * ldX + (y0 & 7) * 8 + (x0 & 7);
            a01 = img + ((long long)(y0 >> 3) * Ws + (x1 >> 3)) * ldX + (y0 & 7) * 8 + (x1 & 7);
            a10 = img + ((long long)(y1 >> 3) * Ws + (x0 >> 3)) * ldX + (y1 & 7) * 8 + (x0 & 7);
            a11 = img + ((long long)(y1 >> 3) * Ws + (x1 >> 3)) * ldX + (y1 & 7) * 8 + (x1 & 7);
        }
        float dz[CMAX], v[CMAX];
#pragma unroll
        for (int o = 0; o < CMAX; ++o) {
            dz[o] = 0.f; v[o] = 0.f;
            if (o < C) {
                const long long idx = ((long long)n * C + o) * HW + pix;
                float g = dout[idx];
                if (o >= nTask) {
                    const float fo = fout[idx];
                    g = (fo > elo && fo < ehi) ? g * fo : 0.f;
                }
                dz[o] = g;
                if (TRIM) v[o] = ly0 * (lx0 * x[a00 + o * 64] + lx1 * x[a01 + o * 64]) + ly1 * (lx0 * x[a10 + o * 64] + lx1 * x[a11 + o * 64]);
                else v[o] = x[src * ldX + o * 64 + sub];
                aB[o] += g;
            }
        }
#pragma unroll
        for (int c = 0; c < CMAX; ++c) {
            if (c < C) {
                float d = 0.f;
#pragma unroll
                for (int o = 0; o < CMAX; ++o) if (o < C) { d = fmaf(w[o * C + c], dz[o], d); aW[o][c] = fmaf(dz[o], v[c], aW[o][c]); }
                if (TRIM) dv[((long long)n * C + c) * HW + pix] = d;
                else dx[src * ldDx + c * 64 + sub] = d;
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 0; o < CMAX; ++o) {
#pragma unroll
        for (int c = 0; c < CMAX; ++c) {
            float t = aW[o][c];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) t += __shfl_xor(t, off);
            if (lane == 0) sRed[wave][o * CMAX + c] = t;
        }
        float t = aB[o];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) t += __shfl_xor(t, off);
        if (lane == 0) sRed[wave][CMAX * CMAX + o] = t;
    }
    __syncthreads();
    const int len = C * C + C;
    if ((int)threadIdx.x < len) {
        const int i = threadIdx.x;
        const int k2 = (i < C * C) ? (i / C) * CMAX + (i % C) : CMAX * CMAX + (i - C * C);
        partial[(long long)blockIdx.x * len + i] = ((sRed[0][k2] + sRed[1][k2]) + sRed[2][k2]) + sRed[3][k2];
    }
}

// Backward of the bilinear trim: one thread per element (n, yy, xx) of the shuffled 8*Hs x 8*Ws grid gathers
// dv [B][C][H][W] over the output pixels whose interpolation stencil contains it, with the forward weights, in a fixed
// order (no atomics).  Rows: output rows y with y0(y) == yy or y1(y) == yy lie in a window around yy / scale.
template <int CMAX>
__global__ __launch_bounds__(256)
void duc_trim_bwd_kernel(const float *__restrict__ dv, float *__restrict__ dx, int B, int Hs, int Ws, int C, int ldDx, int H, int W)
{
    const int Hu = 8 * Hs, Wu = 8 * Ws;
    const float sy = (float)Hu / (float)H, sx = (float)Wu / (float)W;
    const long long total = (long long)B * Hu * Wu;
    const long long HW = (long long)H * W;
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < total; p += (long long)gridDim.x * 256) {
        const int xx = (int)(p % Wu);
        const int yy = (int)((p / Wu) % Hu);
        const int n = (int)(p / ((long long)Wu * Hu));
        int ylo = (int)floorf(((float)yy - 1.5f) / sy) - 1, yhi = (int)ceilf(((float)yy + 1.5f) / sy) + 1;
        int xlo = (int)floorf(((float)xx - 1.5f) / sx) - 1, xhi = (int)ceilf(((float)xx + 1.5f) / sx) + 1;
        if (ylo < 0) ylo = 0;
        if (xlo < 0) xlo = 0;
        if (yhi > H - 1) yhi = H - 1;
        if (xhi > W - 1) xhi = W - 1;
        float acc[CMAX];
#pragma unroll
        for (int c = 0; c < CMAX; ++c) acc[c] = 0.f;
        for (int y = ylo; y <= yhi; ++y) {
            float fy = sy * ((float)y + 0.5f) - 0.5f; if (fy < 0.f) fy = 0.f;
            const int y0 = (int)fy, y1 = y0 + (y0 < Hu - 1 ? 1 : 0);
            const float ly1 = fy - (float)y0, ly0 = 1.f - ly1;
            const float wy = (y0 == yy ? ly0 : 0.f) + (y1 == yy ? ly1 : 0.f);
            if (wy == 0.f) continue;
            for (int x = xlo; x <= xhi; ++x) {
                float fx = sx * ((float)x + 0.5f) - 0.5f; if (fx < 0.f) fx = 0.f;
                const int x0 = (int)fx, x1 = x0 + (x0 < Wu - 1 ? 1 : 0);
                const float lx1 = fx - (float)x0, lx0 = 1.f - lx1;
                const float wx = (x0 == xx ? lx0 : 0.f) + (x1 == xx ? lx1 : 0.f);
                if (wx == 0.f) continue;
                const float wgt = wy * wx;
#pragma unroll
                for (int c = 0; c < CMAX; ++c)
                    if (c < C) acc[c] = fmaf(wgt, dv[((long long)n * C + c) * HW + (long long)y * W + x], acc[c]);
            }
        }
        const long long src = (((long long)n * Hs + (yy >> 3)) * Ws + (xx >> 3));
        const int sub = (yy & 7) * 8 + (xx & 7);
#pragma unroll
        for (int c = 0; c < CMAX; ++c)
            if (c < C) dx[src * ldDx + c * 64 + sub] = acc[c];
    }
}

// sums `count` partial vectors of length `len` in a fixed order.  grid ceil(len / 32), 256 threads = 32 elements x 8
// parts: a part sums a contiguous eighth of the vectors with 4 loads in flight, the eighths are added in order (one
// thread per element walking all `count` vectors is a serial chain of `count` load latencies).
__global__ __launch_bounds__(256)
void partial_sum_kernel(const float *__restrict__ partial, float *__restrict__ out, int count, int len)
{
    __shared__ double sP[8 * 32];
    const int e = threadIdx.x & 31, part = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + e;
    const int per = (count + 7) >> 3;
    const int k0 = part * per;
    int k1 = k0 + per; if (k1 > count) k1 = count;
    double s = 0.0;
    if (i < len) {
#pragma unroll 4
        for (int k = k0; k < k1; ++k) s += (double)partial[(long long)k * len + i];
    }
    sP[part * 32 + e] = s;
    __syncthreads();
    if (part == 0 && i < len) {
        double t = 0.0;
        for (int q = 0; q < 8; ++q) t += sP[q * 32 + e];
        out[i] = (float)t;
    }
}

// ---------------------------------------------------------------------------------------------- conv1 wgrad

// dW[o][c][ky][kx] = sum dY[n,y,x,o] * img[n,c,y+ky-1,x+kx-1]; db[o] = sum dY.  Cout = 32.
// Thread = (4 consecutive output channels og, pixel lane pl of 32): per pixel one 16-byte dY load (the 8 threads of a
// pixel read its 128 contiguous bytes) and 9 ds_read_b128 of the image taps feed 108 FMAs - 12 per LDS read; with one
// output channel per thread (3 per read) the kernel sat on the LDS pipe at 6x its HBM time.
// A block walks `rowsPerBlock` output rows of one image two at a time; the 4 image rows (with halo, channels padded
// to a float4) they touch are staged in LDS.  partial [gridDim.y * gridDim.x][28][32]
// FOLD (round 5): `dy` is the gradient w.r.t. the GroupNorm(+ReLU) OUTPUT of conv1 and `x` conv1's raw output: the apply pass of
// the GroupNorm backward (dx = k1 dv - k2 - xhat k3, the arithmetic of gnb_apply_kernel - same operations, same order, same bits)
// runs on load, from the forward table `fco` and the coefficients `bco` XL_OP_GNB_FINAL left.  conv1 has no data gradient: its dx
// had this kernel as only reader, and the apply pass read and wrote 2.1 GB for it at batch 16.
template <bool FOLD>
__global__ __launch_bounds__(256)
void conv1_wgrad_kernel(const float *__restrict__ img, const float *__restrict__ dy, float *__restrict__ partial,
                        int B, int Cin, int H, int W, int ldY, int rowsPerBlock,
                        const float *__restrict__ x, int ldX, const float *__restrict__ fco, const float *__restrict__ bco, int reluIn)
{
    constexpr int R = 2, CO = 32;
    extern __shared__ __attribute__((aligned(16))) float sDyn[];
    f32x4 *sImg = reinterpret_cast<f32x4 *>(sDyn);                 // [(R+2)][W+2]
    float *sRed = sDyn;                                            // [4 waves][32][28], reuses the tile after the loop
    const int og = threadIdx.x & 7, pl = threadIdx.x >> 3;
    const int n = blockIdx.y;
    const long long HW = (long long)H * W;
    const int yBeg = blockIdx.x * rowsPerBlock;
    int yEnd = yBeg + rowsPerBlock; if (yEnd > H) yEnd = H;
    f32x4 acc[28];                                                 // [tap*3 + c] (27: bias) x 4 output channels
#pragma unroll
    for (int k = 0; k < 28; ++k) acc[k] = f32x4{ 0.f, 0.f, 0.f, 0.f };
    f32x4 kMu, kRs, kSc, kSh, k1, k2, k3;                          // FOLD: this thread's four channels of image n
    if constexpr (FOLD) {
        const float *scsh = fco + ((long long)n * CO + 4 * og) * 2, *murs = fco + (((long long)B + n) * CO + 4 * og) * 2;
        const float *bo = bco + ((long long)n * CO + 4 * og) * 3;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            kSc[j] = scsh[2 * j]; kSh[j] = scsh[2 * j + 1]; kMu[j] = murs[2 * j]; kRs[j] = murs[2 * j + 1];
            k1[j] = bo[3 * j]; k2[j] = bo[3 * j + 1]; k3[j] = bo[3 * j + 2];
        }
    }
    auto grad_of = [&](const f32x4 &d4, const f32x4 &xv) {
        if constexpr (!FOLD) return d4;
        f32x4 dx;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float v = xv[j] * kSc[j] + kSh[j];
            const float xh = (xv[j] - kMu[j]) * kRs[j];
            const float dv = (reluIn && !(v > 0.f)) ? 0.f : d4[j];
            dx[j] = k1[j] * dv - k2[j] - xh * k3[j];
        }
        return dx;
    };
    const int W2 = W + 2;
    for (int y0 = yBeg; y0 < yEnd; y0 += R) {
        __syncthreads();
        // staging in batches of 4 entries per thread: all loads of a batch are in flight before the first LDS write
        // (clamped addresses + a select instead of a branch around the loads)
        for (int base = threadIdx.x; base < (R + 2) * W2; base += 4 * 256) {
            f32x4 v[4];
            bool ok[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + u * 256;
                const int r = idx / W2, xx = idx - r * W2 - 1;
                const int iy = y0 - 1 + r;
                ok[u] = (idx < (R + 2) * W2) & ((unsigned)iy < (unsigned)H) & ((unsigned)xx < (unsigned)W);
                const int iyc = min(max(iy, 0), H - 1), xc = min(max(xx, 0), W - 1);
                const float *q = img + (long long)n * Cin * HW + (long long)iyc * W + xc;
                v[u] = f32x4{ q[0], Cin > 1 ? q[HW] : 0.f, Cin > 1 ? q[2 * HW] : 0.f, 0.f };
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + u * 256;
                if (idx < (R + 2) * W2) sImg[idx] = ok[u] ? v[u] : f32x4{ 0.f, 0.f, 0.f, 0.f };
            }
        }
        __syncthreads();
        const int rows = (yEnd - y0 < R) ? (yEnd - y0) : R;
        // rows y0.. are consecutive in memory: pixel p of the step is dyRow + p*ldY.  The load of the next pixel is
        // issued before the FMAs of the current one (one HBM round trip per iteration otherwise).
        const float *dyRow = dy + ((long long)n * H + y0) * W * ldY + 4 * og;
        const float *xRow = FOLD ? x + ((long long)n * H + y0) * W * ldX + 4 * og : nullptr;
        const int np = rows * W;
        f32x4 gNext = f32x4{ 0.f, 0.f, 0.f, 0.f }, xNext = f32x4{ 0.f, 0.f, 0.f, 0.f };
        if (pl < np) {
            gNext = *reinterpret_cast<const f32x4 *>(dyRow + (long long)pl * ldY);
            if constexpr (FOLD) xNext = *reinterpret_cast<const f32x4 *>(xRow + (long long)pl * ldX);
        }
        for (int p = pl; p < np; p += 32) {
            const int ry = p / W, xcol = p - ry * W;
            const f32x4 g = grad_of(gNext, xNext);
            if (p + 32 < np) {
                gNext = *reinterpret_cast<const f32x4 *>(dyRow + (long long)(p + 32) * ldY);
                if constexpr (FOLD) xNext = *reinterpret_cast<const f32x4 *>(xRow + (long long)(p + 32) * ldX);
            }
            acc[27] += g;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const f32x4 *row = sImg + (ry + ky) * W2 + xcol;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const f32x4 v = row[kx];
                    acc[(ky * 3 + kx) * 3 + 0] += g * v.x;
                    acc[(ky * 3 + kx) * 3 + 1] += g * v.y;
                    acc[(ky * 3 + kx) * 3 + 2] += g * v.z;
                }
            }
        }
    }
    // fixed-order reduction: the 8 pixel lanes of a wave (lane bits 3..5), then the 4 waves through LDS
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 28; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = acc[k][e];
            v += __shfl_xor(v, 8);
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            acc[k][e] = v;
        }
    __syncthreads();
    if (lane < 8) {
#pragma unroll
        for (int k = 0; k < 28; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) sRed[(wave * CO + 4 * og + e) * 28 + k] = acc[k][e];
    }
    __syncthreads();
    const long long blk = (long long)blockIdx.y * gridDim.x + blockIdx.x;
    for (int i = threadIdx.x; i < 28 * CO; i += 256) {
        const int k = i / CO, o = i - k * CO;
        float s = 0.f;
        for (int w = 0; w < 4; ++w) s += sRed[(w * CO + o) * 28 + k];
        partial[(blk * 28 + k) * CO + o] = s;
    }
}

// out: dW OIHW [Cout][Cin][3][3] and db[Cout] from the block partials.  grid (28): one tap/channel slot k per block;
// 256 threads = 32 output channels x 8 parts, each part sums a contiguous eighth of the blocks (4 loads in flight), the
// eighths are added in order.  (One thread per output walking all partials serially was a 0.5 ms chain of loads.)
__global__ __launch_bounds__(256)
void conv1_wgrad_reduce_kernel(const float *__restrict__ partial, float *__restrict__ dw, float *__restrict__ db,
                               int blocks, int Cin)
{
    constexpr int CO = 32;
    __shared__ double sP[8 * CO];
    const int k = blockIdx.x, o = threadIdx.x & 31, part = threadIdx.x >> 5;
    const int per = (blocks + 7) >> 3;
    const int b0 = part * per;
    int b1 = b0 + per; if (b1 > blocks) b1 = blocks;
    double s = 0.0;
#pragma unroll 4
    for (int b = b0; b < b1; ++b) s += (double)partial[((long long)b * 28 + k) * CO + o];
    sP[part * CO + o] = s;
    __syncthreads();
    if (part != 0) return;
    double t = 0.0;
    for (int q = 0; q < 8; ++q) t += sP[q * CO + o];
    if (k == 27) { db[o] = (float)t; return; }
    const int tap = k / 3, c = k - tap * 3;
    if (c < Cin) dw[((long long)o * Cin + c) * 9 + tap] = (float)t;
}

int gnb_threads(int C)
{
    const int C4 = C / 4;
    if (C4 > 256) return C4;
    if (256 % C4 == 0) return 256;
    int T = C4;                                         // e.g. 384 channels: 96 quads -> 192 threads (lcm with 64)
    while (T % 64 != 0) T += C4;
    return T <= 1024 ? T : -1;
}

}  // namespace

int xl_run_bwd_op(const xl_op &op, hipStream_t st)
{
    switch (op.type) {
        case XL_OP_FILL0: {
            if (!op.out || op.Cin < 1) return XL_ERR_ARG;
            return hipMemsetAsync(op.out, 0, (size_t)op.Cin, st) == hipSuccess ? XL_OK : XL_ERR_HIP;
        }
        case XL_OP_WGRAD: {
            if (op.nchunks2 < 1 || op.ld_in % 4 != 0 || op.ld_aux % 4 != 0) return XL_ERR_ARG;
            if (op.flags & XL_CONV_SPLIT_BF16) {                      // 1x1 / batched Winograd products on the bf16 pipe
                if (op.nchunks2 == 1) {
                    // one split (the 64 batched products of a Winograd layer fill the chip without splitting K): the "partial"
                    // tile IS the result, [z][Cout][Cin] - written in place, no reduce pass (it was a 134 MB copy per layer)
                    xl_op direct = op;
                    direct.stats2 = op.out;
                    return (op.flags & XL_CONV_PAIR_F16) ? xl_run_wgrad_pair(direct, st) : xl_run_wgrad_split(direct, st);
                }
                const int rc = (op.flags & XL_CONV_PAIR_F16) ? xl_run_wgrad_pair(op, st) : xl_run_wgrad_split(op, st);
                if (rc != XL_OK) return rc;
                const long long total = (long long)op.Cout * op.Cin;
                long long blocks = (total + 255) / 256;
                if (blocks > 2048) blocks = 2048;
                hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)blocks, op.groups > 1 ? op.groups : 1), dim3(256), 0, st,
                                   (const float *)op.stats2, (float *)op.out, op.nchunks2, 1, op.Cout, op.Cin);
                return XL_OK;
            }
            if (op.Cout % 128 == 0 && op.Cin % 128 == 0) return launch_wgrad<128, 128, 2, 2>(op, st);
            if (op.Cout % 128 == 0 && op.Cin % 64 == 0) return launch_wgrad<128, 64, 2, 2>(op, st);
            if (op.Cout % 64 == 0 && op.Cin % 32 == 0) return launch_wgrad<64, 32, 2, 1>(op, st);
            return XL_ERR_UNSUPPORTED;
        }
        case XL_OP_GNB_STATS:
        case XL_OP_GNB_FINAL:
        case XL_OP_GNB_APPLY: {
            // stats = forward coefficient table of the layer (GN_FINAL with out2: [B][C][2] {scale, shift} followed by
            // [B][C][2] {mean, rstd}); stats2 = scratch: per-chunk sums, per-(image, channel) sums, apply coefficients
            if (op.Cin % 4 != 0 || op.Cin % op.groups != 0 || op.groups > 64 || op.nchunks2 < 1 || !op.stats || !op.stats2)
                return XL_ERR_ARG;
            GnbArgs a;
            a.x = (const float *)op.in; a.dout = (const float *)op.aux; a.outAct = (const float *)op.aux2;
            a.gamma = (const float *)op.w; a.beta = (const float *)op.bias; a.fco = (const float *)op.stats;
            a.bstats = (double *)op.stats2; a.dx = (float *)op.out; a.daux = (float *)op.out2;
            a.ncsums = a.bstats + (long long)op.B * op.nchunks2 * op.Cin * 3;
            a.bco = reinterpret_cast<float *>(a.ncsums + (long long)op.B * op.Cin * 6);
            // (XL_OP_GNB_PARAMS_LIST: the per-(image, channel) sums of this layer go to a buffer of its own, read at the end of the pass)
            if (op.type == XL_OP_GNB_FINAL && op.scale) a.ncsums = (double *)const_cast<void *>(op.scale);
            a.B = op.B; a.HW = op.Hi * op.Wi; a.C = op.Cin; a.ldX = op.ld_in; a.ldD = op.ld_aux; a.ldO = op.ld_out; a.ldDx = op.ld_in;
            a.ldAux = op.Cout > 0 ? op.Cout : op.ld_out; a.G = op.groups; a.nchunks2 = op.nchunks2; a.flags = op.flags;
            static const int gnbRev = getenv("XL_GNB_REVERSE") ? atoi(getenv("XL_GNB_REVERSE")) : 1;
            a.rev = gnbRev;
            a.amax = op.type == XL_OP_GNB_APPLY ? (unsigned *)op.scale : nullptr;      // (round 5: max |dx| for the pair GEMMs that read dx)
            if (op.type == XL_OP_GNB_STATS) {
                const int T = gnb_threads(op.Cin);
                if (T < 0 || T > 1024) return XL_ERR_ARG;
                const size_t lds = sizeof(double) * 12 * T + sizeof(float) * 4 * op.Cin;
                hipLaunchKernelGGL(gnb_stats_kernel, dim3(op.nchunks2, op.B), dim3(T), lds, st, a);
            } else if (op.type == XL_OP_GNB_FINAL) {
                const int cpg = op.Cin / op.groups;
                const int CB = (op.Cin % 64 == 0 && 64 % cpg == 0) ? 64 : op.Cin;      // whole groups per workgroup
                const size_t lds = sizeof(double) * (3 * (size_t)CB + 2 * 64);
                hipLaunchKernelGGL(gnb_final_kernel, dim3(op.B, op.Cin / CB), dim3(1024), lds, st, a, CB);
            } else {
                int achunks = (a.HW * (op.Cin / 4) + 256 * 16 - 1) / (256 * 16);
                if (achunks < 1) achunks = 1;
                if (achunks > 1024) achunks = 1024;
                const size_t lds = sizeof(float) * 7 * op.Cin;
                hipLaunchKernelGGL(gnb_apply_kernel, dim3(achunks, op.B), dim3(256), lds, st, a);
            }
            return XL_OK;
        }
        case XL_OP_GNB_PARAMS: {
            const double *nc = (const double *)op.stats2 + (long long)op.B * op.nchunks2 * op.Cin * 3;
            float *dbias = (op.flags & XL_GN_NO_CONV_BIAS) ? nullptr : (float *)op.aux2;
            hipLaunchKernelGGL(gnb_params_kernel, dim3((op.Cin + 63) / 64), dim3(64), 0, st, nc, (const float *)op.w, op.B,
                               op.Cin, op.groups, op.Hi * op.Wi, (float *)op.out, (float *)op.out2, dbias);
            return XL_OK;
        }
        case XL_OP_GNB_PARAMS_LIST: {
            if (op.Cin < 1 || op.Cin > 65535 || !op.in || op.Cout < 1) return XL_ERR_ARG;      // Cout = the largest channel count
            hipLaunchKernelGGL(gnb_params_list_kernel, dim3((op.Cout + 63) / 64, op.Cin), dim3(64), 0, st,
                               (const xl_gnb_params_item *)op.in);
            return XL_OK;
        }
        case XL_OP_HEAD_BWD: {
            if (op.Cin % 4 != 0 || op.Cin < 4 || op.Cin > 1024 || op.Cout > 4 || op.Cout < 1) return XL_ERR_ARG;
            const long long pix = (long long)op.B * op.Hi * op.Wi;
            long long blocks = (pix + 63) / 64;
            if (blocks > 256) blocks = 256;
            if (blocks < 1) blocks = 1;
            const int waves = (int)blocks * 4;
            float *pW = (float *)op.stats2;
            float *pB = pW + (long long)waves * op.Cout * op.Cin;
            hipLaunchKernelGGL((head_bwd_kernel<4, 4>), dim3((unsigned)blocks), dim3(256), 0, st, (const float *)op.in,
                               (const float *)op.w, (const float *)op.aux, (const float *)op.aux2, (float *)op.out, pW, pB,
                               op.B, op.Hi * op.Wi, op.Cin, op.ld_in, op.ld_out, op.Cout, op.n_task, op.clamp_lo, op.clamp_hi);
            hipLaunchKernelGGL(partial_sum_kernel, dim3((op.Cout * op.Cin + 31) / 32), dim3(256), 0, st, (const float *)pW,
                               (float *)op.out2, waves, op.Cout * op.Cin);
            hipLaunchKernelGGL(partial_sum_kernel, dim3((op.Cout + 31) / 32), dim3(256), 0, st, (const float *)pB,
                               (float *)op.stats, waves, op.Cout);
            return XL_OK;
        }
        case XL_OP_WINO_DY: {
            // ksize = output tile m of F(m x m, 3x3): 4 (default) or 6
            const int m = op.ksize == 6 ? 6 : 4;
            if (op.Cin % 2 != 0 || op.ld_in % 2 != 0 || op.Ho != (op.Hi + m - 1) / m || op.Wo != (op.Wi + m - 1) / m) return XL_ERR_ARG;
            const long long items = (long long)op.B * op.Ho * op.Wo * (op.Cin / 2);
            long long blocks = (items + 255) / 256;
            if (blocks > 262144) blocks = 262144;
            if (m == 6)
                hipLaunchKernelGGL(wino6_dy_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const float *)op.in, (float *)op.out,
                                   op.B, op.Hi, op.Wi, op.Cin, op.ld_in, op.Ho, op.Wo, (unsigned *)op.scale);   // scale: max |dM| slot (optional)
            else
                hipLaunchKernelGGL(wino4_dy_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const float *)op.in, (float *)op.out,
                                   op.B, op.Hi, op.Wi, op.Cin, op.ld_in, op.Ho, op.Wo);
            return XL_OK;
        }
        case XL_OP_WINO_WFINAL: {
            const long long total = (long long)op.Cout * op.Cin;
            long long blocks = (total + 255) / 256;
            if (blocks > 4096) blocks = 4096;
            hipLaunchKernelGGL(op.ksize == 6 ? wino6_wfinal_kernel : wino4_wfinal_kernel, dim3((unsigned)blocks), dim3(256), 0, st,
                               (const float *)op.in, (float *)op.out, op.Cout, op.Cin);
            return XL_OK;
        }
        case XL_OP_DUC_HEAD_BWD: {
            // in: DUC activation [B,Hi,Wi,Cout*64]; aux: dout, aux2: forward output [B,Cout,8Hi,8Wi]; out: d activation;
            // out2: d fc3.weight [Cout][Cout]; stats: d fc3.bias; stats2: scratch (blocks x (Cout^2 + Cout) floats)
            if (op.Cout < 1 || op.Cout > 8 || op.Cin != op.Cout * 64) return XL_ERR_ARG;
            const bool trim = (op.Ho != 8 * op.Hi || op.Wo != 8 * op.Wi);        // networks.py:344-349 bilinear trim
            const long long pix = (long long)op.B * op.Ho * op.Wo;
            long long blocks = (pix + 255) / 256;
            if (blocks > 1024) blocks = 1024;
            const int len = op.Cout * op.Cout + op.Cout;
            float *part = (float *)op.stats2;
            float *tot = part + blocks * len;                           // [len]: weights then bias
            float *dv = tot + len;                                      // trim: [B][Cout][Ho][Wo] behind the partials
            if (trim) {
                hipLaunchKernelGGL((duc_head_bwd_kernel<8, 1>), dim3((unsigned)blocks), dim3(256), 0, st, (const float *)op.in,
                                   (const float *)op.w, (const float *)op.aux, (const float *)op.aux2, (float *)op.out, part,
                                   op.B, op.Hi, op.Wi, op.Cout, op.ld_in, op.ld_out, op.n_task, op.clamp_lo, op.clamp_hi,
                                   op.Ho, op.Wo, dv);
                long long ub = ((long long)op.B * 64 * op.Hi * op.Wi + 255) / 256;
                if (ub > 65536) ub = 65536;
                hipLaunchKernelGGL(duc_trim_bwd_kernel<8>, dim3((unsigned)ub), dim3(256), 0, st, (const float *)dv, (float *)op.out,
                                   op.B, op.Hi, op.Wi, op.Cout, op.ld_out, op.Ho, op.Wo);
            } else
                hipLaunchKernelGGL((duc_head_bwd_kernel<8, 0>), dim3((unsigned)blocks), dim3(256), 0, st, (const float *)op.in,
                                   (const float *)op.w, (const float *)op.aux, (const float *)op.aux2, (float *)op.out, part,
                                   op.B, op.Hi, op.Wi, op.Cout, op.ld_in, op.ld_out, op.n_task, op.clamp_lo, op.clamp_hi,
                                   op.Ho, op.Wo, (float *)nullptr);
            hipLaunchKernelGGL(partial_sum_kernel, dim3((len + 31) / 32), dim3(256), 0, st, (const float *)part, tot, (int)blocks, len);
            if (hipMemcpyAsync(op.out2, tot, sizeof(float) * op.Cout * op.Cout, hipMemcpyDeviceToDevice, st) != hipSuccess ||
                hipMemcpyAsync(op.stats, tot + op.Cout * op.Cout, sizeof(float) * op.Cout, hipMemcpyDeviceToDevice, st) != hipSuccess)
                return XL_ERR_HIP;
            return XL_OK;
        }
        case XL_OP_CONV1_WGRAD: {
            if (op.Cout != 32 || op.Cin > 3) return XL_ERR_UNSUPPORTED;
            const int rowsPerBlock = op.reserved_i > 0 ? op.reserved_i : 16;     // scratch: B*ceil(Hi/rows)*28*Cout floats
            if (op.ld_aux % 4 != 0) return XL_ERR_ARG;
            const int rb = (op.Hi + rowsPerBlock - 1) / rowsPerBlock;
            const int blocks = rb * op.B;
            size_t lds = sizeof(float) * (size_t)4 * (op.Wi + 2) * 4;
            if (lds < sizeof(float) * (size_t)4 * op.Cout * 28) lds = sizeof(float) * (size_t)4 * op.Cout * 28;
            if (lds > 64 * 1024) return XL_ERR_UNSUPPORTED;
            if (op.aux2) {
                // GroupNorm-backward apply on load: aux2 = conv1's raw output (ld_in), w = the layer's forward table (XL_OP_GN_FINAL
                // with out2), bias = the coefficients [B][Cout][3] of XL_OP_GNB_FINAL, flags = the GroupNorm's XL_GN_RELU_IN
                if (!op.w || !op.bias || op.ld_in % 4 != 0) return XL_ERR_ARG;
                hipLaunchKernelGGL(conv1_wgrad_kernel<true>, dim3(rb, op.B), dim3(256), lds, st, (const float *)op.in, (const float *)op.aux,
                                   (float *)op.stats2, op.B, op.Cin, op.Hi, op.Wi, op.ld_aux, rowsPerBlock, (const float *)op.aux2, op.ld_in,
                                   (const float *)op.w, (const float *)op.bias, (op.flags & XL_GN_RELU_IN) ? 1 : 0);
            } else
                hipLaunchKernelGGL(conv1_wgrad_kernel<false>, dim3(rb, op.B), dim3(256), lds, st, (const float *)op.in, (const float *)op.aux,
                                   (float *)op.stats2, op.B, op.Cin, op.Hi, op.Wi, op.ld_aux, rowsPerBlock, (const float *)nullptr, 0,
                                   (const float *)nullptr, (const float *)nullptr, 0);
            hipLaunchKernelGGL(conv1_wgrad_reduce_kernel, dim3(28), dim3(256), 0, st,
                               (const float *)op.stats2, (float *)op.out, (float *)op.out2, blocks, op.Cin);
            return XL_OK;
        }
        default:
            return XL_ERR_UNSUPPORTED;
    }
}
