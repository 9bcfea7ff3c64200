// crossloc_hip: data gradient of the stride-2 3x3 stem convolutions (conv2 32 -> 64, conv3 64 -> 128) of training plans on the
// bf16 matrix pipe with exact three-term operand splits (round 4).
//
//   dX[n][iy][ix][ci] = sum over (ky, kx, co) with (iy + 1 - ky), (ix + 1 - kx) even and in range:
//                       dY[n][(iy + 1 - ky) / 2][(ix + 1 - kx) / 2][co] * W[co][ci][ky][kx]
//
// Until round 3 this ran as four launches of the fp32 implicit-GEMM kernel (one per parity class of the result pixel) with 32
// result channels in a 64-wide tile and K = 64 ... 256: 1.67 ms for conv2 at batch 16 (61 TFLOP/s on a 2 x 51 GFLOP problem
// whose HBM floor - dY read once, dX written once - is 0.2 ms).  Here a workgroup owns a 16 x 32 tile of dX: the 9 x 17 patch of
// dY it needs is split ONCE into bf16 planes in LDS ([pixel][plane][CO] bf16, an odd multiple of 16 bytes per pixel), and every
// wave multiplies one 32-pixel block of EACH parity class (a class = the result pixels with the same (iy & 1, ix & 1): 1, 2, 2 or
// 4 taps reach it) - the four waves of a workgroup carry the same 9 taps' worth of MFMAs.  Activation fragments are ds_read_b128
// at compile-time offsets from the patch (consecutive lanes = consecutive patch pixels), weight fragments come from global
// memory in fragment order (1 KB per load instruction), the structure of the fused stem kernel (csrc/xl_stem_fused.hip).
// Two workgroups per CU; tiles come from a queue.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/crossloc_cnn.h"
#include "../../include/crossloc_dsac.h"   // status codes
#include "xl_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int kTY = 16, kTX = 32;                      // result pixels per tile
constexpr int kPR = kTY / 2 + 1, kPC = kTX / 2 + 1;    // dY patch: 9 x 17

__device__ __forceinline__ float sd_hi(unsigned w) { return __builtin_bit_cast(float, w & 0xffff0000u); }
__device__ __forceinline__ float sd_lo(unsigned w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ void sd_split_pair(f32x2 v, unsigned &w1, unsigned &w2, unsigned &w3)
{
    w1 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    const f32x2 r = v - f32x2{ sd_lo(w1), sd_hi(w1) };
    w2 = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
    const f32x2 r2 = r - f32x2{ sd_lo(w2), sd_hi(w2) };
    w3 = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2));
}

struct S2DgradArgs {
    const float *dy;             // [B][Ho][Wo][CO], pixel stride ldDy
    const u32x4 *wf;             // weight fragments [9 taps][CO/16][3 planes][CI/32][64 lanes] x 16 B (networks._Plan.s2_dgrad_fragments)
    float *dx;                   // [B][Hi][Wi][CI], pixel stride ldDx
    int *queue;                  // tile queue {next, done}, zero before and after the launch
    int B, Hi, Wi, Ho, Wo, ldDy, ldDx, tilesX, tilesY;
};

template <int CO, int CI>
__global__ __launch_bounds__(256, CO == 64 ? 2 : 1)        // (CO = 128: a 120 KB patch, one workgroup per CU)
void s2_dgrad_kernel(S2DgradArgs a)
{
    constexpr int kPix = CO * 6 + 16;                  // bytes per patch pixel: 3 planes of CO bf16 + 16 (an odd multiple of 16)
    static_assert((kPix / 16) % 2 == 1, "patch pitch must be an odd multiple of 16 bytes");
    constexpr int kPatch = kPR * kPC * kPix;
    constexpr int NJ = CI / 32, NC = CO / 16;          // column blocks of the result; K-steps per tap
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
    unsigned char *sP = dsm;
    int *sQ = reinterpret_cast<int *>(dsm + kPatch);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, kh = lane >> 5;
    const int tilesPerImage = a.tilesX * a.tilesY;
    const int total = a.B * tilesPerImage;

    const __amdgpu_buffer_rsrc_t srdW = __builtin_amdgcn_make_buffer_rsrc((void *)a.wf, 0, 9 * NC * 3 * NJ * 1024, 0x00020000);
    if (tid == 0) { sQ[0] = atomicAdd(a.queue, 1); sQ[1] = atomicAdd(a.queue, 1); }
    __syncthreads();
    int t = sQ[0], tNext = sQ[1];
    __syncthreads();

    // my pixel inside a 32-pixel block of a class: (a, b) = (2 wave + (fr >> 4), fr & 15) -> result pixel (2 a + py, 2 b + px)
    const int pa = 2 * wave + (fr >> 4), pb = fr & 15;
    const unsigned aBase = (unsigned)((pa * kPC + pb) * kPix + kh * 16);

    // the dY patch of a tile: 16-byte pieces (4 channels) of the 9 x 17 x CO block, NP per thread, fetched ONE TILE AHEAD into
    // registers (all loads of a tile in flight together; as a load - split - store loop the ten dependent round trips to HBM
    // took longer than the tile's MFMAs)
    constexpr int PIECES = kPR * kPC * (CO / 4), NP = (PIECES + 255) / 256;
    f32x4 pre[NP];
    auto prefetch = [&](int tq) {
        const int n = tq / tilesPerImage, tt = tq - n * tilesPerImage;
        const int ty = tt / a.tilesX, tx = tt - ty * a.tilesX;
        const int oy0 = (kTY * ty) >> 1, ox0 = (kTX * tx) >> 1;
#pragma unroll
        for (int e = 0; e < NP; ++e) {
            const int i = tid + 256 * e;
            const int pix = i / (CO / 4), c4 = i - pix * (CO / 4);
            const int r = pix / kPC, c = pix - r * kPC;
            const int oy = oy0 + r, ox = ox0 + c;
            const bool inb = (tq < total) & (i < PIECES) & (oy < a.Ho) & (ox < a.Wo);
            pre[e] = f32x4{ 0.f, 0.f, 0.f, 0.f };
            if (inb) pre[e] = *reinterpret_cast<const f32x4 *>(a.dy + (((long long)n * a.Ho + oy) * a.Wo + ox) * a.ldDy + 4 * c4);
        }
    };
    prefetch(t);
    while (t < total) {
        if (tid == 0) sQ[0] = atomicAdd(a.queue, 1);
        const int n = t / tilesPerImage, tt = t - n * tilesPerImage;
        const int ty = tt / a.tilesX, tx = tt - ty * a.tilesX;
        const int iy0 = kTY * ty, ix0 = kTX * tx;

        // ---- 1. the dY patch -> LDS, split once
#pragma unroll
        for (int e = 0; e < NP; ++e) {
            const int i = tid + 256 * e;
            if (i < PIECES) {
                const int pix = i / (CO / 4), c4 = i - pix * (CO / 4);
                unsigned wa[3], wb[3];
                sd_split_pair(f32x2{ pre[e][0], pre[e][1] }, wa[0], wa[1], wa[2]);
                sd_split_pair(f32x2{ pre[e][2], pre[e][3] }, wb[0], wb[1], wb[2]);
#pragma unroll
                for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x2 *>(sP + pix * kPix + p * (CO * 2) + c4 * 8) = u32x2{ wa[p], wb[p] };
            }
        }
        __syncthreads();
        const int tAfter = sQ[0];
        prefetch(tNext);                                               // (in flight under this tile's MFMAs)

        // ---- 2. one 32-pixel block of each parity class per wave.  The K-steps of a class - (valid tap, 16-channel chunk) pairs -
        // form a compile-time list; the weight fragments of step s + D are fetched behind the MFMAs of step s (a ring of D
        // register sets: with everything unrolled the compiler would otherwise hoist every load of the tile and spill)
#pragma unroll
        for (int cls = 0; cls < 4; ++cls) {
            const int py = cls >> 1, px = cls & 1;
            const int nky = py ? 2 : 1, nkx = px ? 2 : 1;
            const int nsteps = nky * nkx * NC;
            f32x16 acc[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
            constexpr int D = NJ == 1 ? 4 : 3;             // K-steps of weight fragments in flight (a step is 6 NJ MFMAs = 192 NJ cycles; an L2 hit ~800)
            u32x4 fbr[D][3][NJ];
            auto tap_ky = [&](int s) { const int ti = s / NC / nkx; return py ? 2 * ti : 1; };     // valid ky of a class: py 0 -> {1}; 1 -> {0, 2}
            auto tap_kx = [&](int s) { const int ti = (s / NC) % nkx; return px ? 2 * ti : 1; };
            auto load_b = [&](int s, int slot) {
                const int tap = 3 * tap_ky(s) + tap_kx(s), c = s % NC;
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)       // (a buffer load: ONE lane-offset register + a scalar offset per fragment - with
                                                       //  flat pointers the compiler keeps a 64-bit address per fragment alive and spills)
                        fbr[slot][p][j] = __builtin_amdgcn_raw_buffer_load_b128(srdW, (int)(lane * 16), (int)((((tap * NC + c) * 3 + p) * NJ + j) * 1024), 0);
            };
#pragma unroll
            for (int s = 0; s < D; ++s)
                if (s < nsteps) load_b(s, s);
#pragma unroll
            for (int s = 0; s < 4 * NC; ++s) {
                if (s >= nsteps) break;
                const int ky = tap_ky(s), kx = tap_kx(s), c = s % NC;
                // source pixel of result (2 a + py, 2 b + px): patch (a + dr, b + dc)
                const int dr = (py + 1 - ky) >> 1, dc = (px + 1 - kx) >> 1;
                bf16x8 fa[3];
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    fa[p] = *reinterpret_cast<const bf16x8 *>(sP + aBase + (dr * kPC + dc) * kPix + p * (CO * 2) + c * 32);
                constexpr int PU[6] = { 2, 1, 0, 1, 0, 0 }, PV[6] = { 0, 1, 2, 0, 1, 0 };   // (weights, activations), smallest first
#pragma unroll
                for (int tm = 0; tm < 6; ++tm)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fbr[s % D][PU[tm]][j]), fa[PV[tm]], acc[j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (s + D < nsteps) load_b(s + D, s % D);
                __builtin_amdgcn_sched_barrier(0);
            }
            // store: result pixel (iy0 + 2 pa + py, ix0 + 2 pb + px), channels 32 j + 8 q + 4 kh + {0..3}
            const int iy = iy0 + 2 * pa + py, ix = ix0 + 2 * pb + px;
            if (iy < a.Hi && ix < a.Wi) {
                float *o = a.dx + (((long long)n * a.Hi + iy) * a.Wi + ix) * a.ldDx + 4 * kh;
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<f32x4 *>(o + 32 * j + 8 * q) = f32x4{ acc[j][4 * q], acc[j][4 * q + 1], acc[j][4 * q + 2], acc[j][4 * q + 3] };
            }
        }
        __syncthreads();                                               // every wave is done with the patch
        t = tNext; tNext = tAfter;
    }
    if (tid == 0) {
        __threadfence();
        if (atomicAdd(a.queue + 1, 1) == (int)gridDim.x - 1) { a.queue[0] = 0; a.queue[1] = 0; __threadfence(); }
    }
}

template <int CO, int CI>
int launch_s2_dgrad(S2DgradArgs a, hipStream_t st)
{
    const size_t lds = (size_t)kPR * kPC * (CO * 6 + 16) + 64;
    static XlLdsLimit configured;
    int cfgDev;
    if (configured.needs(lds, &cfgDev)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(s2_dgrad_kernel<CO, CI>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return XL_ERR_HIP;
        configured.done(lds, cfgDev);
    }
    const long long total = (long long)a.B * a.tilesX * a.tilesY;
    int grid = lds * 2 <= 160 * 1024 ? 512 : 256;                      // two workgroups per CU where the patch allows
    if (grid > total) grid = (int)total;
    hipLaunchKernelGGL((s2_dgrad_kernel<CO, CI>), dim3(grid), dim3(256), lds, st, a);
    return XL_OK;
}

}  // namespace

// XL_OP_S2_DGRAD: data gradient of a 3x3 stride-2 pad-1 convolution.  in = dY [B,Hi,Wi,Cin] NHWC (ld_in; Cin = the forward
// layer's OUTPUT channels: 64 or 128), w = weight fragments (networks._Plan.s2_dgrad_fragments), out = dX [B,Ho,Wo,Cout] (ld_out;
// Cout = the forward layer's input channels: 32 or 64; Hi = (Ho - 1) / 2 + 1), stats = two int32, zero (tile queue).  Overwrites dX.
int xl_run_s2_dgrad(const xl_op &op, hipStream_t st)
{
    if (!((op.Cin == 64 && op.Cout == 32) || (op.Cin == 128 && op.Cout == 64)) || op.Hi != (op.Ho - 1) / 2 + 1 || op.Wi != (op.Wo - 1) / 2 + 1 ||
        op.ld_in < op.Cin || op.ld_out < op.Cout || (op.ld_in & 3) || (op.ld_out & 3) || !op.in || !op.w || !op.out || !op.stats ||
        (((uintptr_t)op.in | (uintptr_t)op.out | (uintptr_t)op.w) & 15) || op.B < 1)
        return XL_ERR_ARG;
    S2DgradArgs a;
    a.dy = (const float *)op.in; a.wf = (const u32x4 *)op.w; a.dx = (float *)op.out; a.queue = (int *)op.stats;
    a.B = op.B; a.Hi = op.Ho; a.Wi = op.Wo; a.Ho = op.Hi; a.Wo = op.Wi; a.ldDy = op.ld_in; a.ldDx = op.ld_out;
    a.tilesX = (op.Wo + kTX - 1) / kTX; a.tilesY = (op.Ho + kTY - 1) / kTY;
    if ((long long)a.B * a.tilesX * a.tilesY >= 0x7fffffffLL) return XL_ERR_ARG;
    return op.Cin == 64 ? launch_s2_dgrad<64, 32>(a, st) : launch_s2_dgrad<128, 64>(a, st);
}
