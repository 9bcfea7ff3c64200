// xl_cnn.hip — MI355X (gfx950) kernels for CrossLoc's scene-coordinate CNN forward.
//
// What PyTorch dispatched for TransPoseNet.forward (/root/reference/networks/networks.py:466-502) is
// re-designed here as five kernels executed from an op list (include/crossloc_cnn.h):
//
//   conv1_direct   3x3 s1 conv on the 3-channel NCHW image -> NHWC (HBM-bound: 44 MB out per 480x720 image)
//   igemm_conv     every other conv as an implicit GEMM on exact-fp32 MFMA (v_mfma_f32_32x32x2_f32):
//                  M = B*Ho*Wo output pixels, N = Cout, K = k*k*Cin; 128xBN block tile, BK = 32,
//                  A (im2col gather, zero-filled padding) and B (weights [Cout][K]) staged through LDS with
//                  a 36-float row pitch (conflict-free ds_read_b128), register-prefetched double buffer,
//                  one barrier per K-step; 4 wavefronts as 2x2, each 64 x BN/2 of the tile (2 x BN/64 MFMA
//                  tiles of 32x32); XCD-aware tile order so the n-tiles of one m-tile share an L2; K runs
//                  chunk-major/tap-minor so the 9 shifted reads of an input chunk are cache hits.
//   gn_stats       GroupNorm statistics, coalesced: a workgroup reads a pixel chunk of all channels, fp64
//                  per-thread partials, fixed-order LDS combine -> per-(image, chunk, group) (sum, sumsq)
//   gn_apply       finalises mean/rstd from the chunk partials (fixed order) into per-channel scale/shift
//                  in LDS, then streams y = x*scale + shift with fused ReLU / residual add / ReLU
//   head           fc3 (512 -> 4) + mean offset + exp(hardtanh), one wavefront per pixel, NCHW out
//
// fp32 end to end: the MFMA used is bitwise an fmaf chain, so parity with the fp32 reference is at
// summation-order level (tests compare against torch fp32 on CPU and the golden vectors).
#include <hip/hip_runtime.h>
#include <array>
#include <map>
#include <vector>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

#include "../../include/crossloc_cnn.h"
#include "../../include/crossloc_dsac.h"   // status codes
#include "xl_common.h"

int xl_run_bwd_op(const xl_op &op, hipStream_t st);   // xl_cnn_bwd.hip
int xl_run_split_gemm(const xl_op &op, hipStream_t st);   // xl_gemm_split.hip
int xl_run_split_stem(const xl_op &op, hipStream_t st);   // xl_stem_split.hip
int xl_run_pair_stem(const xl_op &op, hipStream_t st);    // xl_stem_pair.hip
int xl_run_stem12(const xl_op &op, hipStream_t st);       // xl_stem_fused.hip
int xl_run_s2_dgrad(const xl_op &op, hipStream_t st);     // xl_stem_dgrad.hip

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------------------- conv1

// in NCHW [B,Cin,H,W]; w [(ky*3+kx)*Cin + c][Cout]; out NHWC.  Thread = (pixel, 8 output channels).
__global__ __launch_bounds__(256)
void conv1_direct_kernel(const float *__restrict__ in, const float *__restrict__ w, const float *__restrict__ bias,
                         float *__restrict__ out, int B, int Cin, int H, int W, int Cout, int ldOut)
{
    extern __shared__ __attribute__((aligned(16))) float sW[];      // [9*Cin][Cout] + bias[Cout]
    const int nW = 9 * Cin * Cout;
    for (int i = threadIdx.x; i < nW; i += 256) sW[i] = w[i];
    for (int i = threadIdx.x; i < Cout; i += 256) sW[nW + i] = bias[i];
    __syncthreads();
    const int tpp = Cout >> 3;                                       // threads per pixel
    const int pixPerBlock = 256 / tpp;
    const long long HW = (long long)H * W;
    const long long total = (long long)B * HW;
    const int cg = threadIdx.x % tpp;
    for (long long p = (long long)blockIdx.x * pixPerBlock + threadIdx.x / tpp; p < total;
         p += (long long)gridDim.x * pixPerBlock) {
        const int n = (int)(p / HW);
        const int rem = (int)(p - (long long)n * HW);
        const int y = rem / W, x = rem - y * W;
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = sW[nW + cg * 8 + j];
        const float *img = in + (long long)n * Cin * HW;
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = y + ky - 1;
            if ((unsigned)iy >= (unsigned)H) continue;
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = x + kx - 1;
                if ((unsigned)ix >= (unsigned)W) continue;
                for (int c = 0; c < Cin; ++c) {
                    const float v = img[(long long)c * HW + (long long)iy * W + ix];
                    const float *wr = sW + ((ky * 3 + kx) * Cin + c) * Cout + cg * 8;
                    const f32x4 w0 = *reinterpret_cast<const f32x4 *>(wr);
                    const f32x4 w1 = *reinterpret_cast<const f32x4 *>(wr + 4);
                    acc[0] = fmaf(v, w0.x, acc[0]); acc[1] = fmaf(v, w0.y, acc[1]);
                    acc[2] = fmaf(v, w0.z, acc[2]); acc[3] = fmaf(v, w0.w, acc[3]);
                    acc[4] = fmaf(v, w1.x, acc[4]); acc[5] = fmaf(v, w1.y, acc[5]);
                    acc[6] = fmaf(v, w1.z, acc[6]); acc[7] = fmaf(v, w1.w, acc[7]);
                }
            }
        }
        float *o = out + p * ldOut + cg * 8;
        *reinterpret_cast<f32x4 *>(o) = f32x4{ acc[0], acc[1], acc[2], acc[3] };
        *reinterpret_cast<f32x4 *>(o + 4) = f32x4{ acc[4], acc[5], acc[6], acc[7] };
    }
}

// Inference form of the first layer.  conv1 costs 27 MACs per output while its 32-channel full-resolution output is the
// largest tensor of the network: evaluating it twice is cheaper than writing it raw, re-reading it for the GroupNorm
// statistics and re-reading / re-writing it for the apply.  PASS 0 evaluates the convolution and keeps only the
// per-(image, channel) partial sums (32 groups of 1 channel); PASS 1 evaluates it again and writes
// relu(conv * scale + shift) once.  Thread = pixel x all 32 channels: the weights are wave-uniform, so they stream
// through scalar loads and feed v_fmac as SGPR operands (no LDS traffic, no per-lane weight registers).
// in NCHW [B,3,H,W]; w [(ky*3+kx)*3 + c][32]; out NHWC; grid (chunks, B), a workgroup covers ppt*256 pixels of one image.
constexpr int kC1Pitch = 36;                       // LDS row pitch (floats) of the output transpose: 32 channels + 4 pad
template <int PASS>
__global__ __launch_bounds__(256)
void conv1_fused_kernel(const float *__restrict__ in, const float *__restrict__ w, const float *__restrict__ bias,
                        const float *__restrict__ coeff, float *__restrict__ out, double *__restrict__ stats,
                        int H, int W, int ldOut, int ppt, int relu)
{
    constexpr int CO = 32, CI = 3, NT = 9 * CI;
    __shared__ __attribute__((aligned(16))) float sRed[PASS == 0 ? CO * 256 : 256 * kC1Pitch];
    const int n = blockIdx.y, tid = threadIdx.x;
    const int HW = H * W;
    const float *img = in + (long long)n * CI * HW;
    float s[CO], q[CO];
#pragma unroll
    for (int j = 0; j < CO; ++j) { s[j] = 0.f; q[j] = 0.f; }
    for (int k = 0; k < ppt; ++k) {
        const int p = (blockIdx.x * ppt + k) * 256 + tid;
        const bool live = p < HW;
        const int pc = live ? p : HW - 1;
        const int y = pc / W, x = pc - y * W;
        float v[NT];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int iy = y + ky - 1, ix = x + kx - 1;
                const bool inb = ((unsigned)iy < (unsigned)H) & ((unsigned)ix < (unsigned)W);
                const int off = min(max(iy, 0), H - 1) * W + min(max(ix, 0), W - 1);
#pragma unroll
                for (int c = 0; c < CI; ++c) {
                    const float t = img[(long long)c * HW + off];
                    v[(ky * 3 + kx) * CI + c] = inb ? t : 0.f;          // a zero tap leaves the fma chain unchanged
                }
            }
        }
        // the weight offset is laundered per pixel: otherwise all 864 scalar loads are hoisted out of the loop and
        // spilled into VGPR lanes (one v_readlane per weight per pixel - more VALU work than the convolution itself)
        int zero = 0;
        asm volatile("" : "+s"(zero));
        const float *wk = w + zero;
        float acc[CO];
#pragma unroll
        for (int j = 0; j < CO; ++j) acc[j] = bias[j];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int j = 0; j < CO; ++j) acc[j] = fmaf(v[t], wk[t * CO + j], acc[j]);
        }
        if (PASS == 0) {
#pragma unroll
            for (int j = 0; j < CO; ++j) {
                const float a = live ? acc[j] : 0.f;
                s[j] += a;
                q[j] = fmaf(a, a, q[j]);
            }
        } else {
            // a wave owns 64 consecutive pixels = one contiguous 8 KB span of the NHWC output (ldOut == 32): transpose
            // through LDS so that every store instruction writes 1 KB of consecutive addresses instead of 64 scattered
            // 16-byte pieces
            const float *cf = coeff + (long long)n * CO * 2;
            const int lane = tid & 63;
            float *row = sRed + (tid >> 6) * (64 * kC1Pitch);
#pragma unroll
            for (int j = 0; j < CO; j += 4) {
                f32x4 r;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = acc[j + e] * cf[(j + e) * 2] + cf[(j + e) * 2 + 1];
                    if (relu) t = fmaxf(t, 0.f);
                    r[e] = t;
                }
                *reinterpret_cast<f32x4 *>(row + lane * kC1Pitch + j) = r;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int p0 = (blockIdx.x * ppt + k) * 256 + (tid & ~63);     // first pixel of the wave
            float *o = out + ((long long)n * HW + p0) * CO;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int qd = c * 64 + lane;                               // 16-byte piece of the span
                const int px = qd >> 3, part = qd & 7;
                const f32x4 r = *reinterpret_cast<const f32x4 *>(row + px * kC1Pitch + part * 4);
                if (p0 + px < HW) *reinterpret_cast<f32x4 *>(o + (long long)qd * 4) = r;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (PASS == 0) {
        // per-thread fp32 partials -> fp64 across the workgroup: value-major transpose through LDS, 4 threads per value
        const int vsel = tid >> 2, part = tid & 3;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            __syncthreads();
#pragma unroll
            for (int j = 0; j < CO; ++j) sRed[j * 256 + tid] = half ? q[j] : s[j];
            __syncthreads();
            if (vsel < CO) {
                double a = 0.0;
                for (int i = 0; i < 64; ++i) a += (double)sRed[vsel * 256 + part * 64 + ((i + vsel + 16 * part) & 63)];
                a += __shfl_xor(a, 1);
                a += __shfl_xor(a, 2);
                if (part == 0) stats[(((long long)n * gridDim.x + blockIdx.x) * CO + vsel) * 2 + half] = a;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------- conv1 on the matrix pipe
//
// The same two evaluations (statistics only / normalise + ReLU + write) with the multiplies on the bf16 matrix pipe instead
// of 432 packed VALU FMAs per pixel: every fp32 value - image and weights - is an exact sum of three bf16 terms, six term
// pairs per product, fp32 accumulation (as csrc/xl_gemm_split.hip).  A workgroup owns 16 x 64 output pixels: the 18 x 66
// halo of the image is split ONCE while it is staged into LDS (each value is used by nine patches), one 8-byte word
// {R, G, B, 0} per pixel and plane.  The K dimension is laid out so that a K-step of v_mfma_f32_32x32x16_bf16 is one row of
// the 3 x 3 window: 16 slots = 4 pixels (x-1, x, x+1 and a fourth with zero weights) x {R, G, B, 0}; a lane's 8 values are
// two neighbouring pixel words = one ds_read2_b64, no gather.  32 pixels of a row x 32 channels = 3 K-steps x 6 term
// pairs = 18 MFMAs; the weight fragments (9 x 4 registers) are built once per wave.
typedef __bf16 c1_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int c1_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int c1_u32x2 __attribute__((ext_vector_type(2)));
constexpr int kC1TH = 16, kC1TW = 64, kC1HH = kC1TH + 2, kC1HW = kC1TW + 4;   // 66 halo columns + 2 of zeros (slot dx = 3 of the last pixels)
constexpr int kC1Halo = 3 * kC1HH * kC1HW * 8;                 // bytes: [plane][row][col] of 8-byte pixel words

__device__ __forceinline__ unsigned c1_bf16_rn(float x)
{
    unsigned u = __builtin_bit_cast(unsigned, x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ void c1_split3(float a, unsigned &h1, unsigned &h2, unsigned &h3)
{
    h1 = c1_bf16_rn(a);
    const float r1 = a - __builtin_bit_cast(float, h1 << 16);
    h2 = c1_bf16_rn(r1);
    h3 = c1_bf16_rn(r1 - __builtin_bit_cast(float, h2 << 16));
}

template <int PASS>
__global__ __launch_bounds__(256)
void conv1_mfma_kernel(const float *__restrict__ in, const c1_u32x4 *__restrict__ wfrag, const float *__restrict__ bias,
                       const float *__restrict__ coeff, float *__restrict__ out, double *__restrict__ stats,
                       int H, int W, int tilesX, int relu)
{
    constexpr int CO = 32;
    // PASS 0: statistics only; 1: conv * scale + shift (+ReLU) written; 2 (round 3): the RAW convolution written AND its
    // statistics, one evaluation - the consumer (conv2 on the split pipe) applies the GroupNorm while it loads its operand
    constexpr bool kStats = PASS != 1, kWrite = PASS != 0;
    constexpr int kSmW = kC1Halo + 4 * 32 * kC1Pitch * 4, kSmS = kC1Halo > 256 * 32 * 4 ? kC1Halo : 256 * 32 * 4;
    constexpr int kSm = !kWrite ? kSmS : (kStats && kSmS > kSmW ? kSmS : kSmW);
    __shared__ __attribute__((aligned(16))) unsigned char smem[kSm];
    const int n = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int ty = blockIdx.x / tilesX, tx = blockIdx.x - ty * tilesX;
    const int y0 = ty * kC1TH, x0 = tx * kC1TW;
    const long long HW = (long long)H * W;
    const float *img = in + (long long)n * 3 * HW;

    // ---- halo -> LDS, split
    c1_u32x2 *sH = reinterpret_cast<c1_u32x2 *>(smem);
    for (int i = tid; i < kC1HH * kC1HW; i += 256) {
        const int r = i / kC1HW, c = i - r * kC1HW;
        const int y = y0 - 1 + r, x = x0 - 1 + c;
        const bool inb = ((unsigned)y < (unsigned)H) & ((unsigned)x < (unsigned)W) & (c < kC1TW + 2);
        const long long off = (long long)(inb ? y : 0) * W + (inb ? x : 0);
        unsigned h[3][3];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const float v = inb ? img[ch * HW + off] : 0.f;
            c1_split3(v, h[0][ch], h[1][ch], h[2][ch]);
        }
#pragma unroll
        for (int p = 0; p < 3; ++p) sH[p * (kC1HH * kC1HW) + i] = c1_u32x2{ h[p][0] | (h[p][1] << 16), h[p][2] };
    }
    // ---- weight fragments: lane -> (channel lane & 31, K half lane >> 5); slot j of a K-step = (dx = j >> 2, c = j & 3),
    // zero for c = 3 and dx = 3.  Split and packed once per plan on the host side: wfrag[plane][dy][lane], 16 bytes each.
    const int kh = lane >> 5;
    c1_bf16x8 wf[3][3];                                              // [plane][dy]
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) wf[p][dy] = __builtin_bit_cast(c1_bf16x8, wfrag[(p * 3 + dy) * 64 + lane]);
    // accumulator element r of a lane: pixel lane & 31, channel 8 (r >> 2) + 4 kh + (r & 3)
    float b16[16], sc[16], sh[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int c = 8 * (r >> 2) + 4 * kh + (r & 3);
        b16[r] = bias[c];
        if (PASS == 1) { sc[r] = coeff[((long long)n * CO + c) * 2]; sh[r] = coeff[((long long)n * CO + c) * 2 + 1]; }
    }
    float s[16], q[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; q[r] = 0.f; }
    __syncthreads();

    const int px = lane & 31;
#pragma unroll 1
    for (int blk = 0; blk < 8; ++blk) {
        const int ly = wv * 4 + (blk >> 1), lx = (blk & 1) * 32 + px;   // pixel of this lane inside the tile
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = b16[r];
        c1_bf16x8 pf[3][3];                                          // [plane][dy]
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const c1_u32x2 *src = sH + p * (kC1HH * kC1HW) + (ly + dy) * kC1HW + lx + 2 * kh;
                const c1_u32x2 a = src[0], b = src[1];
                pf[p][dy] = __builtin_bit_cast(c1_bf16x8, c1_u32x4{ a[0], a[1], b[0], b[1] });
            }
        constexpr int PW[6] = { 2, 1, 0, 1, 0, 0 }, PP[6] = { 0, 1, 2, 0, 1, 0 };   // smallest terms first
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[PW[t]][dy], pf[PP[t]][dy], acc, 0, 0, 0);
        const int y = y0 + ly, xb = x0 + (blk & 1) * 32;               // first pixel of the block
        if (kStats) {
            const bool live = (y < H) & (xb + px < W);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float a = live ? acc[r] : 0.f;
                s[r] += a;
                q[r] = fmaf(a, a, q[r]);
            }
        }
        if (kWrite) {
            // the block is one contiguous 4 KB span of the NHWC output: transposed through a wave-private LDS area so that
            // every store instruction writes 1 KB of consecutive addresses
            float *row = reinterpret_cast<float *>(smem + kC1Halo) + wv * (32 * kC1Pitch);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = acc[4 * g + e];
                    if (PASS == 1) {
                        t = t * sc[4 * g + e] + sh[4 * g + e];
                        if (relu) t = fmaxf(t, 0.f);
                    }
                    v[e] = t;
                }
                *reinterpret_cast<f32x4 *>(row + px * kC1Pitch + 8 * g + 4 * kh) = v;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (y < H) {
                float *o = out + (((long long)n * H + y) * W + xb) * CO;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int qd = c * 64 + lane;                       // 16-byte piece of the span
                    const int pp = qd >> 3, part = qd & 7;
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(row + pp * kC1Pitch + part * 4);
                    if (xb + pp < W) *reinterpret_cast<f32x4 *>(o + (long long)qd * 4) = v;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (kStats) {
        // per-lane fp32 partials (8 pixels each) -> fp64 over the 128 lanes that hold a channel, fixed order.
        // Staging layout [value][thread] (round 6): with [thread][32 values] every lane of a ds_write_b32 hit the same bank - 32 writes
        // of 64 LDS cycles each per wave, four times the LDS time of the whole convolution loop, and the kernel was bound by it
        // (0.40 of its LDS cycles were conflict cycles).  The readers start at different lanes (l + c) so that they, too, spread
        // over the banks; an fp64 sum of 128 fp32 values of one sign-mixed magnitude range is exact, the order is fixed anyway.
        float *sRed = reinterpret_cast<float *>(smem);
        __syncthreads();                                             // every wave is done with the halo
#pragma unroll
        for (int r = 0; r < 16; ++r) { sRed[r * 256 + tid] = s[r]; sRed[(16 + r) * 256 + tid] = q[r]; }
        __syncthreads();
        if (tid < 64) {
            const int c = tid & 31, half = tid >> 5;
            const int khc = (c >> 2) & 1, r = ((c >> 3) << 2) | (c & 3);
            const float *src = sRed + (half * 16 + r) * 256 + khc * 32;
            double a = 0.0;
            for (int wq = 0; wq < 4; ++wq)
                for (int l = 0; l < 32; ++l) a += (double)src[wq * 64 + ((l + c) & 31)];
            stats[(((long long)n * gridDim.x + blockIdx.x) * CO + c) * 2 + half] = a;
        }
    }
}

// ---------------------------------------------------------------------------------------------- igemm conv

constexpr int kWaitVm0 = 0x0F70;                   // s_waitcnt vmcnt(0) (expcnt / lgkmcnt fields left at their maxima)
constexpr int kBK = 32;                            // K-step; one LDS tile row = 32 floats (128 B)

struct ConvArgs {
    const float *in; const float *w; const float *bias; float *out;
    int B, Hi, Wi, Cin, Ho, Wo, Cout, ldIn, ldOut;
    int M, K, nbm, nbn;
    unsigned inBytes, wBytes, outBytes;   // extents for the buffer descriptors (hardware bounds check)
    int accumulate;                 // epilogue: out += result (XL_CONV_ACCUMULATE)
    // fused GroupNorm statistics of the OUTPUT (forward only): fp64 partial sums per (image, tile-within-image, group)
    double *stats; int HW, G, cpg, nchunks;
    // MODE 2 (stride-2 data gradient, one parity class of result pixels per launch)
    int py, px, Hj, Wj, ntaps; unsigned tapList;
    // batched launch (Winograd: 16 independent GEMMs): tile t of the grid belongs to GEMM z = t / (nbm*nbn)
    int zCount; long long zIn, zW, zOut;   // element strides between consecutive GEMMs
    long long *clk;                 // diagnostics (XL_CONV_CLK=1): per-workgroup shader-clock phase timings, else NULL
    // NORM launches: the producer's GroupNorm is applied to the A operand on its way into LDS ("normalise on load"):
    // coef = {scale, shift} pairs [B][Cin][2] from GN_FINAL, x -> max(x*scale + shift, normLo), normLo = 0 (ReLU) or -inf
    const float *coef; float normLo;
};

// bijective XCD remap: block b runs on XCD b%8; give each XCD a contiguous run of tiles
__device__ __forceinline__ int xcd_remap(int b, int nwg)
{
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = b & 7, local = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
}

// Tiles go global -> LDS directly (buffer_load ... lds through buffer descriptors): no register staging and no
// ds_write phase.  A padding tap gets an out-of-range offset and the hardware writes zeros (no branch, no select on
// the data).  LDS rows are unpadded (32 floats = 128 B = 8 lanes x 16 B, the lane-linear DMA destination); the 16-byte
// slots of a row are XOR-swizzled by (row>>1)&7, applied to the SOURCE offset of the lane that fills a slot and again
// to the fragment reads, which keeps ds_read_b128 conflict-free without padding.
// MODE 0: forward convolution.  MODE 1: data gradient — `in` is dY [B,Hi,Wi,Cin] (the forward OUTPUT, Cin = forward
// Cout), the result is dX [B,Ho,Wo,Cout] (forward input); output pixel (iy,ix) gathers dY[(iy+PAD-ky)/S][(ix+PAD-kx)/S]
// for the taps whose offset is divisible by the forward stride S (others are zero-filled by the bounds check).
// MODE 2: the stride-2 data gradient split by result-pixel parity (py,px): only the 1/2/2/4 taps that can reach a
// pixel of that class are multiplied (9 tap-GEMMs in total over the four launches instead of 36).
// NORM (1x1 forward launches only): the A operand is the RAW output of the producing convolution and its GroupNorm
// (+ReLU) is applied on the way into LDS, so the producer's separate apply pass (one read + one write of the
// activation) does not exist.  The A tile then goes global -> registers -> x*scale + shift -> ds_write instead of by
// DMA (the weights still go by DMA); the per-(image, channel) coefficients of the at most two images a tile touches
// sit in LDS behind the tiles.
template <int KS, int STRIDE, int BN, int CIN, int MODE = 0, int BM = 128, int ZB = 0, int NORM = 0>   // CIN = compile-time Cin tag (0: runtime);
                                         // BM = 64 for launches that would not fill the chip; ZB 1 = batched GEMMs (Winograd)
__global__ __launch_bounds__(256, 2)
void igemm_conv_kernel(ConvArgs a)
{
    static_assert(!NORM || (KS == 1 && STRIDE == 1 && MODE == 0 && ZB == 0), "normalise-on-load exists for 1x1 forward convs");
    constexpr int PAD = KS / 2;
    // The MFMA row operand is the WEIGHT fragment ("swapped"), so an accumulator register quad holds 4 consecutive output
    // channels of one pixel = 16 contiguous bytes of the NHWC result: the epilogue needs 16 dwordx4 stores per wave
    // instead of 64 dword stores.  It matters because the epilogue runs beside the co-resident workgroup's MFMA stream
    // and pays per INSTRUCTION there (XL_CONV_CLK: 27k ticks for the 64-store form with two workgroups per CU, 50k with
    // the per-element statistics on top, 6k for this form); while a workgroup sits in its epilogue its partner cannot
    // keep the matrix pipe full alone, so a long epilogue costs pipe time, not only latency: the 1x1 512->512 layers
    // ran at 78 % with the long one.  Forward launches start the accumulators at the bias instead of adding it per
    // element, and take the GroupNorm statistics from per-lane partial sums (below).
    constexpr int NJ = BN / 64;                 // 32-wide MFMA tiles per wave along N
    constexpr int BROWS = BN / 32;              // B-tile rows loaded per thread
    constexpr int AROWS = BM / 32;              // A-tile rows loaded per thread
    constexpr int TI = BM / 64;                 // 32-high MFMA tiles per wave along M
    constexpr unsigned OOB = 0x80000000u;       // > any legal extent: forces the zero-fill path
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As = smem;                           // [2][BM][32]
    float *Bs = smem + 2 * BM * kBK;            // [2][BN][32]
    if (CIN != 0) a.Cin = CIN, a.K = KS * KS * CIN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // each XCD gets a contiguous run of (GEMM, m-tile, n-tile): the n-tiles of an m-tile share an L2
    int tile = xcd_remap(blockIdx.x, a.nbm * a.nbn * (ZB ? a.zCount : 1));
    if constexpr (ZB) {
        const int z = tile / (a.nbm * a.nbn);
        tile -= z * (a.nbm * a.nbn);
        a.in += z * a.zIn; a.w += z * a.zW; a.out += z * a.zOut;
    }
    const int mt = tile / a.nbn, nt = tile - mt * a.nbn;
    const int m0 = mt * BM, n0 = nt * BN;

    const __amdgpu_buffer_rsrc_t srdA = __builtin_amdgcn_make_buffer_rsrc((void *)a.in, 0, (int)a.inBytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t srdB = __builtin_amdgcn_make_buffer_rsrc((void *)a.w, 0, (int)a.wBytes, 0x00020000);

    // ---- per-thread load coordinates: AROWS A rows (and BROWS B rows); this lane fills physical slot tid&7 of its
    // rows with the logical k-slot kq
    const int lrow = tid >> 3;
    const int kq = (tid & 7) ^ ((lrow >> 1) & 7);
    unsigned aOff[AROWS];                           // byte offset of (n, iy0, ix0, 4*kq); wraps for padding rows
    int aIy[AROWS], aIx[AROWS];
    const int HoWo = (MODE == 2) ? a.Hj * a.Wj : a.Ho * a.Wo;
    const int rowW = (MODE == 2) ? a.Wj : a.Wo;
#pragma unroll
    for (int p = 0; p < AROWS; ++p) {
        const int m = m0 + lrow + 32 * p;
        if (m < a.M) {
            const int n = m / HoWo;
            const int rem = m - n * HoWo;
            const int oy = rem / rowW, ox = rem - oy * rowW;
            if constexpr (MODE == 0) {
                aIy[p] = oy * STRIDE - PAD;
                aIx[p] = ox * STRIDE - PAD;
                aOff[p] = (unsigned)(((n * a.Hi + aIy[p]) * a.Wi + aIx[p]) * a.ldIn + 4 * kq) * 4u;
            } else if constexpr (MODE == 1) {
                aIy[p] = oy + PAD;                   // numerators of the source row / column
                aIx[p] = ox + PAD;
                aOff[p] = (unsigned)(n * a.Hi * a.Wi * a.ldIn + 4 * kq) * 4u;
            } else {
                aIy[p] = 2 * oy + a.py + PAD;
                aIx[p] = 2 * ox + a.px + PAD;
                aOff[p] = (unsigned)(n * a.Hi * a.Wi * a.ldIn + 4 * kq) * 4u;
            }
        } else {
            aIy[p] = -100000; aIx[p] = -100000; aOff[p] = 0;
        }
    }
    unsigned bOff[BROWS];
#pragma unroll
    for (int p = 0; p < BROWS; ++p) bOff[p] = (unsigned)((n0 + lrow + 32 * p) * a.K + 4 * kq) * 4u;

    typedef __attribute__((address_space(3))) void lds_void;
    // Source offsets of K-step kk: A rows (per lane, OOB for padding taps) and the weight K position (scalar)
    auto tile_offsets = [&](int kk, unsigned (&voff)[AROWS], unsigned &kbytes) {
        // K order is (channel chunk of 32, tap, channel-in-chunk): the 9 taps of one chunk run back to back, so
        // the shifted re-reads of the same input pixels hit L1/L2 instead of going back to HBM 9 times
        int kbase = kk * kBK;
        int chunk, tap;
        if constexpr (MODE == 2) {
            chunk = kk / a.ntaps;
            tap = (int)((a.tapList >> (4 * (kk - chunk * a.ntaps))) & 15u);
            kbase = (chunk * (KS * KS) + tap) * kBK;             // position of this (chunk, tap) in the packed weights
        } else {
            chunk = kk / (KS * KS);
            tap = kk - chunk * (KS * KS);
        }
        kbytes = (unsigned)kbase * 4u;
        const int c0 = chunk * kBK;
        const int ky = tap / KS, kx = tap - ky * KS;
        const unsigned tapOff = (unsigned)((ky * a.Wi + kx) * a.ldIn + c0) * 4u;
#pragma unroll
        for (int p = 0; p < AROWS; ++p) {
            if constexpr (MODE == 0) {
                const bool ok = (unsigned)(aIy[p] + ky) < (unsigned)a.Hi && (unsigned)(aIx[p] + kx) < (unsigned)a.Wi;
                voff[p] = ok ? aOff[p] + tapOff : OOB;
            } else {
                const int ty = aIy[p] - ky, tx = aIx[p] - kx;
                const int sy = ty / STRIDE, sx = tx / STRIDE;               // STRIDE is 1 or 2 (shift)
                // (bitwise &: a short-circuit chain becomes nested exec-masked branches)
                const bool ok = ((ty | tx) >= 0) & (STRIDE == 1 || (((ty | tx) & 1) == 0)) & (sy < a.Hi) & (sx < a.Wi);
                // mask arithmetic, not a select: hipcc turns the select into an exec-masked branch around the
                // multiplies, which splits the K-loop into basic blocks that do not overlap with the MFMAs
                const unsigned msk = 0u - (unsigned)ok;
                voff[p] = ((aOff[p] + (unsigned)((sy * a.Wi + sx) * a.ldIn + c0) * 4u) & msk) | (OOB & ~msk);
            }
        }
    };
    // one wave instruction moves 8 rows x 128 B; destination = wave-uniform base (M0) + lane*16
    auto issue_dma_a = [&](const unsigned (&voff)[AROWS], int buf) {
#pragma unroll
        for (int p = 0; p < AROWS; ++p)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srdA, (lds_void *)(As + (buf * BM + 32 * p + 8 * wave) * kBK), 16,
                                                     (int)voff[p], 0, 0, 0);
    };
    auto issue_dma_b = [&](unsigned kbytes, int buf) {
#pragma unroll
        for (int p = 0; p < BROWS; ++p)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srdB, (lds_void *)(Bs + (buf * BN + 32 * p + 8 * wave) * kBK), 16,
                                                     (int)bOff[p], (int)kbytes, 0, 0);
    };
    auto load_dma = [&](int kk, int buf) {
        unsigned voff[AROWS], kbytes;
        tile_offsets(kk, voff, kbytes);
        if constexpr (!NORM) issue_dma_a(voff, buf);
        issue_dma_b(kbytes, buf);
    };
    // ---- NORM: A through registers.  This thread owns the 16-byte slot (row lrow + 32p, physical slot tid&7) of every
    // K-step - the slot the DMA form fills through it - holding channels 4*kq .. 4*kq+3 of the step's 32-channel chunk.
    float *sCoef = smem + 2 * (BM + BN) * kBK;                  // [2 image slots][Cin][2]
    int coefSlot[AROWS];                                        // float offset of this row's image slot in sCoef
    f32x4 aReg[AROWS];
    auto load_a_regs = [&](const unsigned (&voff)[AROWS]) {
#pragma unroll
        for (int p = 0; p < AROWS; ++p)
            aReg[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srdA, (int)voff[p], 0, 0));
    };
    f32x4 cf[AROWS][2];                                         // {s0 t0 s1 t1}, {s2 t2 s3 t3} of the row's image
    auto load_coefs = [&](int kk) {                             // issued a K-step ahead: never waited for in the tail
        const int c = kk * kBK + 4 * kq;                        // first of this thread's 4 channels
#pragma unroll
        for (int p = 0; p < AROWS; ++p) {
            cf[p][0] = *reinterpret_cast<const f32x4 *>(sCoef + coefSlot[p] + 2 * c);
            cf[p][1] = *reinterpret_cast<const f32x4 *>(sCoef + coefSlot[p] + 2 * c + 4);
        }
    };
    auto store_a_regs = [&](int buf, int pLo, int pHi) {        // normalise + ds_write rows [pLo, pHi) into stage `buf`
#pragma unroll
        for (int p = 0; p < AROWS; ++p) {
            if (p < pLo || p >= pHi) continue;
            f32x4 v = aReg[p];
            v[0] = fmaxf(fmaf(v[0], cf[p][0][0], cf[p][0][1]), a.normLo);
            v[1] = fmaxf(fmaf(v[1], cf[p][0][2], cf[p][0][3]), a.normLo);
            v[2] = fmaxf(fmaf(v[2], cf[p][1][0], cf[p][1][1]), a.normLo);
            v[3] = fmaxf(fmaf(v[3], cf[p][1][2], cf[p][1][3]), a.normLo);
            *reinterpret_cast<f32x4 *>(As + (buf * BM + 32 * p + lrow) * kBK + 4 * (tid & 7)) = v;
        }
    };

    // C layout (swapped operands): pixel = tile column lane&31, channel = (r&3) + 8*(r>>2) + 4*(lane>>5) of a 32x32 block
    const int rhalf = (lane >> 5) * 4;
    const int nLane = n0 + wn * (BN / 2) + rhalf;                  // + j*32 + 8*q: first of 4 consecutive channels
    f32x16 acc[TI][NJ];
    if constexpr (MODE == 0 && ZB == 0) {
        // forward: the accumulators start at the bias (a null bias reads as zeros through the bounds check)
        const __amdgpu_buffer_rsrc_t srdBias = __builtin_amdgcn_make_buffer_rsrc((void *)a.bias, 0, a.bias ? a.Cout * 4 : 0, 0x00020000);
        typedef unsigned int u32x4b __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 b4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srdBias, (nLane + j * 32 + 8 * q) * 4, 0, 0));
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][j][4 * q + e] = b4[e];
            }
    } else {
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }

    long long tc0 = 0, tw0 = 0, tc1 = 0, tc2 = 0;
    if (a.clk) { tc0 = clock64(); tw0 = wall_clock64(); }
    const int nk = (MODE == 2) ? a.ntaps * (a.Cin / kBK) : a.K / kBK;
    load_dma(0, 0);
    load_dma(1, 1);                                   // (nk == 1: an unused tile, drained with the others below)
    if constexpr (NORM) {
        // coefficient rows of the two images this tile can touch (HW >= BM) by DMA, the first two A stages through
        // registers; everything is issued before the first wait
        const int nLoN = m0 / a.HW;
        const int splitN = (nLoN + 1) * a.HW - m0;              // first tile row of the second image
#pragma unroll
        for (int p = 0; p < AROWS; ++p) coefSlot[p] = (lrow + 32 * p >= splitN) ? 2 * a.Cin : 0;
        const int n1 = (nLoN + 1 < a.B) ? nLoN + 1 : nLoN;
        const __amdgpu_buffer_rsrc_t srdC = __builtin_amdgcn_make_buffer_rsrc((void *)a.coef, 0, a.B * a.Cin * 8, 0x00020000);
        // a wave instruction moves 1 KB = 128 {scale, shift} pairs; wave w copies pieces w, w+4, ... of 2*Cin/128
        for (int pc = wave; pc < (2 * a.Cin) / 128; pc += 4) {
            const int img = pc / (a.Cin / 128), part = pc - img * (a.Cin / 128);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srdC, (lds_void *)(sCoef + img * 2 * a.Cin + part * 256), 16,
                                                     (int)(((img ? n1 : nLoN) * a.Cin + part * 128) * 8 + lane * 16), 0, 0, 0);
        }
        unsigned v0[AROWS], v1[AROWS], kb;
        f32x4 aReg1[AROWS];
        tile_offsets(0, v0, kb);
        tile_offsets(1, v1, kb);
        load_a_regs(v0);
#pragma unroll
        for (int p = 0; p < AROWS; ++p)
            aReg1[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srdA, (int)v1[p], 0, 0));
        __builtin_amdgcn_s_waitcnt(kWaitVm0);
        __syncthreads();                                        // the table is complete
        load_coefs(0);
        store_a_regs(0, 0, AROWS);
#pragma unroll
        for (int p = 0; p < AROWS; ++p) aReg[p] = aReg1[p];
        load_coefs(1);
        store_a_regs(1, 0, AROWS);
    }
    // An LDS-DMA is ordered for other waves' ds_reads only by the issuing wave's vmcnt wait followed by a barrier;
    // the workgroup fence of __syncthreads() waits for LDS operations (lgkmcnt) only, so the vmcnt(0) is explicit.
    __builtin_amdgcn_s_waitcnt(kWaitVm0);
    __syncthreads();
    if (a.clk) tc1 = clock64();

    // MFMA 32x32x2: lane l multiplies row l&31 at k = l>>5.  One ds_read_b128 per 32-row fragment holds the four
    // k-values 4*slot..4*slot+3 of logical slot 2c + (l>>5) of chunk c; the e-th element feeds the e-th MFMA.
    const int fragRow = lane & 31, khalf = lane >> 5;
    const int swz = (fragRow >> 1) & 7;
    const float *Afrag = As + (wm * (BM / 2) + fragRow) * kBK;
    const float *Bfrag = Bs + (wn * (BN / 2) + fragRow) * kBK;
    int cOff[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) cOff[c] = ((2 * c + khalf) ^ swz) * 4;
    // Fragments ping-pong between two register sets: chunk c+1 is read from LDS while chunk c multiplies, across
    // K-steps too (the last chunk of a step prefetches the first fragments of the next one from the other buffer).
    f32x4 fa[2][TI], fb[2][NJ];
    auto read_frags = [&](int set, int buf, int c) {
#pragma unroll
        for (int i = 0; i < TI; ++i) fa[set][i] = *reinterpret_cast<const f32x4 *>(Afrag + (buf * BM + i * 32) * kBK + cOff[c]);
#pragma unroll
        for (int j = 0; j < NJ; ++j) fb[set][j] = *reinterpret_cast<const f32x4 *>(Bfrag + (buf * BN + j * 32) * kBK + cOff[c]);
    };
    auto multiply_e = [&](int set, int e) {
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[set][j][e], fa[set][i][e], acc[i][j], 0, 0, 0);
    };
    auto multiply = [&](int set) {
#pragma unroll
        for (int e = 0; e < 4; ++e) multiply_e(set, e);
    };
    read_frags(0, 0, 0);
    // One barrier per K-step, placed before its LAST chunk: by then every wave has issued all its reads of the
    // current buffer (so the DMA of step kk+2 may overwrite it) and, after the vmcnt(0) of the fence, the tile of
    // step kk+1 has landed in the other buffer (so the cross-step prefetch may read it).  Each DMA has a full
    // K-step (4096 MFMA cycles) to land.  Within a chunk the order is pinned to [first MFMA] [fragment reads of the
    // next chunk] [remaining MFMAs]: the reads start early in the shadow of the MFMA stream and are the only LDS
    // operations outstanding when the next chunk needs them.  The loop body is branch-free (a branch would split
    // the scheduling region): the two DMAs past the last K-step fetch out-of-range / unused data into a buffer
    // nobody reads, and are drained before the epilogue reuses the LDS.
    constexpr int NM = 4 * TI * NJ, ND = TI + NJ, NV = AROWS + BROWS;
    for (int kk = 0; kk < nk; ++kk) {
        const int buf = kk & 1;
        unsigned voffN[AROWS], kbytesN;                // offsets of step kk+2, computed under the MFMAs of this one
        tile_offsets(kk + 2, voffN, kbytesN);
        if constexpr (NORM) {
            // A of step kk+2 and its coefficients: issued first (in the shadow of the MFMAs still executing), in flight
            // for the whole K-step (past the last step: reads of unused data inside the allocations)
            load_a_regs(voffN);
            load_coefs(kk + 2);
            __builtin_amdgcn_sched_barrier(0);
        }
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
        __builtin_amdgcn_sched_barrier(0);            // nothing moves across: the MFMAs are not ordered by the fence
        __builtin_amdgcn_s_waitcnt(kWaitVm0);         // tile kk+1 (issued one K-step ago) has landed
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        // last chunk: MFMAs first, then the DMA issue and the cross-step fragment prefetch in their shadow
        // (ALU may move across the fences, MFMA / LDS / VMEM instructions may not)
        multiply_e(1, 0);
        __builtin_amdgcn_sched_barrier(0x6);
        if constexpr (NORM) store_a_regs(buf, 0, AROWS / 2);
        else issue_dma_a(voffN, buf);
        __builtin_amdgcn_sched_barrier(0x6);
        multiply_e(1, 1);
        __builtin_amdgcn_sched_barrier(0x6);
        issue_dma_b(kbytesN, buf);
        read_frags(0, buf ^ 1, 0);
        __builtin_amdgcn_sched_barrier(0x6);
        multiply_e(1, 2);
        if constexpr (NORM) {
            __builtin_amdgcn_sched_barrier(0x6);
            store_a_regs(buf, AROWS / 2, AROWS);
            __builtin_amdgcn_sched_barrier(0x6);
        }
        multiply_e(1, 3);
    }
    __builtin_amdgcn_s_waitcnt(kWaitVm0);             // drain the trailing DMAs before the epilogue reuses the LDS
    __syncthreads();

    if (a.clk) tc2 = clock64();
    // ---- epilogue.  Branch-free stores: out-of-range rows / columns get an out-of-range buffer offset (stores dropped,
    // loads return 0); with branches the compiler must assume a load pending at every block entry and emits vmcnt(0)
    // before each store.  The accumulate switch is a compile-time tag for the same reason.
    const __amdgpu_buffer_rsrc_t srdO = __builtin_amdgcn_make_buffer_rsrc((void *)a.out, 0, (int)a.outBytes, 0x00020000);
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    unsigned pixOff[TI];
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        const int m = m0 + wm * (BM / 2) + i * 32 + (lane & 31);
        int pix = m;
        if constexpr (MODE == 2) {
            const int nn = m / HoWo;
            const int rem = m - nn * HoWo;
            const int jy = rem / a.Wj, jx = rem - jy * a.Wj;
            pix = (nn * a.Ho + 2 * jy + a.py) * a.Wo + 2 * jx + a.px;
        }
        pixOff[i] = (m < a.M) ? (unsigned)(pix * a.ldOut) * 4u : OOB;
    }
    auto store_swapped = [&](auto accTag) {
        constexpr bool ACC = decltype(accTag)::value;
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                unsigned off[4];
                f32x4 old[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = nLane + j * 32 + 8 * q;
                    off[q] = (n < a.Cout && pixOff[i] != OOB) ? pixOff[i] + (unsigned)n * 4u : OOB;
                    if constexpr (ACC) old[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srdO, (int)off[q], 0, 0));
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v = f32x4{ acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3] };
                    if constexpr (ACC) v += old[q];
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), srdO, (int)off[q], 0, 0);
                }
            }
    };
    if (a.accumulate) store_swapped(std::true_type{});
    else store_swapped(std::false_type{});

    // Fused GroupNorm statistics of the OUTPUT (forward only).  A BM-row tile touches at most two images (HW >= BM):
    // slot 0 = rows before `split`, slot 1 = the rest.  Three fixed-order stages, one writer per (image, tile, group):
    //   1. per lane, fp32: sum and sum of squares of each PG-channel piece of its TI pixels, per slot -> LDS
    //      (a lane holds 16 channels per 32-wide block as 4 quads; PG = 4, or 2 when a group has only 2 channels);
    //   2. per (piece, slot), fp64: over the 2 waves and 32 pixel lanes that hold the piece, in index order;
    //   3. per (group, slot), fp64: over the pieces of the group, in channel order.
    if constexpr (MODE == 0 && ZB == 0) {
        if (a.stats != nullptr) {
            const int nLo = m0 / a.HW;
            const int split = (nLo + 1) * a.HW - m0;
            float *sP = smem;                                      // [256 threads][2 slots][NP][2], tiles are dead now
            auto stage1 = [&](auto pgTag) {
                constexpr int PG = decltype(pgTag)::value;         // channels per piece
                constexpr int PQ = 4 / PG;                         // pieces per register quad
                constexpr int NP = NJ * 4 * PQ;                    // pieces per lane
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int h = 0; h < PQ; ++h) {
                            float s[2] = { 0.f, 0.f }, ss[2] = { 0.f, 0.f };
#pragma unroll
                            for (int i = 0; i < TI; ++i) {
                                const int row = wm * (BM / 2) + i * 32 + (lane & 31);
                                float t = 0.f, tt = 0.f;
#pragma unroll
                                for (int e = 0; e < PG; ++e) {
                                    const float v = acc[i][j][4 * q + h * PG + e];
                                    t += v;
                                    tt = fmaf(v, v, tt);
                                }
                                const bool live = m0 + row < a.M, hi = row >= split;
                                s[0] += (live && !hi) ? t : 0.f; ss[0] += (live && !hi) ? tt : 0.f;
                                s[1] += (live && hi) ? t : 0.f;  ss[1] += (live && hi) ? tt : 0.f;
                            }
#pragma unroll
                            for (int sl = 0; sl < 2; ++sl) {
                                float *o = sP + ((tid * 2 + sl) * NP + (j * 4 + q) * PQ + h) * 2;
                                o[0] = s[sl]; o[1] = ss[sl];
                            }
                        }
            };
            const int PGr = (a.cpg >= 4) ? 4 : 2;                  // launch_igemm admits cpg = 2 or a multiple of 4
            if (PGr == 4) stage1(std::integral_constant<int, 4>{});
            else stage1(std::integral_constant<int, 2>{});
            __syncthreads();
            const int PQr = 4 / PGr, NPr = NJ * 4 * PQr;
            const int pieces = BN / PGr;                           // pieces of the tile's BN channels
            double *sC = reinterpret_cast<double *>(smem + 256 * 2 * NPr * 2);      // [pieces][2 slots][2]
            for (int w = tid; w < pieces * 2; w += 256) {
                const int pc = w >> 1, sl = w & 1;
                const int c = pc * PGr;                            // first channel of the piece within the tile
                const int wnP = c / (BN / 2), l = c - wnP * (BN / 2);
                const int j = l >> 5, q = (l & 31) >> 3, half = ((l & 31) & 7) >> 2, h = ((l & 3) / PGr);
                double s1 = 0.0, s2 = 0.0;
                for (int wmP = 0; wmP < 2; ++wmP)
                    for (int pl = 0; pl < 32; ++pl) {
                        const int t = (wmP * 2 + wnP) * 64 + half * 32 + pl;
                        const float *o = sP + ((t * 2 + sl) * NPr + (j * 4 + q) * PQr + h) * 2;
                        s1 += (double)o[0]; s2 += (double)o[1];
                    }
                sC[w * 2] = s1; sC[w * 2 + 1] = s2;
            }
            __syncthreads();
            const int groupsInTile = BN / a.cpg;
            if (tid < groupsInTile * 2) {
                const int gi = tid >> 1, sl = tid & 1;
                const int n = nLo + sl;
                const int firstRow = sl ? split : 0;
                if (n < a.B && m0 + firstRow < a.M && (sl == 0 || split < BM)) {
                    double s1 = 0.0, s2 = 0.0;
                    const int ppg = a.cpg / PGr;                   // pieces per group
                    for (int pc = gi * ppg; pc < (gi + 1) * ppg; ++pc) { s1 += sC[(pc * 2 + sl) * 2]; s2 += sC[(pc * 2 + sl) * 2 + 1]; }
                    const int g = (n0 + gi * a.cpg) / a.cpg;
                    if (g < a.G) {
                        const int k = mt - (int)(((long long)n * a.HW) / BM);       // tile index within the image
                        double *o = a.stats + (((long long)n * a.nchunks + k) * a.G + g) * 2;
                        o[0] = s1; o[1] = s2;
                    }
                }
            }
        }
    }
    if (a.clk && tid == 0) {
        long long *c = a.clk + (long long)blockIdx.x * 8;
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        c[0] = tc0; c[1] = tw0; c[2] = tc1; c[3] = tc2; c[4] = clock64(); c[5] = wall_clock64(); c[6] = hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(hw));
        c[7] = hw;
    }
}

// ---------------------------------------------------------------------------------------------- Winograd F(2x2, 3x3)

// The stride-1 3x3 convolutions of inference plans run as Winograd F(2x2,3x3): 16 multiplies per 2x2 output tile and
// (ci, co) pair instead of 36.  V[xi][t][c] = (B^T d B)[xi] for the 4x4 input patch d of output tile t (zero padded),
// M[xi] = V[xi] U[xi]^T (16 GEMMs through igemm_conv_kernel, one batched launch), Y = A^T M A + bias.
// HBM-bound elementwise passes, one tile x 4 channels per thread, coalesced along the channel.
__global__ __launch_bounds__(256)
void wino_in_kernel(const float *__restrict__ in, float *__restrict__ V, int B, int H, int W, int C, int ldIn, int Th, int Tw)
{
    const int C4 = C >> 2;
    const long long T = (long long)B * Th * Tw;
    const long long items = T * C4;
    for (long long it = (long long)blockIdx.x * 256 + threadIdx.x; it < items; it += (long long)gridDim.x * 256) {
        const int c4 = (int)(it % C4);
        const long long t = it / C4;
        const int tx = (int)(t % Tw);
        const int ty = (int)((t / Tw) % Th);
        const int n = (int)(t / ((long long)Tw * Th));
        f32x4 d[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int y = 2 * ty - 1 + a;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int x = 2 * tx - 1 + b;
                if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W)
                    d[a][b] = *reinterpret_cast<const f32x4 *>(in + (((long long)n * H + y) * W + x) * ldIn + 4 * c4);
                else
                    d[a][b] = f32x4{ 0.f, 0.f, 0.f, 0.f };
            }
        }
        f32x4 w[4][4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            w[0][b] = d[0][b] - d[2][b];
            w[1][b] = d[1][b] + d[2][b];
            w[2][b] = d[2][b] - d[1][b];
            w[3][b] = d[1][b] - d[3][b];
        }
        float *o = V + t * C + 4 * c4;
        const long long zs = T * C;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<f32x4 *>(o + (4 * i + 0) * zs) = w[i][0] - w[i][2];
            *reinterpret_cast<f32x4 *>(o + (4 * i + 1) * zs) = w[i][1] + w[i][2];
            *reinterpret_cast<f32x4 *>(o + (4 * i + 2) * zs) = w[i][2] - w[i][1];
            *reinterpret_cast<f32x4 *>(o + (4 * i + 3) * zs) = w[i][1] - w[i][3];
        }
    }
}

// grid (nchunks, B): block k of image n transforms tiles [k*tpb, (k+1)*tpb) and writes the fp64 GroupNorm partial sums
// of what it produced to stats[n][k][g] (fixed order: tiles per thread, then the S tile lanes, then the channels of
// the group).  256 threads = S tile lanes x C/4 channel quads.
__global__ __launch_bounds__(256)
void wino_out_kernel(const float *__restrict__ M, const float *__restrict__ bias, float *__restrict__ out,
                     double *__restrict__ stats, int B, int H, int W, int C, int ldOut, int Th, int Tw, int tpb,
                     int G, int nchunks)
{
    __shared__ double sS[256 * 8];
    const int tid = threadIdx.x;
    const int C4 = C >> 2, S = 256 / C4;
    const int c4 = tid % C4, sub = tid / C4;
    const int n = blockIdx.y, k = blockIdx.x;
    const int Timg = Th * Tw;
    const long long T = (long long)B * Timg;
    const long long zs = T * C;
    int t1 = (k + 1) * tpb; if (t1 > Timg) t1 = Timg;
    f32x4 bv = f32x4{ 0.f, 0.f, 0.f, 0.f };
    if (bias) bv = *reinterpret_cast<const f32x4 *>(bias + 4 * c4);
    f32x4 s1 = f32x4{ 0.f, 0.f, 0.f, 0.f }, s2 = f32x4{ 0.f, 0.f, 0.f, 0.f };
    for (int tl = k * tpb + sub; tl < t1; tl += S) {
        const int ty = tl / Tw, tx = tl - ty * Tw;
        const float *m = M + ((long long)n * Timg + tl) * C + 4 * c4;
        f32x4 r[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) r[i][j] = *reinterpret_cast<const f32x4 *>(m + (4 * i + j) * zs);
        f32x4 q[2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            q[0][j] = r[0][j] + r[1][j] + r[2][j];
            q[1][j] = r[1][j] - r[2][j] - r[3][j];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const f32x4 y0 = q[i][0] + q[i][1] + q[i][2] + bv;
            const f32x4 y1 = q[i][1] - q[i][2] - q[i][3] + bv;
            float *o = out + (((long long)n * H + 2 * ty + i) * W + 2 * tx) * ldOut + 4 * c4;
            *reinterpret_cast<f32x4 *>(o) = y0;
            *reinterpret_cast<f32x4 *>(o + ldOut) = y1;
            s1 += y0; s1 += y1;
            s2 += y0 * y0; s2 += y1 * y1;
        }
    }
    if (!stats) return;
#pragma unroll
    for (int e = 0; e < 4; ++e) { sS[tid * 8 + e] = (double)s1[e]; sS[tid * 8 + 4 + e] = (double)s2[e]; }
    __syncthreads();
    if (tid < G) {
        const int cpg = C / G;
        double a = 0.0, b = 0.0;
        for (int sb = 0; sb < S; ++sb)
            for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) {
                const int th = sb * C4 + (c >> 2);
                a += sS[th * 8 + (c & 3)];
                b += sS[th * 8 + 4 + (c & 3)];
            }
        double *o = stats + (((long long)n * nchunks + k) * G + tid) * 2;
        o[0] = a; o[1] = b;
    }
}

// F(4x4, 3x3): 36 multiplies per 4x4 output tile and (ci, co) pair instead of 144 (Lavin & Gray 2016, interpolation
// points 0, +-1, +-2, inf).  6x6 input patches at stride 4; H and W need not be multiples of 4 (partial tiles read zeros
// and their surplus outputs are neither stored nor counted).  Same launch structure as the F(2x2,3x3) pair.
template <typename V>
__device__ __forceinline__ void wino4_bt(const V (&d)[6], V (&o)[6])
{
    o[0] = 4.f * d[0] - 5.f * d[2] + d[4];
    o[1] = -4.f * d[1] - 4.f * d[2] + d[3] + d[4];
    o[2] = 4.f * d[1] - 4.f * d[2] - d[3] + d[4];
    o[3] = -2.f * d[1] - d[2] + 2.f * d[3] + d[4];
    o[4] = 2.f * d[1] - d[2] - 2.f * d[3] + d[4];
    o[5] = 4.f * d[1] - 5.f * d[3] + d[5];
}

template <typename V>
__device__ __forceinline__ void wino4_at(const V (&m)[6], V (&o)[4])
{
    o[0] = m[0] + m[1] + m[2] + m[3] + m[4];
    o[1] = m[1] - m[2] + 2.f * m[3] - 2.f * m[4];
    o[2] = m[1] + m[2] + 4.f * m[3] + 4.f * m[4];
    o[3] = m[1] - m[2] + 8.f * m[3] - 8.f * m[4] + m[5];
}

// one tile x 2 channels per thread
// DEFER: 0 = plain gather; 1 / 2 = the producer's GroupNorm (x*scale + shift; 2: + ReLU) is applied while gathering.
// The deferred form loads through clamped addresses and masks afterwards: a use inside the bounds branch would put a
// vmcnt(0) after every load and serialise the 36 gathers of a tile.
template <int DEFER>
__global__ __launch_bounds__(256)
void wino4_in_kernel(const float *__restrict__ in, float *__restrict__ V, int B, int H, int W, int C, int ldIn, int Th, int Tw,
                     const float *__restrict__ coeff)
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
        // deferred GroupNorm of the producer: x*scale + shift (+ReLU) applied to the pixels as they are gathered
        f32x4 ss = f32x4{ 1.f, 0.f, 1.f, 0.f };
        if (DEFER) ss = *reinterpret_cast<const f32x4 *>(coeff + ((long long)n * C + 2 * c2) * 2);
        f32x2 w[6][6];                               // w[i][b] = (B^T d)[i][b]
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            const int x = 4 * tx - 1 + b;
            f32x2 col[6], o[6];
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                const int y = 4 * ty - 1 + a;
                if (DEFER) {
                    const int yc = min(max(y, 0), H - 1), xc = min(max(x, 0), W - 1);
                    col[a] = *reinterpret_cast<const f32x2 *>(in + (((long long)n * H + yc) * W + xc) * ldIn + 2 * c2);
                } else if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W)
                    col[a] = *reinterpret_cast<const f32x2 *>(in + (((long long)n * H + y) * W + x) * ldIn + 2 * c2);
                else
                    col[a] = f32x2{ 0.f, 0.f };
            }
            if (DEFER) {
#pragma unroll
                for (int a = 0; a < 6; ++a) {
                    const int y = 4 * ty - 1 + a;
                    const bool inb = ((unsigned)y < (unsigned)H) & ((unsigned)x < (unsigned)W);
                    float v0 = fmaf(col[a][0], ss[0], ss[1]), v1 = fmaf(col[a][1], ss[2], ss[3]);   // one rounding, as every apply site
                    if (DEFER == 2) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
                    col[a] = f32x2{ inb ? v0 : 0.f, inb ? v1 : 0.f };
                }
            }
            wino4_bt(col, o);
#pragma unroll
            for (int i = 0; i < 6; ++i) w[i][b] = o[i];
        }
        float *op = V + t * C + 2 * c2;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            f32x2 o[6];
            wino4_bt(w[i], o);
#pragma unroll
            for (int j = 0; j < 6; ++j) *reinterpret_cast<f32x2 *>(op + (6 * i + j) * zs) = o[j];
        }
    }
}

// grid (nchunks, B); 256 threads = S tile lanes x C/2 channel pairs... C/2 may exceed 256: a block covers CB = min(C, 512)
// channels and blockIdx.z walks the channel blocks.  Statistics: one fp64 partial per (image, chunk, group).
__global__ __launch_bounds__(256)
void wino4_out_kernel(const float *__restrict__ M, const float *__restrict__ bias, float *__restrict__ out,
                      double *__restrict__ stats, int B, int H, int W, int C, int ldOut, int Th, int Tw, int tpb,
                      int G, int nchunks, int accumulate)
{
    __shared__ double sS[256 * 4];
    const int tid = threadIdx.x;
    const int CB = C < 512 ? C : 512;                 // channels per block
    const int C2 = CB >> 1, S = 256 / C2;
    const int cb0 = blockIdx.z * CB;
    const int c2 = tid % C2, sub = tid / C2;
    const int cch = cb0 + 2 * c2;                      // first of this thread's two channels
    const int n = blockIdx.y, k = blockIdx.x;
    const int Timg = Th * Tw;
    const long long T = (long long)B * Timg;
    const long long zs = T * C;
    int t1 = (k + 1) * tpb; if (t1 > Timg) t1 = Timg;
    f32x2 bv = f32x2{ 0.f, 0.f };
    if (bias) bv = *reinterpret_cast<const f32x2 *>(bias + cch);
    f32x2 s1 = f32x2{ 0.f, 0.f }, s2 = f32x2{ 0.f, 0.f };
    for (int tl = k * tpb + sub; tl < t1; tl += S) {
        const int ty = tl / Tw, tx = tl - ty * Tw;
        const float *m = M + ((long long)n * Timg + tl) * C + cch;
        f32x2 q[4][6];                               // q[p][j] = (A^T r)[p][j]
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            f32x2 col[6], o[4];
#pragma unroll
            for (int i = 0; i < 6; ++i) col[i] = *reinterpret_cast<const f32x2 *>(m + (6 * i + j) * zs);
            wino4_at(col, o);
#pragma unroll
            for (int pI = 0; pI < 4; ++pI) q[pI][j] = o[pI];
        }
#pragma unroll
        for (int pI = 0; pI < 4; ++pI) {
            f32x2 y[4];
            wino4_at(q[pI], y);
            const int oy = 4 * ty + pI;
            if (oy >= H) continue;
#pragma unroll
            for (int qI = 0; qI < 4; ++qI) {
                const int ox = 4 * tx + qI;
                if (ox >= W) continue;
                f32x2 v = y[qI] + bv;
                f32x2 *dst = reinterpret_cast<f32x2 *>(out + (((long long)n * H + oy) * W + ox) * ldOut + cch);
                if (accumulate) v += *dst;                     // data gradients: second producer of a gradient tensor
                *dst = v;
                s1 += v; s2 += v * v;
            }
        }
    }
    if (!stats) return;
    sS[tid * 4 + 0] = (double)s1[0]; sS[tid * 4 + 1] = (double)s1[1];
    sS[tid * 4 + 2] = (double)s2[0]; sS[tid * 4 + 3] = (double)s2[1];
    __syncthreads();
    const int cpg = C / G;
    const int gPerBlock = CB / cpg;                    // groups whose channels all live in this block
    if (tid < gPerBlock) {
        double a = 0.0, b = 0.0;
        for (int sb = 0; sb < S; ++sb)
            for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) {
                const int th = sb * C2 + (c >> 1);
                a += sS[th * 4 + (c & 1)];
                b += sS[th * 4 + 2 + (c & 1)];
            }
        const int g = cb0 / cpg + tid;
        double *o = stats + (((long long)n * nchunks + k) * G + g) * 2;
        o[0] = a; o[1] = b;
    }
}

// F(6x6, 3x3): 64 multiplies per 6x6 output tile and (ci, co) pair instead of 324 (5.06x fewer than the direct form,
// 1.27x fewer than F(4x4,3x3)); interpolation points 0, +-1, +-2, +-1/2, inf with the scaling of Lavin's wincnn set.
// Weights are transformed in float64 on the host side; measured error of a 256-channel layer against a float64
// convolution: 1.6e-5 of max|ref| (F(4x4,3x3): 1.2e-5, direct fp32: 2e-7).  8x8 input patches at stride 6; partial tiles
// as in the F(4x4,3x3) kernels.  Inference plans only.
template <typename V>
__device__ __forceinline__ void wino6_bt(const V (&d)[8], V (&o)[8])
{
    const V a = d[2] + d[6] - 4.25f * d[4], b = d[1] + d[5] - 4.25f * d[3];
    const V c = d[6] + 0.25f * d[2] - 1.25f * d[4], e = 0.5f * d[1] - 2.5f * d[3] + 2.f * d[5];
    const V f = d[6] + 4.f * d[2] - 5.f * d[4], g = 2.f * d[1] - 2.5f * d[3] + 0.5f * d[5];
    o[0] = d[0] - d[6] + 5.25f * (d[4] - d[2]);
    o[1] = a + b; o[2] = a - b;
    o[3] = c + e; o[4] = c - e;
    o[5] = f + g; o[6] = f - g;
    o[7] = d[7] - d[1] + 5.25f * (d[3] - d[5]);
}

template <typename V>
__device__ __forceinline__ void wino6_at(const V (&m)[8], V (&o)[6])
{
    const V s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
    const V s56 = m[5] + m[6], d56 = m[5] - m[6];
    o[0] = m[0] + s12 + s34 + 32.f * s56;
    o[1] = d12 + 2.f * d34 + 16.f * d56;
    o[2] = s12 + 4.f * s34 + 8.f * s56;
    o[3] = d12 + 8.f * d34 + 4.f * d56;
    o[4] = s12 + 16.f * s34 + 2.f * s56;
    o[5] = d12 + 32.f * d34 + d56 + m[7];
}

// fp32 -> bf16 (round to nearest even) and back; a = h1 + h2 + h3 splits 24 mantissa bits exactly
__device__ __forceinline__ unsigned bf16_rn(float x)
{
    unsigned u = __builtin_bit_cast(unsigned, x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ float bf16_f(unsigned h) { return __builtin_bit_cast(float, h << 16); }
__device__ __forceinline__ void bf16_split3(float a, unsigned &h1, unsigned &h2, unsigned &h3)
{
    h1 = bf16_rn(a);
    const float r1 = a - bf16_f(h1);
    h2 = bf16_rn(r1);
    h3 = bf16_rn(r1 - bf16_f(h2));
}

// one tile x 2 channels per thread; DEFER as in wino4_in_kernel.  SPLIT: V is written as three bf16 planes
// ([plane][64][tiles][C], the operand form of csrc/xl_gemm_split.hip) instead of fp32.
// The interleaved-plane forms are held to 168 VGPRs = three waves per SIMD (the deferred one would take 178 and runs 20 %
// slower at two waves: 413 vs 346 us per 512-channel launch; at 168 it keeps 5 values in scratch and takes 364).
// FOLD (with DEFER = 1): `in` is the RAW output of a convolution whose GroupNorm(+ReLU, +residual, +ReLU) apply pass was not
// run: every in-image pixel becomes v = relu_out(relu_in(fmaf(x, scale, shift)) + res) on load (the arithmetic of
// gn_apply_kernel, flags as XL_GN_*), and the thread that owns a pixel - the tile whose 6 x 6 output footprint contains it -
// also writes v to `side`: the activation is materialised for its other consumers (the residual branch) by the pass that had
// to read it anyway.  `side` must not alias `in` or `res` (other tiles read their halo pixels from those).
// resCoef: the residual is itself the RAW output of a convolution whose GroupNorm + ReLU apply was deferred (its only consumer is
// this addition): {scale, shift} pairs like `coeff`, applied with ReLU while the residual is read.
struct WinoFold { const float *res; float *side; int ldRes, ldSide, flags; const float *resCoef; };

template <int DEFER, int SPLIT = 0, int FOLD = 0>
__global__ __launch_bounds__(256, (SPLIT == 2 ? 3 : 1))
void wino6_in_kernel(const float *__restrict__ in, float *__restrict__ V, int B, int H, int W, int C, int ldIn, int Th, int Tw,
                     const float *__restrict__ coeff, WinoFold fold, const float *__restrict__ pairScale)
{
    static_assert(!FOLD || DEFER == 1, "the fold form takes its ReLU / residual flags at run time");
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
        f32x4 ss = f32x4{ 1.f, 0.f, 1.f, 0.f };
        if (DEFER) ss = *reinterpret_cast<const f32x4 *>(coeff + ((long long)n * C + 2 * c2) * 2);
        f32x4 rs = f32x4{ 1.f, 0.f, 1.f, 0.f };
        if constexpr (FOLD != 0) {
            if (fold.resCoef) rs = *reinterpret_cast<const f32x4 *>(fold.resCoef + ((long long)n * C + 2 * c2) * 2);
        }
        f32x2 w[8][8];                               // w[i][b] = (B^T d)[i][b]
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const int x = 6 * tx - 1 + b;
            f32x2 col[8], o[8];
#pragma unroll
            for (int a = 0; a < 8; ++a) {
                const int y = 6 * ty - 1 + a;
                if (DEFER) {
                    const int yc = min(max(y, 0), H - 1), xc = min(max(x, 0), W - 1);
                    col[a] = *reinterpret_cast<const f32x2 *>(in + (((long long)n * H + yc) * W + xc) * ldIn + 2 * c2);
                } else if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W)
                    col[a] = *reinterpret_cast<const f32x2 *>(in + (((long long)n * H + y) * W + x) * ldIn + 2 * c2);
                else
                    col[a] = f32x2{ 0.f, 0.f };
            }
            if (DEFER) {
                f32x2 rcol[8];
                if constexpr (FOLD != 0) {
                    if (fold.res) {
#pragma unroll
                        for (int a = 0; a < 8; ++a) {
                            const int yc = min(max(6 * ty - 1 + a, 0), H - 1), xc = min(max(x, 0), W - 1);
                            rcol[a] = *reinterpret_cast<const f32x2 *>(fold.res + (((long long)n * H + yc) * W + xc) * fold.ldRes + 2 * c2);
                        }
                        if (fold.resCoef) {                  // the residual's own deferred GroupNorm + ReLU (gn_apply's arithmetic)
#pragma unroll
                            for (int a = 0; a < 8; ++a)
                                rcol[a] = f32x2{ fmaxf(fmaf(rcol[a][0], rs[0], rs[1]), 0.f), fmaxf(fmaf(rcol[a][1], rs[2], rs[3]), 0.f) };
                        }
                    }
                }
#pragma unroll
                for (int a = 0; a < 8; ++a) {
                    const int y = 6 * ty - 1 + a;
                    const bool inb = ((unsigned)y < (unsigned)H) & ((unsigned)x < (unsigned)W);
                    float v0 = fmaf(col[a][0], ss[0], ss[1]), v1 = fmaf(col[a][1], ss[2], ss[3]);   // one rounding, as every apply site
                    if (DEFER == 2) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
                    if constexpr (FOLD != 0) {
                        if (fold.flags & XL_GN_RELU_IN) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
                        if (fold.res) { v0 += rcol[a][0]; v1 += rcol[a][1]; }
                        if (fold.flags & XL_GN_RELU_OUT) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
                        if (inb && a >= 1 && a <= 6 && b >= 1 && b <= 6)
                            *reinterpret_cast<f32x2 *>(fold.side + (((long long)n * H + y) * W + x) * fold.ldSide + 2 * c2) = f32x2{ v0, v1 };
                    }
                    col[a] = f32x2{ inb ? v0 : 0.f, inb ? v1 : 0.f };
                }
            }
            wino6_bt(col, o);
#pragma unroll
            for (int i = 0; i < 8; ++i) w[i][b] = o[i];
        }
        if constexpr (SPLIT == 2) {
            // interleaved planes (256 x 256 split GEMM): V[xi][t][c / 16][plane][c % 16] bf16.  The 64 lanes of a wave hold
            // 128 consecutive channels of one tile = 8 chunks = one contiguous 768-byte piece per frequency, but a lane's
            // three words (its channel pair in the three planes) lie 32 bytes apart.  They are exchanged through a
            // wave-private LDS row so that every lane stores 12 contiguous bytes: one dwordx3 store instead of three
            // scattered dword stores per frequency (C % 128 == 0 and whole waves: checked by the launcher).
            __shared__ unsigned sX[4][2][192];
            const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
            const int c = 2 * c2;
            unsigned *op = reinterpret_cast<unsigned *>(V) + ((t * (C >> 4) + ((c - 2 * lane) >> 4)) * 48 >> 1) + 3 * lane;
            const long long zw = (zs * 3) >> 1;                                           // words per frequency
            const int wr = (lane >> 3) * 24 + (lane & 7);                                 // chunk, pair within the chunk
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                f32x2 o[8];
                wino6_bt(w[i], o);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    unsigned a1, a2, a3, b1, b2, b3;
                    bf16_split3(o[j][0], a1, a2, a3);
                    bf16_split3(o[j][1], b1, b2, b3);
                    unsigned *row = sX[wv][j & 1];
                    row[wr] = a1 | (b1 << 16); row[wr + 8] = a2 | (b2 << 16); row[wr + 16] = a3 | (b3 << 16);
                    typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
                    const u32x3 v3 = u32x3{ row[3 * lane], row[3 * lane + 1], row[3 * lane + 2] };
                    *reinterpret_cast<u32x3 *>(op + (8 * i + j) * zw) = v3;
                }
            }
            continue;
        }
        if constexpr (SPLIT == 3) {
            // fp16 pairs (round 5, XL_CONV_PAIR_F16, csrc/xl_gemm_pair.hip): V[xi][t][c / 8][2][8] fp16 = {hi, (v - hi) 2^11} of
            // v = V * scale - a 16-byte slot of hi then one of lo' per 8 channels, so a K-step of 16 channels is the four slots
            // hi(k 0-7) lo'(k 0-7) hi(k 8-15) lo'(k 8-15).  A row is C words of 4 bytes like the fp32 form, and the four lanes
            // that hold 8 consecutive channels own the 32 bytes of their chunk: a lane's two words (hi and lo' of its channel pair)
            // are exchanged inside the QUAD on the DPP path - neighbours trade one word (even lanes collect the hi pair, odd lanes
            // the lo' pair), then lanes 1 and 2 swap - so that lane q stores bytes 8 q .. 8 q + 7: the same dwordx2 store, at the
            // same address, as the fp32 form.  (Through a wave-private LDS row instead: 0.60 / 1.07 ms per 512-channel launch at 95
            // frames, plain / fold form, against 0.51 / 0.83 for fp32.)
            typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            const float sc = pairScale[0];
            const bool odd = (threadIdx.x & 1) != 0, mid = ((threadIdx.x + 1) & 2) != 0;      // lanes 1 and 2 of a quad
            unsigned *op = reinterpret_cast<unsigned *>(V) + t * C + 2 * c2;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                f32x2 o[8];
                wino6_bt(w[i], o);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const f32x2 x = o[j] * sc;
                    const f16x2 h = __builtin_convertvector(x, f16x2);
                    const f16x2 l = __builtin_convertvector((x - __builtin_convertvector(h, f32x2)) * 2048.f, f16x2);
                    const int H = __builtin_bit_cast(int, h), L = __builtin_bit_cast(int, l);
                    const int recv = __builtin_amdgcn_update_dpp(0, odd ? H : L, 0xB1, 0xF, 0xF, false);      // quad_perm [1,0,3,2]
                    int p0 = odd ? recv : H, p1 = odd ? L : recv;                 // even lanes: (hi, hi'), odd lanes: (lo', lo'')
                    const int s0 = __builtin_amdgcn_update_dpp(0, p0, 0xD8, 0xF, 0xF, false);                 // quad_perm [0,2,1,3]
                    const int s1 = __builtin_amdgcn_update_dpp(0, p1, 0xD8, 0xF, 0xF, false);
                    p0 = mid ? s0 : p0; p1 = mid ? s1 : p1;
                    *reinterpret_cast<u32x2 *>(op + (8 * i + j) * zs) = u32x2{ (unsigned)p0, (unsigned)p1 };
                }
            }
            continue;
        }
        if (SPLIT) {
            unsigned *op = reinterpret_cast<unsigned *>(V) + ((t * C + 2 * c2) >> 1);      // 2 bf16 per 32-bit word
            const long long plane = (64 * zs) >> 1;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                f32x2 o[8];
                wino6_bt(w[i], o);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    unsigned a1, a2, a3, b1, b2, b3;
                    bf16_split3(o[j][0], a1, a2, a3);
                    bf16_split3(o[j][1], b1, b2, b3);
                    unsigned *q = op + (((8 * i + j) * zs) >> 1);
                    q[0] = a1 | (b1 << 16); q[plane] = a2 | (b2 << 16); q[2 * plane] = a3 | (b3 << 16);
                }
            }
            continue;
        }
        float *op = V + t * C + 2 * c2;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            f32x2 o[8];
            wino6_bt(w[i], o);
#pragma unroll
            for (int j = 0; j < 8; ++j) *reinterpret_cast<f32x2 *>(op + (8 * i + j) * zs) = o[j];
        }
    }
}

// grid (nchunks, B, C / CB), as wino4_out_kernel: M [64][B*Th*Tw][C] -> out [B,H,W,C] + bias + GroupNorm partial sums.
// VW channels per thread.  VW = 1: the factored transform on a 6x8 intermediate, one dword per lane and access.
// VW = 2 (channel counts that are multiples of 512): 8-byte accesses through a buffer descriptor with scalar plane
// offsets and a column-accumulated row transform - the factored form with two channels needs more than 256 VGPRs.
template <int VW> struct WinoVec { typedef float type; };
template <> struct WinoVec<2> { typedef f32x2 type; };
__device__ __forceinline__ float wino_lane(const float &v, int) { return v; }
__device__ __forceinline__ float wino_lane(const f32x2 &v, int i) { return v[i]; }

template <int VW>
__global__ __launch_bounds__(256)
void wino6_out_kernel(const float *__restrict__ M, const float *__restrict__ bias, float *__restrict__ out,
                      double *__restrict__ stats, int B, int H, int W, int C, int ldOut, int Th, int Tw, int tpb,
                      int G, int nchunks, int accumulate, int tileMajor)
{
    typedef typename WinoVec<VW>::type V;
    __shared__ double sS[256 * 2 * VW];
    const int tid = threadIdx.x;
    const int CB = C < 256 * VW ? C : 256 * VW;       // channels per block
    const int CV = CB / VW, S = 256 / CV;
    const int cb0 = blockIdx.z * CB;
    const int cv = tid % CV, sub = tid / CV;
    const int cch = cb0 + VW * cv;                     // first of this thread's channels
    const int n = blockIdx.y, k = blockIdx.x;
    const int Timg = Th * Tw;
    const long long T = (long long)B * Timg;
    // M is [64][tiles][C] (plane-major: what the fp32 GEMMs write) or - tileMajor, XL_CONV_M_TILE_MAJOR - [tiles][64][C]: the
    // 64 x C block of a tile is then ONE contiguous piece (128 KB at 512 channels) instead of 64 rows 14 MB apart
    const long long zs = tileMajor ? (long long)C : T * C;            // elements between frequency planes
    const long long ts = tileMajor ? 64LL * C : (long long)C;         // ... between tiles
    int t1 = (k + 1) * tpb; if (t1 > Timg) t1 = Timg;
    V bv = V(0.f);
    if (bias) bv = *reinterpret_cast<const V *>(bias + cch);
    V s1 = V(0.f), s2 = V(0.f);
    const __amdgpu_buffer_rsrc_t srdM = __builtin_amdgcn_make_buffer_rsrc((void *)M, 0, (int)(unsigned)(64 * T * C * 4 > 0xffffffffLL ? 0xffffffffLL : 64 * T * C * 4), 0x00020000);
    // VW = 2: the first frequency column of a tile is fetched while the tile BEFORE it is still being reduced and stored (round
    // 4): without that every tile started with an empty memory pipeline - 8 loads issued, one full HBM latency waited - and the
    // 36 stores of a tile went out with no read in flight behind them.
    typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
    const unsigned zsB = (unsigned)(zs * 4);
    auto tile_off = [&](int tl) { return (unsigned)(((long long)n * Timg + tl) * ts + cch) * 4u; };
    auto ldp = [&](unsigned vo, int plane) {
        // frequency planes through one buffer descriptor: the lane part of the address is ONE VGPR and the plane offset a
        // scalar (with flat pointers the compiler keeps 64 loop-invariant 64-bit addresses in registers)
        return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(srdM, (int)vo, (int)(plane * zsB), 0));
    };
    f32x2 colN[8];
    if constexpr (VW == 2) {
        if (k * tpb + sub < t1) {
            const unsigned vo0 = tile_off(k * tpb + sub);
#pragma unroll
            for (int i = 0; i < 8; ++i) colN[i] = ldp(vo0, 8 * i);
        }
    }
    for (int tl = k * tpb + sub; tl < t1; tl += S) {
        const int ty = tl / Tw, tx = tl - ty * Tw;
        const float *m = M + ((long long)n * Timg + tl) * ts + cch;
        if constexpr (VW == 1) {
            V q[6][8];                               // q[p][j] = (A^T r)[p][j]
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                V col[8], o[6];
#pragma unroll
                for (int i = 0; i < 8; ++i) col[i] = *reinterpret_cast<const V *>(m + (8 * i + j) * zs);
                wino6_at(col, o);
#pragma unroll
                for (int pI = 0; pI < 6; ++pI) q[pI][j] = o[pI];
            }
#pragma unroll
            for (int pI = 0; pI < 6; ++pI) {
                V y[6];
                wino6_at(q[pI], y);
                const int oy = 6 * ty + pI;
                if (oy >= H) continue;
#pragma unroll
                for (int qI = 0; qI < 6; ++qI) {
                    const int ox = 6 * tx + qI;
                    if (ox >= W) continue;
                    V v = y[qI] + bv;
                    V *dst = reinterpret_cast<V *>(out + (((long long)n * H + oy) * W + ox) * ldOut + cch);
                    if (accumulate) v += *dst;                     // data gradients: second producer of a gradient tensor
                    *dst = v;
                    s1 += v; s2 += v * v;
                }
            }
        } else {
            // Two channels per lane: 8-byte accesses (a dword per lane caps the streaming rate of this pass near 4.5 TB/s).
            // The 6x8 intermediate of the factored form does not fit beside 16-byte-wide columns in flight, so the row
            // transform is accumulated column by column: y[p][.] += (A^T)[., j] * (A^T col_j)[p]; 36 accumulators, one
            // column being reduced and one in flight.
            constexpr float AT[8][6] = { { 1.f, 0.f, 0.f, 0.f, 0.f, 0.f }, { 1.f, 1.f, 1.f, 1.f, 1.f, 1.f },
                                         { 1.f, -1.f, 1.f, -1.f, 1.f, -1.f }, { 1.f, 2.f, 4.f, 8.f, 16.f, 32.f },
                                         { 1.f, -2.f, 4.f, -8.f, 16.f, -32.f }, { 32.f, 16.f, 8.f, 4.f, 2.f, 1.f },
                                         { 32.f, -16.f, 8.f, -4.f, 2.f, -1.f }, { 0.f, 0.f, 0.f, 0.f, 0.f, 1.f } };
            V y[6][6];
#pragma unroll
            for (int pI = 0; pI < 6; ++pI)
#pragma unroll
                for (int qI = 0; qI < 6; ++qI) y[pI][qI] = bv;
            const unsigned vo = tile_off(tl);
            // the tile after this one (behind the last tile: this tile's own first column again - a valid address, L2 hits)
            const unsigned voNext = tile_off(tl + S < t1 ? tl + S : tl);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                V col[8], o[6];
#pragma unroll
                for (int i = 0; i < 8; ++i) col[i] = colN[i];
#pragma unroll
                for (int i = 0; i < 8; ++i) colN[i] = j + 1 < 8 ? ldp(vo, 8 * i + j + 1) : ldp(voNext, 8 * i);
                __builtin_amdgcn_sched_barrier(0);
                wino6_at(col, o);
#pragma unroll
                for (int pI = 0; pI < 6; ++pI)
#pragma unroll
                    for (int qI = 0; qI < 6; ++qI)
                        if (AT[j][qI] != 0.f) y[pI][qI] += AT[j][qI] * o[pI];
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int pI = 0; pI < 6; ++pI) {
                const int oy = 6 * ty + pI;
                if (oy >= H) continue;
#pragma unroll
                for (int qI = 0; qI < 6; ++qI) {
                    const int ox = 6 * tx + qI;
                    if (ox >= W) continue;
                    V v = y[pI][qI];
                    V *dst = reinterpret_cast<V *>(out + (((long long)n * H + oy) * W + ox) * ldOut + cch);
                    if (accumulate) v += *dst;
                    *dst = v;
                    s1 += v; s2 += v * v;
                }
            }
        }
    }
    if (!stats) return;
    const int cpg = C / G;
    const int lpg = cpg / VW;                          // lanes that hold one group
    if (S == 1 && lpg >= 1 && lpg <= 64 && (lpg & (lpg - 1)) == 0 && cpg == lpg * VW) {
        // every channel of the block lives in exactly one thread and a group in lpg neighbouring lanes of one wave: fp64
        // butterfly over those lanes (fixed order), one writer per (image, chunk, group) - no LDS, no barrier.  (The LDS
        // form below reads 16 doubles per group at a 256-byte lane stride: 32-way bank conflicts at the tail of every block.)
        double a = 0.0, b = 0.0;
#pragma unroll
        for (int e = 0; e < VW; ++e) { a += (double)wino_lane(s1, e); b += (double)wino_lane(s2, e); }
        for (int off = 1; off < lpg; off <<= 1) { a += __shfl_xor(a, off, 64); b += __shfl_xor(b, off, 64); }
        if ((tid & (lpg - 1)) == 0) {
            double *o = stats + (((long long)n * nchunks + k) * G + cch / cpg) * 2;
            o[0] = a; o[1] = b;
        }
        return;
    }
#pragma unroll
    for (int e = 0; e < VW; ++e) {
        sS[(tid * VW + e) * 2] = (double)wino_lane(s1, e);
        sS[(tid * VW + e) * 2 + 1] = (double)wino_lane(s2, e);
    }
    __syncthreads();
    const int gPerBlock = CB / cpg;                    // groups whose channels all live in this block
    if (tid < gPerBlock) {
        double a = 0.0, b = 0.0;
        for (int sb = 0; sb < S; ++sb)
            for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) {
                const int th = sb * CV + c / VW;
                a += sS[(th * VW + c % VW) * 2];
                b += sS[(th * VW + c % VW) * 2 + 1];
            }
        const int g = cb0 / cpg + tid;
        double *o = stats + (((long long)n * nchunks + k) * G + g) * 2;
        o[0] = a; o[1] = b;
    }
}

// F(6x6,3x3) output transform with the product M staged through LDS by DMA (round 3).  The register form above keeps one
// column of the 8 x 8 frequency block in flight per lane (8 loads of 8 bytes): ~32 KB per CU, and the pass runs at 4.8 TB/s.
// Here every WAVE is an independent unit - one (image, chunk of `tpb` tiles, 128-channel slice) - that owns a 32 KB ring
// in LDS: the 8 columns of a tile's frequency block, 8 planes x 512 bytes each, fetched by `buffer_load ... lds` (4
// instructions per column: lanes 0-31 one plane, lanes 32-63 the next), the column of the NEXT tile refilling a slot as
// soon as the current one has been reduced: 28-32 KB in flight per wave whatever the register pressure.  A lane reads its two
// channels of a plane with one ds_read_b64 (consecutive lanes, consecutive words).  In-order vmcnt bookkeeping: younger
// than the column being waited for are the other 7 columns (28 instructions) and, from the second tile on, the 36 stores
// of the tile before - every tile issues exactly 36, out-of-image pixels fall outside the buffer descriptor.
// GroupNorm partial sums: fp64 butterfly over the lanes of a group, one writer per (image, chunk, group).
__global__ __launch_bounds__(256, 1)
void wino6_out_dma_kernel(const float *__restrict__ M, const float *__restrict__ bias, float *__restrict__ out,
                          double *__restrict__ stats, int B, int H, int W, int C, int ldOut, int Th, int Tw, int tpb,
                          int G, int nchunks, long long units, int tileMajor)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char dsmW[];
    typedef __attribute__((address_space(3))) void lds_void;
    constexpr unsigned OOB = 0x80000000u;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long unit = (long long)blockIdx.x * 4 + wave;            // (image, chunk, slice), slice fastest
    if (unit >= units) return;
    const int nsl = C >> 7;
    const int slice = (int)(unit % nsl);
    const int k = (int)((unit / nsl) % nchunks);
    const int n = (int)(unit / ((long long)nsl * nchunks));
    unsigned char *ring = dsmW + wave * 32768;
    const int Timg = Th * Tw;
    const long long T = (long long)B * Timg;
    const unsigned zsB = (unsigned)((tileMajor ? (long long)C : T * C) * 4);   // bytes between frequency planes (host: M < 2 GiB)
    const long long tsB = (tileMajor ? 64LL * C : (long long)C) * 4;     // ... between tiles
    const int t0 = k * tpb;
    int t1 = t0 + tpb; if (t1 > Timg) t1 = Timg;
    const int cch = slice * 128 + 2 * lane;                              // first of this lane's two channels
    const __amdgpu_buffer_rsrc_t srdM = __builtin_amdgcn_make_buffer_rsrc((void *)M, 0, (int)(unsigned)(64ull * T * C * 4), 0x00020000);
    // lane part of a DMA address: plane parity (lane >> 5) -> +8 planes... see dma_col: planes 8 (2h + (lane >> 5)) + j
    const unsigned laneOff = (unsigned)(lane >> 5) * 8u * zsB + (unsigned)(lane & 31) * 16u + (unsigned)slice * 512u;
    auto dma_col = [&](int tl, int j) {                                  // column j of tile tl (of this image) -> ring slot j
        const unsigned vo = tl < t1 ? (unsigned)(((long long)n * Timg + tl) * tsB) + laneOff : OOB;
#pragma unroll
        for (int h = 0; h < 4; ++h)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srdM, (lds_void *)(ring + j * 4096 + h * 1024), 16, (int)vo,
                                                     (int)((unsigned)(16 * h + j) * zsB), 0, 0);
    };
    f32x2 bv = f32x2{ 0.f, 0.f };
    if (bias) bv = *reinterpret_cast<const f32x2 *>(bias + cch);
    f32x2 s1 = f32x2{ 0.f, 0.f }, s2 = f32x2{ 0.f, 0.f };
    constexpr float AT[8][6] = { { 1.f, 0.f, 0.f, 0.f, 0.f, 0.f }, { 1.f, 1.f, 1.f, 1.f, 1.f, 1.f },
                                 { 1.f, -1.f, 1.f, -1.f, 1.f, -1.f }, { 1.f, 2.f, 4.f, 8.f, 16.f, 32.f },
                                 { 1.f, -2.f, 4.f, -8.f, 16.f, -32.f }, { 32.f, 16.f, 8.f, 4.f, 2.f, 1.f },
                                 { 32.f, -16.f, 8.f, -4.f, 2.f, -1.f }, { 0.f, 0.f, 0.f, 0.f, 0.f, 1.f } };
    const long long outBytes = ((long long)B * H * W - 1) * ldOut * 4 + (long long)C * 4;
    const __amdgpu_buffer_rsrc_t srdO = __builtin_amdgcn_make_buffer_rsrc((void *)out, 0, (int)(unsigned)outBytes, 0x00020000);
#pragma unroll
    for (int j = 0; j < 8; ++j) dma_col(t0, j);
    for (int tl = t0; tl < t1; ++tl) {
        const int ty = tl / Tw, tx = tl - ty * Tw;
        f32x2 y[6][6];
#pragma unroll
        for (int pI = 0; pI < 6; ++pI)
#pragma unroll
            for (int qI = 0; qI < 6; ++qI) y[pI][qI] = bv;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            // column j has landed when at most the 28 column fetches behind it (+ the 36 stores of the tile before) remain
            if (tl == t0) __builtin_amdgcn_s_waitcnt(0x0F70 | (28 & 15) | ((28 >> 4) << 14));
            else __builtin_amdgcn_s_waitcnt(0x0F70 | (63 & 15) | ((63 >> 4) << 14));
            __builtin_amdgcn_sched_barrier(0);
            f32x2 col[8], o[6];
#pragma unroll
            for (int i = 0; i < 8; ++i) col[i] = *reinterpret_cast<const f32x2 *>(ring + j * 4096 + i * 512 + lane * 8);
            wino6_at(col, o);
#pragma unroll
            for (int pI = 0; pI < 6; ++pI)
#pragma unroll
                for (int qI = 0; qI < 6; ++qI)
                    if (AT[j][qI] != 0.f) y[pI][qI] += AT[j][qI] * o[pI];
            // (the reads above have returned - their values were consumed - before the slot is refilled)
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xC07F);                          // lgkmcnt(0)
            dma_col(tl + 1, j);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int pI = 0; pI < 6; ++pI) {
            const int oy = 6 * ty + pI;
#pragma unroll
            for (int qI = 0; qI < 6; ++qI) {
                const int ox = 6 * tx + qI;
                const bool live = (oy < H) & (ox < W);
                const f32x2 v = y[pI][qI];
                const unsigned off = live ? (unsigned)(((((long long)n * H + oy) * W + ox) * ldOut + cch) * 4) : OOB;
                typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), srdO, (int)off, 0, 0);
                if (live) { s1 += v; s2 += v * v; }
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                                  // vmcnt(0): the dummy fetches behind the last tile
    if (!stats) return;
    const int cpg = C / G, lpg = cpg >> 1;                               // lanes per group (host: a power of two, 1 .. 64)
    double a = (double)s1[0] + (double)s1[1], b = (double)s2[0] + (double)s2[1];
    if (lpg == 0) {                                                      // one channel per group (never with 32 groups of >= 128 channels)
        double *o = stats + (((long long)n * nchunks + k) * G + cch) * 2;
        o[0] = (double)s1[0]; o[1] = (double)s2[0]; o[2] = (double)s1[1]; o[3] = (double)s2[1];
        return;
    }
    for (int off = 1; off < lpg; off <<= 1) { a += __shfl_xor(a, off, 64); b += __shfl_xor(b, off, 64); }
    if ((lane & (lpg - 1)) == 0) {
        double *o = stats + (((long long)n * nchunks + k) * G + cch / cpg) * 2;
        o[0] = a; o[1] = b;
    }
}

// ---------------------------------------------------------------------------------------------- GroupNorm

// grid (nchunks, B); T threads with T % (C/4) == 0.  stats[((n*nchunks + chunk)*G + g)*2 + {0,1}] = sum, sumsq
__global__ void gn_stats_kernel(const float *__restrict__ x, double *__restrict__ stats, int HW, int C, int ld,
                                int G, int nchunks)
{
    extern __shared__ __attribute__((aligned(16))) double sPart[];     // [T][8]
    const int T = blockDim.x, tid = threadIdx.x;
    const int C4 = C >> 2;
    const int c4 = tid % C4, prow = tid / C4, rows = T / C4;
    const int chunk = blockIdx.x, n = blockIdx.y;
    const int per = (HW + nchunks - 1) / nchunks;
    const int p0 = chunk * per;
    int p1 = p0 + per; if (p1 > HW) p1 = HW;
    double s[4] = { 0, 0, 0, 0 }, ss[4] = { 0, 0, 0, 0 };
    const float *base = x + ((long long)n * HW) * ld + 4 * c4;
    for (int p = p0 + prow; p < p1; p += rows) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(base + (long long)p * ld);
#pragma unroll
        for (int j = 0; j < 4; ++j) { const double d = (double)v[j]; s[j] += d; ss[j] += d * d; }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { sPart[tid * 8 + j] = s[j]; sPart[tid * 8 + 4 + j] = ss[j]; }
    __syncthreads();
    if (tid < G) {
        const int cpg = C / G;
        double a = 0.0, b = 0.0;
        for (int r = 0; r < rows; ++r)
            for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) {
                const int th = r * C4 + (c >> 2), sl = c & 3;
                a += sPart[th * 8 + sl];
                b += sPart[th * 8 + 4 + sl];
            }
        double *o = stats + (((long long)n * nchunks + chunk) * G + tid) * 2;
        o[0] = a; o[1] = b;
    }
}

// per-(image, channel) scale/shift from the fp64 partial sums, fixed summation order.  grid (B, slices), 256 threads: a workgroup
// finalises G / slices consecutive groups of one image.  slices > 1 for long entry lists only (the stem: conv2 leaves 2760 entries
// per image and group, 1.4 MB that ONE workgroup read in 41 us whatever the number of loads in flight - a single CU's miss queue -
// out of a single frame's 1.46 ms); the launcher derives it from the layer's geometry, never from the batch.
__global__ __launch_bounds__(256)
void gn_final_kernel(const double *__restrict__ stats, const float *__restrict__ gamma, const float *__restrict__ beta,
                     float *__restrict__ coeff, int HW, int C, int G, int nchunks, float eps, int statTile,
                     float *__restrict__ murs, int perTile)
{
    __shared__ double sG[2 * 64];
    const int n = blockIdx.x, tid = threadIdx.x;
    const int cpg = C / G;
    const int Gw = G / (int)gridDim.y, g0 = (int)blockIdx.y * Gw;     // this workgroup's groups
    int valid = nchunks;
    if (statTile > 0)          // statistics came from a conv epilogue: one entry per conv tile overlapping image n
        valid = (int)((((long long)(n + 1) * HW - 1) / statTile) - (((long long)n * HW) / statTile)) + 1;
    else if (statTile < 0)     // ... from a conv whose tiles start at image boundaries (batch-invariant plans)
        valid = (HW - statTile - 1) / -statTile;
    valid *= perTile;          // (the stride-2 stem convolutions write one entry per tile and row block of waves)
    // P = 256 / (G rounded up to a power of two) threads per group (8 for 32 groups, 128 for the 2 groups of conv1) each sum
    // a contiguous slice of the chunks, then the P slices are added as a fixed binary tree.  P depends on the layer only.
    __shared__ double sP[256 * 2];
    int gp = 1;
    while (gp < Gw) gp <<= 1;
    const int P = 256 / gp;
    {
        const int gl = tid / P, part = tid - gl * P, g = g0 + gl;
        double a = 0.0, b = 0.0;
        if (gl < Gw) {
            const int per = (valid + P - 1) / P;
            int k0 = part * per, k1 = k0 + per;
            if (k1 > valid) k1 = valid;
            // four interleaved partial sums (k % 4), added in order: the loads of four entries are in flight together - as
            // one dependent chain a slice of 19 entries (a single frame: 150 one-tile chunks) took ~8 us of L2 latency
            typedef double f64x2 __attribute__((ext_vector_type(2)));
            const f64x2 *st = reinterpret_cast<const f64x2 *>(stats) + ((long long)n * nchunks * G + g);
            double pa[4] = { 0.0, 0.0, 0.0, 0.0 }, pb[4] = { 0.0, 0.0, 0.0, 0.0 };
            int k = k0;
            for (; k + 4 <= k1; k += 4) {
                f64x2 v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = st[(long long)(k + e) * G];
#pragma unroll
                for (int e = 0; e < 4; ++e) { pa[e] += v[e][0]; pb[e] += v[e][1]; }
            }
            for (int e = 0; k < k1; ++k, ++e) { const f64x2 v = st[(long long)k * G]; pa[e] += v[0]; pb[e] += v[1]; }
            a = (pa[0] + pa[1]) + (pa[2] + pa[3]);
            b = (pb[0] + pb[1]) + (pb[2] + pb[3]);
        }
        sP[2 * tid] = a; sP[2 * tid + 1] = b;
        for (int s = P >> 1; s >= 1; s >>= 1) {
            __syncthreads();
            if (part < s) {
                a += sP[2 * (tid + s)]; b += sP[2 * (tid + s) + 1];
                sP[2 * tid] = a; sP[2 * tid + 1] = b;
            }
        }
    }
    __syncthreads();
    for (int g = tid; g < Gw; g += 256) {                              // (sG: indexed by the group's number within the workgroup)
        const double a = sP[2 * (g * P)], b = sP[2 * (g * P) + 1];
        const double cnt = (double)HW * (double)cpg;
        const double mean = a / cnt;
        double var = b / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        sG[2 * g] = mean; sG[2 * g + 1] = 1.0 / sqrt(var + (double)eps);
    }
    __syncthreads();
    for (int c = g0 * cpg + tid; c < (g0 + Gw) * cpg; c += 256) {
        const int g = c / cpg - g0;
        const double sc = (double)gamma[c] * sG[2 * g + 1];
        coeff[((long long)n * C + c) * 2] = (float)sc;
        coeff[((long long)n * C + c) * 2 + 1] = (float)((double)beta[c] - sG[2 * g] * sc);
        if (murs) {                                  // training plans: mean / rstd for the GroupNorm backward kernels
            murs[((long long)n * C + c) * 2] = (float)sG[2 * g];
            murs[((long long)n * C + c) * 2 + 1] = (float)sG[2 * g + 1];
        }
    }
}

// grid (achunks, B), 256 threads. v = x*scale + shift; flags as in crossloc_cnn.h
__global__ __launch_bounds__(256)
void gn_apply_kernel(const float *__restrict__ x, const double *__restrict__ stats, const float *__restrict__ gamma,
                     const float *__restrict__ beta, const float *__restrict__ aux, float *__restrict__ out,
                     int HW, int C, int ldIn, int ldOut, int ldAux, int G, int nchunks, float eps, int flags, int statTile,
                     const float *__restrict__ coeff)
{
    extern __shared__ __attribute__((aligned(16))) float sSS[];        // scale[C], shift[C]
    const int tid = threadIdx.x, n = blockIdx.y;
    const int cpg = C / G;
    if (coeff) {
        for (int c = tid; c < C; c += 256) {
            sSS[c] = coeff[((long long)n * C + c) * 2];
            sSS[C + c] = coeff[((long long)n * C + c) * 2 + 1];
        }
    } else {
    for (int c = tid; c < C; c += 256) {
        const int g = c / cpg;
        double a = 0.0, b = 0.0;
        const double *st = stats + ((long long)n * nchunks * G + g) * 2;
        int valid = nchunks;
        if (statTile > 0)      // statistics came from the conv epilogue: one entry per conv tile overlapping image n
            valid = (int)((((long long)(n + 1) * HW - 1) / statTile) - (((long long)n * HW) / statTile)) + 1;
        else if (statTile < 0) // ... tiles that start at image boundaries
            valid = (HW - statTile - 1) / -statTile;
        for (int k = 0; k < valid; ++k) { a += st[(long long)k * G * 2]; b += st[(long long)k * G * 2 + 1]; }
        const double cnt = (double)HW * (double)cpg;
        const double mean = a / cnt;
        double var = b / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        const double rstd = 1.0 / sqrt(var + (double)eps);
        const double sc = (double)gamma[c] * rstd;
        sSS[c] = (float)sc;
        sSS[C + c] = (float)((double)beta[c] - mean * sc);
    }
    }
    __syncthreads();
    const int C4 = C >> 2;
    const int achunks = gridDim.x;
    const int per = (HW + achunks - 1) / achunks;
    const int p0 = blockIdx.x * per;
    int p1 = p0 + per; if (p1 > HW) p1 = HW;
    const long long nElem4 = (long long)(p1 - p0) * C4;
    const bool reluIn = flags & XL_GN_RELU_IN, add = flags & XL_GN_ADD, reluOut = flags & XL_GN_RELU_OUT;
    if (C4 <= 256 && 256 % C4 == 0) {
        // the usual case (C = 32 ... 1024, a power of two): a thread keeps ONE channel quad - {scale, shift} in registers, no
        // 64-bit division per element - and walks the pixels, four per trip with all loads issued first (round 4).  Element by
        // element the same arithmetic as the general loop below.
        const int c = 4 * (tid % C4), rows = 256 / C4;
        const f32x4 sc = *reinterpret_cast<const f32x4 *>(sSS + c), sh = *reinterpret_cast<const f32x4 *>(sSS + C + c);
        auto one = [&](long long pix, f32x4 v, const f32x4 &r) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float t = fmaf(v[j], sc[j], sh[j]);
                if (reluIn) t = fmaxf(t, 0.f);
                if (add) t += r[j];
                if (reluOut) t = fmaxf(t, 0.f);
                v[j] = t;
            }
            *reinterpret_cast<f32x4 *>(out + pix * ldOut + c) = v;
        };
        const f32x4 zero = { 0.f, 0.f, 0.f, 0.f };
        int p = p0 + tid / C4;
        for (; p + 3 * rows < p1; p += 4 * rows) {
            f32x4 v[4], r[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const long long pix = (long long)n * HW + p + k * rows;
                v[k] = *reinterpret_cast<const f32x4 *>(x + pix * ldIn + c);
                r[k] = add ? *reinterpret_cast<const f32x4 *>(aux + pix * ldAux + c) : zero;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) one((long long)n * HW + p + k * rows, v[k], r[k]);
        }
        for (; p < p1; p += rows) {
            const long long pix = (long long)n * HW + p;
            one(pix, *reinterpret_cast<const f32x4 *>(x + pix * ldIn + c), add ? *reinterpret_cast<const f32x4 *>(aux + pix * ldAux + c) : zero);
        }
        return;
    }
    for (long long f = tid; f < nElem4; f += 256) {
        const int p = p0 + (int)(f / C4);
        const int c = (int)(f - (long long)(p - p0) * C4) * 4;
        const long long pix = (long long)n * HW + p;
        f32x4 v = *reinterpret_cast<const f32x4 *>(x + pix * ldIn + c);
        const f32x4 sc = *reinterpret_cast<const f32x4 *>(sSS + c);
        const f32x4 sh = *reinterpret_cast<const f32x4 *>(sSS + C + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float t = fmaf(v[j], sc[j], sh[j]);       // one rounding - the same arithmetic as the consumers that apply a
                                                      // deferred GroupNorm while loading (a frame must not depend on which form ran)
            if (reluIn) t = fmaxf(t, 0.f);
            v[j] = t;
        }
        if (add) {
            const f32x4 r = *reinterpret_cast<const f32x4 *>(aux + pix * ldAux + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += r[j];
        }
        if (reluOut) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
        }
        *reinterpret_cast<f32x4 *>(out + pix * ldOut + c) = v;
    }
}

// ---------------------------------------------------------------------------------------------- head

// in NHWC [B*HW][Cin], w [Cout][Cin], out NCHW [B][Cout][HW]; one wavefront per pixel (grid-stride)
template <int COUT_MAX>
__global__ __launch_bounds__(256)
void head_kernel(const float *__restrict__ in, const float *__restrict__ w, const float *__restrict__ bias,
                 const float *__restrict__ mean, float *__restrict__ out, int B, int HW, int Cin, int ldIn,
                 int Cout, int nTask, float lo, float hi, const float *__restrict__ coef, float normLo)
{
    // coef != NULL (512 -> <=4 form only): `in` is the raw output of the producing convolution and its GroupNorm (+ReLU)
    // is applied on load: x -> max(x*scale + shift, normLo) with {scale, shift} pairs [B][Cin][2] (XL_OP_GN_FINAL)
    const int lane = threadIdx.x & 63;
    const int waveGlobal = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int nWaves = (gridDim.x * 256) >> 6;
    const int nq = (Cin + 255) >> 8;                          // float4 per lane (256 channels per trip), <= 8
    const long long total = (long long)B * HW;
    if (Cin == 512 && Cout <= 4) {
        // the coordinate head (512 -> 4): the lane's 8 x 4 weights stay in registers across the pixel loop (re-reading
        // them per pixel made 8 of the 10 loads of a pixel weight loads), two pixels in flight per iteration
        f32x4 wr[4][2];
#pragma unroll
        for (int o = 0; o < 4; ++o)
#pragma unroll
            for (int q = 0; q < 2; ++q)
                wr[o][q] = o < Cout ? *reinterpret_cast<const f32x4 *>(w + (long long)o * Cin + q * 256 + lane * 4)
                                    : f32x4{ 0.f, 0.f, 0.f, 0.f };
        for (long long p0 = waveGlobal; p0 < total; p0 += 2LL * nWaves) {
            const long long p1 = p0 + nWaves;
            const bool two = p1 < total;
            const long long p1c = two ? p1 : p0;
            f32x4 v[2][2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                v[0][q] = *reinterpret_cast<const f32x4 *>(in + p0 * ldIn + q * 256 + lane * 4);
                v[1][q] = *reinterpret_cast<const f32x4 *>(in + p1c * ldIn + q * 256 + lane * 4);
            }
            if (coef) {
                const long long pp[2] = { p0, p1c };
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const float *cf = coef + ((pp[u] / HW) * Cin) * 2;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const f32x4 c0 = *reinterpret_cast<const f32x4 *>(cf + 2 * (q * 256 + lane * 4));
                        const f32x4 c1 = *reinterpret_cast<const f32x4 *>(cf + 2 * (q * 256 + lane * 4) + 4);
                        v[u][q].x = fmaxf(fmaf(v[u][q].x, c0[0], c0[1]), normLo);
                        v[u][q].y = fmaxf(fmaf(v[u][q].y, c0[2], c0[3]), normLo);
                        v[u][q].z = fmaxf(fmaf(v[u][q].z, c1[0], c1[1]), normLo);
                        v[u][q].w = fmaxf(fmaf(v[u][q].w, c1[2], c1[3]), normLo);
                    }
                }
            }
            float acc[2][4];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    float t = 0.f;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        t = fmaf(v[u][q].x, wr[o][q].x, t); t = fmaf(v[u][q].y, wr[o][q].y, t);
                        t = fmaf(v[u][q].z, wr[o][q].z, t); t = fmaf(v[u][q].w, wr[o][q].w, t);
                    }
                    acc[u][o] = t;
                }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int o = 0; o < 4; ++o) acc[u][o] += __shfl_xor(acc[u][o], off);
            if (lane < 2 * Cout) {
                const int u = lane / Cout, o = lane - u * Cout;
                if (u == 0 || two) {
                    float r = 0.f;
#pragma unroll
                    for (int uu = 0; uu < 2; ++uu)
#pragma unroll
                        for (int oo = 0; oo < 4; ++oo) if (uu == u && oo == o) r = acc[uu][oo];
                    r += bias[o];
                    if (o < nTask) r += mean[o];
                    else r = expf(fminf(fmaxf(r, lo), hi));
                    const long long p = u ? p1 : p0;
                    const int n = (int)(p / HW);
                    const int pix = (int)(p - (long long)n * HW);
                    out[((long long)n * Cout + o) * HW + pix] = r;
                }
            }
        }
        return;
    }
    for (long long p = waveGlobal; p < total; p += nWaves) {
        float acc[COUT_MAX];
#pragma unroll
        for (int o = 0; o < COUT_MAX; ++o) acc[o] = 0.f;
        for (int q = 0; q < nq; ++q) {
            const int c = q * 256 + lane * 4;
            if (c >= Cin) break;                                  // (Cin = 128 - the tiny network: lanes 32 .. 63 hold nothing)
            const f32x4 v = *reinterpret_cast<const f32x4 *>(in + p * ldIn + c);
#pragma unroll
            for (int o = 0; o < COUT_MAX; ++o) {
                if (o < Cout) {
                    const f32x4 wv = *reinterpret_cast<const f32x4 *>(w + (long long)o * Cin + c);
                    acc[o] = fmaf(v.x, wv.x, acc[o]); acc[o] = fmaf(v.y, wv.y, acc[o]);
                    acc[o] = fmaf(v.z, wv.z, acc[o]); acc[o] = fmaf(v.w, wv.w, acc[o]);
                }
            }
        }
#pragma unroll
        for (int o = 0; o < COUT_MAX; ++o)
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) acc[o] += __shfl_xor(acc[o], off);
        if (lane < Cout) {
            float v = 0.f;
#pragma unroll
            for (int o = 0; o < COUT_MAX; ++o) if (o == lane) v = acc[o];
            v += bias[lane];
            if (lane < nTask) v += mean[lane];                               // networks.py:351
            else v = expf(fminf(fmaxf(v, lo), hi));                          // networks.py:355-358
            const int n = (int)(p / HW);
            const int pix = (int)(p - (long long)n * HW);
            out[((long long)n * Cout + lane) * HW + pix] = v;
        }
    }
}

// Full-size (semantics) head, networks.py:259-273 + 344-358: the x8 pixel shuffle of the DUC activation is an index map
//   shuffled[b][c][8h+i][8w+j] = in[b][h][w][c*64 + i*8 + j]          (in: NHWC, GroupNorm + ReLU already applied)
// followed by F.interpolate(bilinear, align_corners=False) to H x W (the identity when H = 8 Hs and W = 8 Ws), the
// 1x1 fc3 over the C shuffled channels, the mean offset / exp(hardtanh) epilogue.  out NCHW [B][C][H][W]; one output
// pixel per thread (stores coalesced along x; the gathers of a wavefront fall into a few NHWC pixels).
template <int CMAX>
__global__ __launch_bounds__(256)
void duc_head_kernel(const float *__restrict__ in, const float *__restrict__ w, const float *__restrict__ bias,
                     const float *__restrict__ mean, float *__restrict__ out, int B, int Hs, int Ws, int C, int ldIn,
                     int H, int W, int nTask, float lo, float hi)
{
    const int Hu = 8 * Hs, Wu = 8 * Ws;
    const float sy = (float)Hu / (float)H, sx = (float)Wu / (float)W;       // area_pixel_compute_scale
    const bool ident = (Hu == H) && (Wu == W);
    const long long total = (long long)B * H * W;
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < total; p += (long long)gridDim.x * 256) {
        const int x = (int)(p % W);
        const int y = (int)((p / W) % H);
        const int n = (int)(p / ((long long)W * H));
        float fy = sy * ((float)y + 0.5f) - 0.5f; if (fy < 0.f) fy = 0.f;
        float fx = sx * ((float)x + 0.5f) - 0.5f; if (fx < 0.f) fx = 0.f;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < Hu - 1 ? 1 : 0), x1 = x0 + (x0 < Wu - 1 ? 1 : 0);
        const float ly1 = fy - (float)y0, lx1 = fx - (float)x0;
        const float ly0 = 1.f - ly1, lx0 = 1.f - lx1;
        const float *img = in + (long long)n * Hs * Ws * ldIn;
        const long long a00 = ((long long)(y0 >> 3) * Ws + (x0 >> 3)) * ldIn + (y0 & 7) * 8 + (x0 & 7);
        const long long a01 = ((long long)(y0 >> 3) * Ws + (x1 >> 3)) * ldIn + (y0 & 7) * 8 + (x1 & 7);
        const long long a10 = ((long long)(y1 >> 3) * Ws + (x0 >> 3)) * ldIn + (y1 & 7) * 8 + (x0 & 7);
        const long long a11 = ((long long)(y1 >> 3) * Ws + (x1 >> 3)) * ldIn + (y1 & 7) * 8 + (x1 & 7);
        float v[CMAX];
#pragma unroll
        for (int c = 0; c < CMAX; ++c) {
            v[c] = 0.f;
            if (c < C) {
                if (ident) v[c] = img[a00 + c * 64];
                else v[c] = ly0 * (lx0 * img[a00 + c * 64] + lx1 * img[a01 + c * 64])
                          + ly1 * (lx0 * img[a10 + c * 64] + lx1 * img[a11 + c * 64]);
            }
        }
#pragma unroll
        for (int o = 0; o < CMAX; ++o) {
            if (o < C) {
                float acc = bias[o];
#pragma unroll
                for (int c = 0; c < CMAX; ++c) if (c < C) acc = fmaf(w[o * C + c], v[c], acc);
                if (o < nTask) acc += mean[o];
                else acc = expf(fminf(fmaxf(acc, lo), hi));
                out[(((long long)n * C + o) * H + y) * W + x] = acc;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------- weight pack

__global__ void pack_weight_kernel(const float *__restrict__ src, float *__restrict__ dst, int Cout, int Cin, int k)
{
    const long long total = (long long)Cout * Cin * k * k;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        // dst index i = ((o*(Cin/32) + chunk)*k*k + tap)*32 + cl   (chunk-major, tap, channel-in-chunk)
        const int cl = (int)(i % 32);
        long long t = i / 32;
        const int tap = (int)(t % (k * k)); t /= (k * k);
        const int chunk = (int)(t % (Cin / 32));
        const int o = (int)(t / (Cin / 32));
        const int c = chunk * 32 + cl, ky = tap / k, kx = tap - ky * k;
        dst[i] = src[(((long long)o * Cin + c) * k + ky) * k + kx];
    }
}

// dgrad operand: dst[((c*(Cout/32) + ochunk)*k*k + tap)*32 + ol] = src[o][c][ky][kx], o = ochunk*32 + ol
__global__ void pack_weight_dgrad_kernel(const float *__restrict__ src, float *__restrict__ dst, int Cout, int Cin, int k)
{
    const long long total = (long long)Cout * Cin * k * k;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int ol = (int)(i % 32);
        long long t = i / 32;
        const int tap = (int)(t % (k * k)); t /= (k * k);
        const int ochunk = (int)(t % (Cout / 32));
        const int c = (int)(t / (Cout / 32));
        const int o = ochunk * 32 + ol, ky = tap / k, kx = tap - ky * k;
        dst[i] = src[(((long long)o * Cin + c) * k + ky) * k + kx];
    }
}

// ---------------------------------------------------------------------------------------------- host side

thread_local char g_err[256] = "";

// ---- optional per-op HIP-event timing (bench.py roofline leg): events are recorded on the launch stream
// around every op while profiling is on, and resolved after the caller has synchronised.
struct ProfRec { hipEvent_t a, b; int opIndex; int type; };
ProfRec *g_prof = nullptr;
int g_profCap = 0, g_profCount = 0;
bool g_profOn = false;
int g_profType = -1, g_profMinBatched = 0;     // record only ops of this type (-1: all) with nchunks2 >= the minimum

template <int KS, int STRIDE, int BN, int CIN, int MODE = 0, int BM = 128, int ZB = 0, int NORM = 0>
int launch_igemm(const xl_op &op, hipStream_t st, int py = 0, int px = 0)
{
    ConvArgs a;
    a.coef = nullptr; a.normLo = 0.f;
    if (NORM) {
        if (!op.aux2 || op.Ho * op.Wo < BM) return XL_ERR_ARG;
        a.coef = (const float *)op.aux2;
        a.normLo = (op.flags & XL_CONV_NORM_RELU) ? 0.f : -__builtin_huge_valf();
    }
    a.in = (const float *)op.in; a.w = (const float *)op.w; a.bias = (const float *)op.bias; a.out = (float *)op.out;
    a.B = op.B; a.Hi = op.Hi; a.Wi = op.Wi; a.Cin = op.Cin; a.Ho = op.Ho; a.Wo = op.Wo; a.Cout = op.Cout;
    a.ldIn = op.ld_in; a.ldOut = op.ld_out;
    a.M = op.B * op.Ho * op.Wo; a.K = op.ksize * op.ksize * op.Cin;
    a.py = py; a.px = px; a.Hj = 0; a.Wj = 0; a.ntaps = 0; a.tapList = 0;
    a.stats = nullptr; a.HW = op.Ho * op.Wo; a.G = 0; a.cpg = 1; a.nchunks = 0;
    a.zCount = 1; a.zIn = 0; a.zW = 0; a.zOut = 0;
    if (ZB) {                                        // batched GEMMs over consecutive blocks of in / w / out
        if (op.stats || op.nchunks2 < 1) return XL_ERR_ARG;
        a.zCount = op.nchunks2;
        a.zIn = (long long)a.M * op.ld_in; a.zW = (long long)op.Cout * a.K; a.zOut = (long long)a.M * op.ld_out;
    }
    if (MODE == 0 && op.stats && op.groups > 0) {
        a.G = op.groups; a.cpg = op.Cout / op.groups; a.nchunks = op.nchunks;
        if (a.HW < BM || op.Cout % op.groups != 0 || BN % a.cpg != 0 || a.nchunks < (a.HW + BM - 1) / BM + 1) return XL_ERR_ARG;
        if (a.cpg != 2 && a.cpg % 4 != 0) return XL_ERR_ARG;         // the statistics epilogue sums 2- or 4-channel pieces
        a.stats = (double *)op.stats;
    }
    if (MODE == 2) {
        // result pixel (iy,ix) = (2jy+py, 2jx+px); source row (iy+1-ky)/2 needs ky of parity (py+1)&1
        a.Hj = (op.Ho - py + 1) / 2; a.Wj = (op.Wo - px + 1) / 2;
        a.M = op.B * a.Hj * a.Wj;
        for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx)
                if (((py + 1 - ky) & 1) == 0 && ((px + 1 - kx) & 1) == 0) { a.tapList |= (unsigned)(ky * 3 + kx) << (4 * a.ntaps); ++a.ntaps; }
        if (a.M == 0) return XL_OK;
    }
    a.nbm = (a.M + BM - 1) / BM; a.nbn = (op.Cout + BN - 1) / BN;   // weight rows past Cout read as zero (bounds check)
    const long long inBytes = (((long long)op.B * op.Hi * op.Wi - 1) * op.ld_in + op.Cin) * 4;
    const long long wBytes = (long long)op.Cout * a.K * 4;
    const long long outBytes = (((long long)op.B * op.Ho * op.Wo - 1) * op.ld_out + op.Cout) * 4;
    if (wBytes >= 0x7fffffffLL) { snprintf(g_err, sizeof(g_err), "conv weights of %lld bytes", wBytes); return XL_ERR_ARG; }
    if (inBytes >= 0x7fffffffLL || outBytes >= 0x7fffffffLL) {
        // The kernel addresses a tensor with 32-bit byte offsets through one buffer descriptor (2 GiB).  Larger batches
        // run as several launches over image ranges, each with its own base pointers.  A range starts on a tile
        // boundary of the un-split launch (b0 * Ho*Wo divisible by BM), so every output tile - and every
        // (image, tile) slot of the fused GroupNorm statistics - is the same as in one launch.
        const long long perIn = (long long)op.Hi * op.Wi * op.ld_in * 4, perOut = (long long)op.Ho * op.Wo * op.ld_out * 4;
        const long long perMax = perIn > perOut ? perIn : perOut;
        int g = op.Ho * op.Wo;                                    // gcd(Ho*Wo, BM)
        for (int y = BM; y; ) { const int t = g % y; g = y; y = t; }
        const int align = BM / g;
        long long seg = (0x7ffffff0LL - 4LL * (op.Cin > op.Cout ? op.Cin : op.Cout)) / perMax;
        seg -= seg % align;
        if (ZB || MODE == 2 || seg < 1 || seg >= op.B) {
            snprintf(g_err, sizeof(g_err), "conv tensors of %lld / %lld bytes exceed 32-bit buffer addressing", inBytes, outBytes);
            return XL_ERR_ARG;
        }
        for (int b0 = 0; b0 < op.B; b0 += (int)seg) {
            xl_op part = op;
            part.B = (op.B - b0 < (int)seg) ? op.B - b0 : (int)seg;
            part.in = (const char *)op.in + (long long)b0 * perIn;
            part.out = (char *)op.out + (long long)b0 * perOut;
            if (op.stats) part.stats = (char *)op.stats + (long long)b0 * op.nchunks * op.groups * 2 * sizeof(double);
            if (NORM) part.aux2 = (const char *)op.aux2 + (long long)b0 * op.Cin * 2 * sizeof(float);
            const int rc = launch_igemm<KS, STRIDE, BN, CIN, MODE, BM, ZB, NORM>(part, st, py, px);
            if (rc != XL_OK) return rc;
        }
        return XL_OK;
    }
    a.inBytes = (unsigned)inBytes; a.wBytes = (unsigned)wBytes; a.outBytes = (unsigned)outBytes;
    a.accumulate = (op.flags & XL_CONV_ACCUMULATE) ? 1 : 0;
    if (op.ld_out % 4 != 0) return XL_ERR_ARG;                    // dwordx4 stores of 4 consecutive channels
    if ((ZB || MODE != 0) && (a.bias || a.stats)) return XL_ERR_ARG;
    if (a.accumulate && a.stats) return XL_ERR_ARG;
    size_t lds = sizeof(float) * 2 * (BM + BN) * kBK;
    if (NORM) lds += sizeof(float) * (4 * (size_t)op.Cin + 256);   // {scale, shift} of two images (+ slack for the unused tail stage)
    if (a.stats) {                                   // the statistics epilogue reuses the tile storage: make sure it fits
        const int pg = a.cpg >= 4 ? 4 : 2, np = (BN / 64) * 4 * (4 / pg);
        const size_t need = sizeof(float) * 256 * 2 * np * 2 + sizeof(double) * (BN / pg) * 2 * 2;
        if (need > lds) lds = need;
    }
    static XlLdsLimit configured;                    // one per template instantiation, tracked per device
    int cfgDev;
    if (configured.needs(lds, &cfgDev)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(igemm_conv_kernel<KS, STRIDE, BN, CIN, MODE, BM, ZB, NORM>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { snprintf(g_err, sizeof(g_err), "hipFuncSetAttribute: %s", hipGetErrorString(e)); return XL_ERR_HIP; }
        configured.done(lds, cfgDev);
    }
    static const bool clkDbg = getenv("XL_CONV_CLK") != nullptr;
    a.clk = nullptr;
    const int nwg = a.nbm * a.nbn * a.zCount;
    if (clkDbg && hipMalloc(&a.clk, sizeof(long long) * 8 * nwg) != hipSuccess) return XL_ERR_HIP;
    hipLaunchKernelGGL((igemm_conv_kernel<KS, STRIDE, BN, CIN, MODE, BM, ZB, NORM>), dim3(nwg), dim3(256), lds, st, a);
    if (clkDbg) {
        std::vector<long long> h(8 * (size_t)nwg);
        const bool copied = hipStreamSynchronize(st) == hipSuccess &&
                            hipMemcpy(h.data(), a.clk, sizeof(long long) * 8 * nwg, hipMemcpyDeviceToHost) == hipSuccess;
        (void)hipFree(a.clk);
        if (!copied) return XL_ERR_HIP;
        double pro = 0, loop = 0, epi = 0, tot = 0, wall = 0;
        long long wmin = h[1], wmax = h[5];
        std::map<long long, int> perCu;
        for (int i = 0; i < nwg; ++i) {
            const long long *c = &h[8 * (size_t)i];
            pro += c[2] - c[0]; loop += c[3] - c[2]; epi += c[4] - c[3]; tot += c[4] - c[0]; wall += c[5] - c[1];
            if (c[1] < wmin) wmin = c[1];
            if (c[5] > wmax) wmax = c[5];
            // HW_ID: cu_id bits 8-11, sh 12, se 13-15 ; XCC id low bits
            perCu[((c[7] & 15) << 16) | ((c[6] >> 8) & 0xff)]++;
        }
        int cmin = 1 << 30, cmax = 0;
        for (auto &kv : perCu) { if (kv.second < cmin) cmin = kv.second; if (kv.second > cmax) cmax = kv.second; }
        // residency: workgroup-lifetime per CU over that CU's active span (2.0 = both slots always occupied)
        std::map<long long, std::array<double, 3>> occ;               // busy, first start, last end (100 MHz wall ticks)
        for (int i = 0; i < nwg; ++i) {
            const long long *c = &h[8 * (size_t)i];
            auto &o = occ[((c[7] & 15) << 16) | ((c[6] >> 8) & 0xff)];
            if (o[0] == 0.0) { o[1] = (double)c[1]; o[2] = (double)c[5]; }
            o[0] += (double)(c[5] - c[1]);
            if ((double)c[1] < o[1]) o[1] = (double)c[1];
            if ((double)c[5] > o[2]) o[2] = (double)c[5];
        }
        double resid = 0.0;
        for (auto &kv : occ) resid += kv.second[0] / (kv.second[2] - kv.second[1]);
        fprintf(stderr, "[clk] mean resident workgroups per CU %.3f, mean lifetime %.2f us\n", resid / occ.size(), wall / nwg / 100.0);
        fprintf(stderr, "[clk] wgs %d nk %d: prologue %.0f loop %.0f (%.1f/step) epilogue %.0f total %.0f ticks; clock %.1f MHz; kernel span %.3f ms; CUs seen %zu, tiles per CU %d..%d\n",
                nwg, a.K / kBK, pro / nwg, loop / nwg, loop / nwg / (a.K / kBK), epi / nwg, tot / nwg, tot / wall * 100.0,
                (wmax - wmin) / 1e5, perCu.size(), cmin, cmax);
    }
    return XL_OK;
}

int run_conv(const xl_op &op, hipStream_t st)
{
    if (op.Cin % 32 != 0 || op.Cout % 32 != 0 || op.ld_in % 4 != 0) return XL_ERR_ARG;
    const bool wide = (op.Cout % 128 == 0);
    if (op.flags & XL_CONV_DGRAD) {
        if (op.ksize == 3 && op.stride == 1) return wide ? launch_igemm<3, 1, 128, 0, 1>(op, st) : launch_igemm<3, 1, 64, 0, 1>(op, st);
        if (op.ksize == 3 && op.stride == 2) {
            // four parity classes of result pixels, each with only the taps that reach it
            for (int py = 0; py < 2; ++py)
                for (int px = 0; px < 2; ++px) {
                    const int rc = wide ? launch_igemm<3, 2, 128, 0, 2>(op, st, py, px) : launch_igemm<3, 2, 64, 0, 2>(op, st, py, px);
                    if (rc != XL_OK) return rc;
                }
            return XL_OK;
        }
        if (op.ksize == 1 && op.stride == 1) return wide ? launch_igemm<1, 1, 128, 0, 1>(op, st) : launch_igemm<1, 1, 64, 0, 1>(op, st);
        return XL_ERR_UNSUPPORTED;
    }
    const bool small = (op.reserved_i == 64);          // 64-row tiles: the host asks for them when 128-row tiles
                                                        // would leave most of the 256 CUs idle (small batches)
#define XL_FWD(KS, S, BN, CIN, BM) launch_igemm<KS, S, BN, CIN, 0, BM>(op, st)
    if (op.ksize == 3 && op.stride == 1) {
        if (small) {
            if (wide && op.Cin == 512) return XL_FWD(3, 1, 128, 512, 64);
            return wide ? XL_FWD(3, 1, 128, 0, 64) : XL_FWD(3, 1, 64, 0, 64);
        }
        if (wide && op.Cin == 512) return XL_FWD(3, 1, 128, 512, 128);      // 78 % of the forward FLOPs
        return wide ? XL_FWD(3, 1, 128, 0, 128) : XL_FWD(3, 1, 64, 0, 128);
    }
    if (op.ksize == 3 && op.stride == 2) {
        if (op.flags & XL_CONV_PAIR_F16) return xl_run_pair_stem(op, st);
        if (op.flags & XL_CONV_SPLIT_BF16) return xl_run_split_stem(op, st);
        if (small) return wide ? XL_FWD(3, 2, 128, 0, 64) : XL_FWD(3, 2, 64, 0, 64);
        return wide ? XL_FWD(3, 2, 128, 0, 128) : XL_FWD(3, 2, 64, 0, 128);
    }
    if (op.flags & XL_CONV_SPLIT_BF16) return xl_run_split_gemm(op, st);
    if (op.ksize == 1 && op.stride == 1 && op.nchunks2 > 1) {         // Winograd: nchunks2 GEMMs in one launch
        if (!wide) return XL_ERR_UNSUPPORTED;
        if (small) return op.Cin == 512 ? launch_igemm<1, 1, 128, 512, 0, 64, 1>(op, st) : launch_igemm<1, 1, 128, 0, 0, 64, 1>(op, st);
        return op.Cin == 512 ? launch_igemm<1, 1, 128, 512, 0, 128, 1>(op, st) : launch_igemm<1, 1, 128, 0, 0, 128, 1>(op, st);
    }
    if (op.ksize == 1 && op.stride == 1 && (op.flags & XL_CONV_NORM_IN)) {   // producer's GroupNorm applied on load
        if (!wide || small) return XL_ERR_UNSUPPORTED;
        return op.Cin == 512 ? launch_igemm<1, 1, 128, 512, 0, 128, 0, 1>(op, st) : launch_igemm<1, 1, 128, 0, 0, 128, 0, 1>(op, st);
    }
    if (op.ksize == 1 && op.stride == 1) {
        if (small) {
            if (wide && op.Cin == 512) return XL_FWD(1, 1, 128, 512, 64);
            return wide ? XL_FWD(1, 1, 128, 0, 64) : XL_FWD(1, 1, 64, 0, 64);
        }
        if (wide && op.Cin == 512) return XL_FWD(1, 1, 128, 512, 128);
        return wide ? XL_FWD(1, 1, 128, 0, 128) : XL_FWD(1, 1, 64, 0, 128);
    }
#undef XL_FWD
    return XL_ERR_UNSUPPORTED;
}

int run_op(const xl_op &op, hipStream_t st)
{
    switch (op.type) {
        case XL_OP_CONV1: {
            if (op.stats || op.aux2) {
                // inference form: statistics-only pass (stats, nchunks workgroups per image, reserved_i pixels/256 each)
                // or conv + deferred GroupNorm (aux2 = {scale, shift} pairs) + ReLU (flags & XL_GN_RELU_IN)
                if (op.Cin == 3 && op.Cout == 32 && op.ld_out == 32 && op.reserved_i == 0) {
                    // matrix-pipe form: one workgroup per 16 x 64 output tile, nchunks = tiles per image; `w` = the weight
                    // fragments [3 planes][3 rows of the window][64 lanes][8] bf16 (crossloc_amd/networks.py, pack_conv1_split)
                    const int tilesX = (op.Wi + kC1TW - 1) / kC1TW, tilesY = (op.Hi + kC1TH - 1) / kC1TH;
                    if (op.nchunks != tilesX * tilesY || (op.stats && op.groups != 32)) return XL_ERR_ARG;
                    if (op.stats && op.out)
                        hipLaunchKernelGGL(conv1_mfma_kernel<2>, dim3(op.nchunks, op.B), dim3(256), 0, st, (const float *)op.in,
                                           (const c1_u32x4 *)op.w, (const float *)op.bias, (const float *)nullptr, (float *)op.out,
                                           (double *)op.stats, op.Hi, op.Wi, tilesX, 0);
                    else if (op.stats)
                        hipLaunchKernelGGL(conv1_mfma_kernel<0>, dim3(op.nchunks, op.B), dim3(256), 0, st, (const float *)op.in,
                                           (const c1_u32x4 *)op.w, (const float *)op.bias, (const float *)nullptr, (float *)nullptr,
                                           (double *)op.stats, op.Hi, op.Wi, tilesX, 0);
                    else
                        hipLaunchKernelGGL(conv1_mfma_kernel<1>, dim3(op.nchunks, op.B), dim3(256), 0, st, (const float *)op.in,
                                           (const c1_u32x4 *)op.w, (const float *)op.bias, (const float *)op.aux2, (float *)op.out,
                                           (double *)nullptr, op.Hi, op.Wi, tilesX, (op.flags & XL_GN_RELU_IN) ? 1 : 0);
                    return XL_OK;
                }
                if (op.Cin != 3 || op.Cout != 32 || op.ld_out != 32 || op.reserved_i < 1 ||
                    (long long)op.nchunks * op.reserved_i * 256 < (long long)op.Hi * op.Wi || (op.stats && op.groups != 32))
                    return XL_ERR_ARG;
                if (op.stats)
                    hipLaunchKernelGGL(conv1_fused_kernel<0>, dim3(op.nchunks, op.B), dim3(256), 0, st, (const float *)op.in,
                                       (const float *)op.w, (const float *)op.bias, (const float *)nullptr, (float *)nullptr,
                                       (double *)op.stats, op.Hi, op.Wi, op.ld_out, op.reserved_i, 0);
                else
                    hipLaunchKernelGGL(conv1_fused_kernel<1>, dim3(op.nchunks, op.B), dim3(256), 0, st, (const float *)op.in,
                                       (const float *)op.w, (const float *)op.bias, (const float *)op.aux2, (float *)op.out,
                                       (double *)nullptr, op.Hi, op.Wi, op.ld_out, op.reserved_i,
                                       (op.flags & XL_GN_RELU_IN) ? 1 : 0);
                return XL_OK;
            }
            if (op.Cout % 8 != 0 || op.Cout > 64 || 256 % (op.Cout / 8) != 0 || op.ld_out % 4 != 0) return XL_ERR_ARG;
            const size_t lds = sizeof(float) * (size_t)(9 * op.Cin * op.Cout + op.Cout);
            const long long pix = (long long)op.B * op.Hi * op.Wi;
            const int ppb = 256 / (op.Cout / 8);
            long long blocks = (pix + ppb - 1) / ppb;
            if (blocks > 65536) blocks = 65536;
            hipLaunchKernelGGL(conv1_direct_kernel, dim3((unsigned)blocks), dim3(256), lds, st,
                               (const float *)op.in, (const float *)op.w, (const float *)op.bias, (float *)op.out,
                               op.B, op.Cin, op.Hi, op.Wi, op.Cout, op.ld_out);
            return XL_OK;
        }
        case XL_OP_CONV:
            return run_conv(op, st);
        case XL_OP_STEM12:
            return xl_run_stem12(op, st);
        case XL_OP_S2_DGRAD:
            return xl_run_s2_dgrad(op, st);
        case XL_OP_WINO_IN: {
            if (op.ksize == 6) {                    // F(6x6,3x3): Ho x Wo tiles of 6x6 outputs, partial tiles allowed
                if (op.Cin % 4 != 0 || op.ld_in % 2 != 0 || op.Ho != (op.Hi + 5) / 6 || op.Wo != (op.Wi + 5) / 6) return XL_ERR_ARG;
                const long long items6 = (long long)op.B * op.Ho * op.Wo * (op.Cin / 2);
                long long blocks6 = (items6 + 255) / 256;
                if (blocks6 > 262144) blocks6 = 262144;
                auto kin = !op.aux2 ? wino6_in_kernel<0> : (op.flags & XL_GN_RELU_IN) ? wino6_in_kernel<2> : wino6_in_kernel<1>;
                if (op.flags & XL_CONV_SPLIT_BF16)
                    kin = !op.aux2 ? wino6_in_kernel<0, 1> : (op.flags & XL_GN_RELU_IN) ? wino6_in_kernel<2, 1> : wino6_in_kernel<1, 1>;
                if ((op.flags & XL_CONV_SPLIT_BF16) && (op.flags & XL_CONV_SPLIT_IL)) {
                    if (op.Cin % 128 != 0) return XL_ERR_ARG;          // a wave = 128 consecutive channels of one tile
                    kin = !op.aux2 ? wino6_in_kernel<0, 2> : (op.flags & XL_GN_RELU_IN) ? wino6_in_kernel<2, 2> : wino6_in_kernel<1, 2>;
                }
                WinoFold fold = { nullptr, nullptr, 0, 0, 0, nullptr };
                if (op.out2) {
                    // fold form: aux2 = coefficients, out2 = the materialised activation (pixel stride ld_out), aux = residual
                    // (pixel stride ld_aux) when XL_GN_ADD is set; flags & (XL_GN_RELU_IN | XL_GN_ADD | XL_GN_RELU_OUT);
                    // w (optional) = {scale, shift} pairs of the residual's own deferred GroupNorm + ReLU
                    if (!op.aux2 || op.ld_out % 2 != 0 || ((op.flags & XL_GN_ADD) && (!op.aux || op.ld_aux % 2 != 0)) ||
                        op.out2 == op.in || op.out2 == op.aux) return XL_ERR_ARG;
                    fold.res = (op.flags & XL_GN_ADD) ? (const float *)op.aux : nullptr;
                    fold.side = (float *)op.out2; fold.ldRes = op.ld_aux; fold.ldSide = op.ld_out;
                    fold.flags = op.flags & (XL_GN_RELU_IN | XL_GN_RELU_OUT);
                    fold.resCoef = (op.flags & XL_GN_ADD) ? (const float *)op.w : nullptr;
                    if ((op.flags & XL_CONV_SPLIT_BF16) && !(op.flags & XL_CONV_PAIR_F16)) return XL_ERR_UNSUPPORTED;   // fp32 or fp16 pairs
                    kin = (op.flags & XL_CONV_PAIR_F16) ? wino6_in_kernel<1, 3, 1> : wino6_in_kernel<1, 0, 1>;
                } else if (op.flags & XL_CONV_PAIR_F16)
                    kin = !op.aux2 ? wino6_in_kernel<0, 3> : (op.flags & XL_GN_RELU_IN) ? wino6_in_kernel<2, 3> : wino6_in_kernel<1, 3>;
                if ((op.flags & XL_CONV_PAIR_F16) && (!op.scale || op.Cin % 16 != 0)) return XL_ERR_ARG;       // (a quad = 8 channels of one tile)
                hipLaunchKernelGGL(kin, dim3((unsigned)blocks6), dim3(256), 0, st, (const float *)op.in,
                                   (float *)op.out, op.B, op.Hi, op.Wi, op.Cin, op.ld_in, op.Ho, op.Wo,
                                   (const float *)op.aux2, fold, (const float *)op.scale);
                return XL_OK;
            }
            if (op.ksize == 4) {                    // F(4x4,3x3): Ho x Wo tiles of 4x4 outputs, partial tiles allowed
                if (op.Cin % 4 != 0 || op.ld_in % 2 != 0 || op.Ho != (op.Hi + 3) / 4 || op.Wo != (op.Wi + 3) / 4) return XL_ERR_ARG;
                const long long items4 = (long long)op.B * op.Ho * op.Wo * (op.Cin / 2);
                long long blocks4 = (items4 + 255) / 256;
                if (blocks4 > 262144) blocks4 = 262144;
                auto kin = !op.aux2 ? wino4_in_kernel<0> : (op.flags & XL_GN_RELU_IN) ? wino4_in_kernel<2> : wino4_in_kernel<1>;
                hipLaunchKernelGGL(kin, dim3((unsigned)blocks4), dim3(256), 0, st, (const float *)op.in,
                                   (float *)op.out, op.B, op.Hi, op.Wi, op.Cin, op.ld_in, op.Ho, op.Wo,
                                   (const float *)op.aux2);
                return XL_OK;
            }
            if (op.aux2) return XL_ERR_ARG;                      // the deferred GroupNorm exists for F(4x4,3x3) only
            if (op.Cin % 4 != 0 || op.ld_in % 4 != 0 || op.Hi != 2 * op.Ho || op.Wi != 2 * op.Wo) return XL_ERR_ARG;
            const long long items = (long long)op.B * op.Ho * op.Wo * (op.Cin / 4);
            long long blocks = (items + 255) / 256;
            if (blocks > 262144) blocks = 262144;
            hipLaunchKernelGGL(wino_in_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const float *)op.in, (float *)op.out,
                               op.B, op.Hi, op.Wi, op.Cin, op.ld_in, op.Ho, op.Wo);
            return XL_OK;
        }
        case XL_OP_WINO_OUT: {
            // in: M [16][B*Th*Tw][C]; out [B,H,W,C] (Hi x Wi = output size); reserved_i = tiles per block
            if (op.ksize == 6) {
                const int Th6 = (op.Hi + 5) / 6, Tw6 = (op.Wi + 5) / 6;
                // two channels per lane where the channel count allows (3.7 -> 3.3 ms per 44-frame step); XL_WINO_OUT_VW=1: one
                static const int vw = getenv("XL_WINO_OUT_VW") ? atoi(getenv("XL_WINO_OUT_VW")) : 2;
                const bool two = vw == 2 && op.Cin % 512 == 0 && 64LL * op.B * Th6 * Tw6 * op.Cin * 4 < 0xffffffffLL;
                const int CB = two ? 512 : (op.Cin < 256 ? op.Cin : 256);
                if (op.Cin % 2 != 0 || op.Cin % CB != 0 || (256 * (two ? 2 : 1)) % CB != 0 || op.ld_out % 2 != 0 || op.reserved_i < 1 ||
                    op.nchunks != (Th6 * Tw6 + op.reserved_i - 1) / op.reserved_i)
                    return XL_ERR_ARG;
                if (op.stats && (op.groups < 1 || op.Cin % op.groups != 0 || CB % (op.Cin / op.groups) != 0 ||
                                 CB / (op.Cin / op.groups) > 256))
                    return XL_ERR_ARG;
                // round 3: M staged through LDS by DMA, one wave per (image, chunk, 128-channel slice).  Measured at 47 frames:
                // 256 channels 0.155 vs 0.174 ms for the register form with one channel per lane; 512 channels 0.29 vs 0.28 ms
                // for the two-channels-per-lane register form (both ~5.1 TB/s: bytes in flight were not what limits this
                // pass) - so the DMA form takes the layers the 8-byte register form does not cover.  XL_WINO_OUT_DMA=1: all.
                static const bool noDma = getenv("XL_WINO_OUT_NO_DMA") != nullptr;
                static const bool allDma = getenv("XL_WINO_OUT_DMA") != nullptr;
                const int cpg6 = op.groups > 0 ? op.Cin / op.groups : 2;
                const long long outB = ((long long)op.B * op.Hi * op.Wi - 1) * op.ld_out * 4 + (long long)op.Cin * 4;
                if (!noDma && (allDma || !two) && !(op.flags & XL_CONV_ACCUMULATE) && op.Cin % 128 == 0 && 64LL * op.B * Th6 * Tw6 * op.Cin * 4 < 0x7fffffffLL &&
                    outB < 0x7fffffffLL && (!op.stats || (cpg6 >= 2 && cpg6 <= 128 && (cpg6 & (cpg6 - 1)) == 0 && op.Cin % op.groups == 0))) {
                    static XlLdsLimit configuredDma;
                    int cfgDev;
                    if (configuredDma.needs(131072, &cfgDev)) {
                        if (hipFuncSetAttribute(reinterpret_cast<const void *>(wino6_out_dma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                131072) != hipSuccess) return XL_ERR_HIP;
                        configuredDma.done(131072, cfgDev);
                    }
                    const long long units = (long long)op.B * op.nchunks * (op.Cin / 128);
                    hipLaunchKernelGGL(wino6_out_dma_kernel, dim3((unsigned)((units + 3) / 4)), dim3(256), 131072, st, (const float *)op.in,
                                       (const float *)op.bias, (float *)op.out, (double *)op.stats, op.B, op.Hi, op.Wi, op.Cin,
                                       op.ld_out, Th6, Tw6, op.reserved_i, op.groups, op.nchunks, units,
                                       (op.flags & XL_CONV_M_TILE_MAJOR) ? 1 : 0);
                    return XL_OK;
                }
                hipLaunchKernelGGL(two ? wino6_out_kernel<2> : wino6_out_kernel<1>, dim3(op.nchunks, op.B, op.Cin / CB), dim3(256), 0, st, (const float *)op.in,
                                   (const float *)op.bias, (float *)op.out, (double *)op.stats, op.B, op.Hi, op.Wi, op.Cin,
                                   op.ld_out, Th6, Tw6, op.reserved_i, op.groups, op.nchunks,
                                   (op.flags & XL_CONV_ACCUMULATE) ? 1 : 0, (op.flags & XL_CONV_M_TILE_MAJOR) ? 1 : 0);
                return XL_OK;
            }
            if (op.ksize == 4) {
                if (op.flags & XL_CONV_M_TILE_MAJOR) return XL_ERR_UNSUPPORTED;
                const int Th4 = (op.Hi + 3) / 4, Tw4 = (op.Wi + 3) / 4;
                const int CB = op.Cin < 512 ? op.Cin : 512;
                if (op.Cin % 2 != 0 || op.Cin % CB != 0 || 256 % (CB / 2) != 0 || op.ld_out % 2 != 0 || op.reserved_i < 1 ||
                    op.nchunks != (Th4 * Tw4 + op.reserved_i - 1) / op.reserved_i)
                    return XL_ERR_ARG;
                if (op.stats && (op.groups < 1 || op.Cin % op.groups != 0 || CB % (op.Cin / op.groups) != 0 ||
                                 CB / (op.Cin / op.groups) > 256))
                    return XL_ERR_ARG;
                hipLaunchKernelGGL(wino4_out_kernel, dim3(op.nchunks, op.B, op.Cin / CB), dim3(256), 0, st, (const float *)op.in,
                                   (const float *)op.bias, (float *)op.out, (double *)op.stats, op.B, op.Hi, op.Wi, op.Cin,
                                   op.ld_out, Th4, Tw4, op.reserved_i, op.groups, op.nchunks,
                                   (op.flags & XL_CONV_ACCUMULATE) ? 1 : 0);
                return XL_OK;
            }
            const int C4 = op.Cin / 4;
            if (op.Cin % 4 != 0 || C4 > 256 || 256 % C4 != 0 || op.ld_out % 4 != 0 || op.reserved_i < 1) return XL_ERR_ARG;
            const int Th = op.Hi / 2, Tw = op.Wi / 2;
            if (op.Hi != 2 * Th || op.Wi != 2 * Tw || op.nchunks != (Th * Tw + op.reserved_i - 1) / op.reserved_i) return XL_ERR_ARG;
            if (op.stats && (op.groups < 1 || op.groups > 256 || op.Cin % op.groups != 0)) return XL_ERR_ARG;
            hipLaunchKernelGGL(wino_out_kernel, dim3(op.nchunks, op.B), dim3(256), 0, st, (const float *)op.in,
                               (const float *)op.bias, (float *)op.out, (double *)op.stats, op.B, op.Hi, op.Wi, op.Cin,
                               op.ld_out, Th, Tw, op.reserved_i, op.groups, op.nchunks);
            return XL_OK;
        }
        case XL_OP_GN_STATS: {
            const int C4 = op.Cin / 4;
            if (op.Cin % 4 != 0 || op.Cin % op.groups != 0 || op.ld_in % 4 != 0) return XL_ERR_ARG;
            int T = 256;
            if (C4 > 256) T = C4;
            else if (256 % C4 != 0) {                        // e.g. 384 channels: 96 quads -> 192 threads (lcm with 64)
                T = C4;
                while (T % 64 != 0) T += C4;
            }
            if (T > 1024 || T % 64 != 0 || op.groups > T) return XL_ERR_ARG;
            hipLaunchKernelGGL(gn_stats_kernel, dim3(op.nchunks, op.B), dim3(T), sizeof(double) * 8 * T, st,
                               (const float *)op.in, (double *)op.stats, op.Hi * op.Wi, op.Cin, op.ld_in, op.groups,
                               op.nchunks);
            return XL_OK;
        }
        case XL_OP_GN_APPLY: {
            if (op.Cin % 4 != 0 || op.ld_in % 4 != 0 || op.ld_out % 4 != 0) return XL_ERR_ARG;
            const int HW = op.Hi * op.Wi;
            int achunks = (HW * (op.Cin / 4) + 256 * 16 - 1) / (256 * 16);     // ~16 float4 per thread
            if (achunks < 1) achunks = 1;
            if (achunks > 1024) achunks = 1024;
            hipLaunchKernelGGL(gn_apply_kernel, dim3(achunks, op.B), dim3(256), sizeof(float) * 2 * op.Cin, st,
                               (const float *)op.in, (const double *)op.stats, (const float *)op.w,
                               (const float *)op.bias, (const float *)op.aux, (float *)op.out, HW, op.Cin, op.ld_in,
                               op.ld_out, op.ld_aux, op.groups, op.nchunks, op.eps, op.flags, op.reserved_i, (const float *)op.aux2);
            return XL_OK;
        }
        case XL_OP_GN_FINAL: {
            if (op.groups > 32 || op.Cin % op.groups != 0) return XL_ERR_ARG;
            if (op.reserved_i != 0) {
                // one entry (x stride) per producer tile overlapping an image: they must fit the nchunks slots the producer wrote,
                // or the finalisation sums the next image's entries / stale memory
                const long long rows = op.reserved_i < 0 ? -op.reserved_i : op.reserved_i, hw = (long long)op.Hi * op.Wi;
                const long long perTile = op.stride > 1 ? op.stride : 1;
                const long long tiles = op.reserved_i < 0 ? (hw + rows - 1) / rows : (hw + rows - 1) / rows + 1;
                if (tiles * perTile > op.nchunks) return XL_ERR_ARG;
            }
            // entries per image and group (an upper estimate from the layer's geometry): long lists are split over workgroups by
            // groups - 4 slices above 256 entries, 8 above 1024 (the stem; every other layer: one workgroup per image, as ever)
            long long entries = op.nchunks;
            if (op.reserved_i != 0) {
                const long long rows = op.reserved_i < 0 ? -op.reserved_i : op.reserved_i;
                entries = (((long long)op.Hi * op.Wi + rows - 1) / rows + 1) * (op.stride > 1 ? op.stride : 1);
            }
            int slices = entries > 1024 ? 8 : entries > 256 ? 4 : 1;
            static const char *noSlices = getenv("XL_GN_FINAL_ONE_WG");
            if (op.groups % slices != 0 || noSlices) slices = 1;
            hipLaunchKernelGGL(gn_final_kernel, dim3(op.B, slices), dim3(256), 0, st, (const double *)op.stats, (const float *)op.w,
                               (const float *)op.bias, (float *)op.out, op.Hi * op.Wi, op.Cin, op.groups, op.nchunks, op.eps,
                               op.reserved_i, (float *)op.out2, op.reserved_i != 0 && op.stride > 1 ? op.stride : 1);
            return XL_OK;
        }
        case XL_OP_HEAD: {
            if (op.Cin % 4 != 0 || op.Cin < 4 || op.Cin > 2048 || op.Cout > 8 || op.Cout < 1 || op.ld_in % 4 != 0) return XL_ERR_ARG;
            const long long pix = (long long)op.B * op.Hi * op.Wi;
            long long blocks = (pix + 3) / 4;
            if (blocks > 4096) blocks = 4096;
            if (op.aux2 && !(op.Cin == 512 && op.Cout <= 4)) return XL_ERR_UNSUPPORTED;    // normalise-on-load: 512 -> <=4 form
            hipLaunchKernelGGL(head_kernel<8>, dim3((unsigned)blocks), dim3(256), 0, st, (const float *)op.in,
                               (const float *)op.w, (const float *)op.bias, (const float *)op.aux, (float *)op.out,
                               op.B, op.Hi * op.Wi, op.Cin, op.ld_in, op.Cout, op.n_task, op.clamp_lo, op.clamp_hi,
                               (const float *)op.aux2, (op.flags & XL_CONV_NORM_RELU) ? 0.f : -__builtin_huge_valf());
            return XL_OK;
        }
        case XL_OP_DUC_HEAD: {
            // in [B,Hi,Wi,Cout*64] NHWC, out [B,Cout,Ho,Wo] NCHW
            if (op.Cout < 1 || op.Cout > 8 || op.Cin != op.Cout * 64 || op.Ho < 1 || op.Wo < 1) return XL_ERR_ARG;
            const long long pix = (long long)op.B * op.Ho * op.Wo;
            long long blocks = (pix + 255) / 256;
            if (blocks > 65536) blocks = 65536;
            hipLaunchKernelGGL(duc_head_kernel<8>, dim3((unsigned)blocks), dim3(256), 0, st, (const float *)op.in,
                               (const float *)op.w, (const float *)op.bias, (const float *)op.aux, (float *)op.out,
                               op.B, op.Hi, op.Wi, op.Cout, op.ld_in, op.Ho, op.Wo, op.n_task, op.clamp_lo, op.clamp_hi);
            return XL_OK;
        }
        default:
            return xl_run_bwd_op(op, st);
    }
}

}  // namespace

extern "C" {

// which op of a list a launcher refused (the launchers return a bare status): read back through xl_cnn_last_error()
static void note_failed_op(const xl_op &op, int index, int rc)
{
    (void)rc;
    char why[160];
    snprintf(why, sizeof(why), "%s", g_err);                              // (a launcher's own text - cleared at the head of the list - stays in front)
    snprintf(g_err, sizeof(g_err), "%s%sop %d refused (type %d, k%d s%d, %d -> %d channels, in %dx%d out %dx%d, B %d, ld %d/%d, flags 0x%x, Z %d, form %d)",
             why, why[0] ? "; " : "", index, op.type, op.ksize, op.stride, op.Cin, op.Cout, op.Hi, op.Wi, op.Ho, op.Wo, op.B, op.ld_in, op.ld_out,
             (unsigned)op.flags, op.nchunks2, op.reserved_i);
}

int xl_cnn_run(const xl_op *ops, int n_ops, void *stream)
{
    if (!ops || n_ops < 0) return XL_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    g_err[0] = 0;
    for (int i = 0; i < n_ops; ++i) {
        const bool rec = g_profOn && g_profCount < g_profCap && (g_profType < 0 || ops[i].type == g_profType) &&
                         ops[i].nchunks2 >= g_profMinBatched;
        if (rec) (void)hipEventRecord(g_prof[g_profCount].a, st);
        const int rc = run_op(ops[i], st);
        if (rc != XL_OK) { note_failed_op(ops[i], i, rc); return rc; }
        if (rec) {
            (void)hipEventRecord(g_prof[g_profCount].b, st);
            g_prof[g_profCount].opIndex = i; g_prof[g_profCount].type = ops[i].type;
            ++g_profCount;
        }
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_err, sizeof(g_err), "kernel launch: %s", hipGetErrorString(e)); return XL_ERR_HIP; }
    return XL_OK;
}

int xl_cnn_op_size(void) { return (int)sizeof(xl_op); }

// ---- the op list of a plan as ONE executable HIP graph (latency path: a single frame is ~95 short launches, and the host
// side of an eager launch costs about as much as the shortest kernels run)
int xl_cnn_graph_capture(const xl_op *ops, int n_ops, void *stream, void **graph_out)
{
    if (!ops || n_ops < 1 || !stream || !graph_out) return XL_ERR_ARG;     // (the NULL stream cannot be captured)
    hipStream_t st = (hipStream_t)stream;
    const bool prof = g_profOn;
    g_profOn = false;                                                      // event records are not part of a graph
    if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) != hipSuccess) { g_profOn = prof; return XL_ERR_HIP; }
    g_err[0] = 0;
    int rc = XL_OK;
    for (int i = 0; i < n_ops && rc == XL_OK; ++i) {
        rc = run_op(ops[i], st);
        if (rc != XL_OK) note_failed_op(ops[i], i, rc);
    }
    hipGraph_t g = nullptr;
    const hipError_t e = hipStreamEndCapture(st, &g);
    g_profOn = prof;
    if (rc != XL_OK || e != hipSuccess || !g) {
        if (g) (void)hipGraphDestroy(g);
        (void)hipGetLastError();
        if (rc == XL_OK) snprintf(g_err, sizeof(g_err), "hipStreamEndCapture: %s", hipGetErrorString(e));
        return rc != XL_OK ? rc : XL_ERR_HIP;
    }
    hipGraphExec_t ex = nullptr;
    const hipError_t ei = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (ei != hipSuccess || !ex) { snprintf(g_err, sizeof(g_err), "hipGraphInstantiate: %s", hipGetErrorString(ei)); return XL_ERR_HIP; }
    *graph_out = (void *)ex;
    return XL_OK;
}

int xl_cnn_graph_launch(void *graph, void *stream)
{
    if (!graph) return XL_ERR_ARG;
    if (g_profOn) return XL_ERR_UNSUPPORTED;                               // per-op timing needs the eager path
    const hipError_t e = hipGraphLaunch((hipGraphExec_t)graph, (hipStream_t)stream);
    if (e != hipSuccess) { snprintf(g_err, sizeof(g_err), "hipGraphLaunch: %s", hipGetErrorString(e)); return XL_ERR_HIP; }
    return XL_OK;
}

int xl_cnn_graph_destroy(void *graph)
{
    if (!graph) return XL_OK;
    return hipGraphExecDestroy((hipGraphExec_t)graph) == hipSuccess ? XL_OK : XL_ERR_HIP;
}

int xl_cnn_pack_conv_weight(const float *w_oihw_dev, float *w_ohwi_dev, int Cout, int Cin, int k, void *stream)
{
    if (!w_oihw_dev || !w_ohwi_dev || Cout <= 0 || Cin <= 0 || k <= 0) return XL_ERR_ARG;
    const long long total = (long long)Cout * Cin * k * k;
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w_oihw_dev,
                       w_ohwi_dev, Cout, Cin, k);
    return hipGetLastError() == hipSuccess ? XL_OK : XL_ERR_HIP;
}

int xl_cnn_pack_conv_weight_dgrad(const float *w_oihw_dev, float *w_dgrad_dev, int Cout, int Cin, int k, void *stream)
{
    if (!w_oihw_dev || !w_dgrad_dev || Cout <= 0 || Cin <= 0 || k <= 0 || Cout % 32 != 0) return XL_ERR_ARG;
    const long long total = (long long)Cout * Cin * k * k;
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_weight_dgrad_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w_oihw_dev,
                       w_dgrad_dev, Cout, Cin, k);
    return hipGetLastError() == hipSuccess ? XL_OK : XL_ERR_HIP;
}

const char *xl_cnn_last_error(void) { return g_err; }

int xl_cnn_prof_begin(int max_records)
{
    if (max_records <= 0) return XL_ERR_ARG;
    if (g_prof) return XL_ERR_ARG;
    g_prof = (ProfRec *)calloc((size_t)max_records, sizeof(ProfRec));
    if (!g_prof) return XL_ERR_ARG;
    for (int i = 0; i < max_records; ++i) {
        if (hipEventCreate(&g_prof[i].a) != hipSuccess || hipEventCreate(&g_prof[i].b) != hipSuccess) return XL_ERR_HIP;
    }
    g_profCap = max_records; g_profCount = 0; g_profOn = true;
    return XL_OK;
}

int xl_cnn_prof_pause(int on) { g_profOn = (on != 0) && g_prof; return XL_OK; }

int xl_cnn_prof_filter(int op_type, int min_nchunks2) { g_profType = op_type; g_profMinBatched = min_nchunks2; return XL_OK; }

int xl_cnn_prof_end(int32_t *op_index, int32_t *op_type, float *ms, int capacity)
{
    if (!g_prof) return XL_ERR_ARG;
    g_profOn = false;
    int n = g_profCount < capacity ? g_profCount : capacity;
    for (int i = 0; i < n; ++i) {
        float t = 0.f;
        if (hipEventSynchronize(g_prof[i].b) != hipSuccess || hipEventElapsedTime(&t, g_prof[i].a, g_prof[i].b) != hipSuccess) t = -1.f;
        if (op_index) op_index[i] = g_prof[i].opIndex;
        if (op_type) op_type[i] = g_prof[i].type;
        if (ms) ms[i] = t;
    }
    for (int i = 0; i < g_profCap; ++i) { (void)hipEventDestroy(g_prof[i].a); (void)hipEventDestroy(g_prof[i].b); }
    free(g_prof); g_prof = nullptr; g_profCap = 0; g_profCount = 0;
    return n;
}

}  // extern "C"
