// crossloc_hip: conv1 evaluated INSIDE conv2's operand stage (round 4; inference plans).
//
//   conv2( relu( groupnorm( conv1(image) ) ) )      networks.py:186-193 of the reference: 3 -> 32 (3x3 s1) -> 32 -> 64 (3x3 s2)
//
// Until round 3 conv1 wrote its raw 32-channel full-resolution output (2.08 GB at 47 frames of 480 x 720) and conv2 read it
// back, applying GroupNorm + ReLU and splitting every value into its three bf16 terms once per (output pixel, tap) that uses
// it - 2.25 times per value - inside its K-loop: 7 VALU instructions per MFMA, the layer ran at a third of the matrix pipe.
// Here the full-resolution tensor never exists.  The GroupNorm statistics of conv1 come from a statistics-only evaluation
// (conv1_mfma_kernel<0>, csrc/xl_cnn.hip, then XL_OP_GN_FINAL); this kernel then walks tiles of 8 x 16 conv2 outputs:
//   1. the 19 x 36 image halo of the tile is split into bf16 planes in LDS ({R, G, B, 0} words, as conv1_mfma_kernel);
//   2. conv1 is evaluated on the 17 x 33 patch of its outputs the tile needs - 18 blocks of 32 pixels, 18 MFMAs each, the
//      arithmetic of conv1_mfma_kernel term for term -, normalised with the {scale, shift} pairs (one fmaf, one v_med3_f32 that
//      is both the ReLU and the zero padding of conv2), split ONCE per value and written to the patch in LDS:
//      [pixel][plane][32 channels] bf16, 208 bytes per pixel (13 x 16: consecutive pixels fall on distinct 16-byte slots of the
//      256-byte bank window), the columns of a patch row de-interleaved (even columns first) so that the stride-2 gather of
//      conv2 reads CONSECUTIVE pixel records;
//   3. conv2: every wave owns 32 output pixels (two tile rows) x all 64 channels; a K-step = 16 channels of one tap; the
//      activation fragments are plain ds_read_b128 from the patch at compile-time offsets, the weight fragments come from
//      global memory in fragment order (1 KB contiguous per load instruction, L2-resident: 110 KB for the layer), three
//      K-steps ahead in registers.  No barrier, no conversion, no staging in the K-loop: 12 MFMAs per 9 fragment reads.
// Term pairs and K order are those of split_conv3x3s2_kernel<64,...> (csrc/xl_stem_split.hip): the raw conv2 output is
// bitwise what the two-kernel path produces (tests/test_cnn_gpu.py::test_fused_stem_is_bitwise_the_two_kernel_path).
// One workgroup of 4 waves per CU (133 KB of LDS), persistent over tiles; two barriers per tile.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

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
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr int kTW = 16;                                // conv2 output columns per tile (rows: the template parameter TH)
constexpr int kPW = 2 * kTW + 1;                       // conv1 output columns the tile needs: 33
constexpr int kEven = kTW + 1;                         // even patch columns come first in a patch row (17 of them)
constexpr int kHW = kPW + 3;                           // image halo columns: 36 (35 + the dx = 3 slot of the last pixel)
constexpr int kPix = 208;                              // bytes per patch pixel: 3 planes x 64 B + 16
constexpr int kPixPair = 144;                          // ... of the fp16-pair form (round 5): 2 planes x 64 B + 16 (9 x 16 as 13 x 16: odd)
// TH = 8: 17 x 33 patch (116 688 B) + 19 x 36 halo (16 416 B): one workgroup per CU.  TH = 4: 9 x 33 patch (61 776 B) + 11 x 36
// halo (9 504 B) = 71 KB: TWO workgroups per CU (8 waves: the conversion-heavy conv1 phase of one overlaps the MFMA-only conv2
// phase of the other), or one beside a workgroup of the solver (64.8 KB), which runs on a side stream under the stem.
template <int TH, bool PAIR = false> struct S12 {
    static constexpr int kPH = 2 * TH + 1, kHH = kPH + 2;
    static constexpr int kNPix = kPH * kPW, kBlocks = (kNPix + 31) / 32;
    static constexpr int kPixB = PAIR ? kPixPair : kPix;
    // Patch rows are padded to a multiple of 128 bytes (round 6).  A ds_read_b128 is serviced in lane groups {0-3, 12-15, 20-27}, ...:
    // lanes 20-27 of conv2's fragment reads sit on the SECOND tile row of their wave (two patch rows further), and with the dense
    // pitch (33 records: 297 / 429 slots of 16 bytes) their slots collided with those of lanes 12-15 - every fragment read took 8
    // LDS cycles instead of 4 (SQ_LDS_BANK_CONFLICT 0.40 of the kernel's LDS cycles in round 5).  With 2 x pitch = 0 mod 256 bytes
    // the 16 lanes of a group fall on 16 distinct slots of the bank row whatever the tap.
    static constexpr int kPitch = kPW * kPixB + (PAIR ? 112 : 48);
    static constexpr int kPatchBytes = kPH * kPitch, kHaloPix = kHH * kHW, kHaloBytes = 3 * kHaloPix * 8;
    static constexpr int kE = (kHaloPix + 255) / 256;                // halo pixels per thread
    static constexpr int kPB = TH / 2;                               // 32-pixel blocks per tile (4 waves: kPB x 4 / kPB column blocks)
};

// the split of conv1_mfma_kernel (integer round-to-nearest-even) - the image halo must be split exactly as there
__device__ __forceinline__ unsigned s12_bf16_rn(float x)
{
    unsigned u = __builtin_bit_cast(unsigned, x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ void s12_split3(float a, unsigned &h1, unsigned &h2, unsigned &h3)
{
    h1 = s12_bf16_rn(a);
    const float r1 = a - __builtin_bit_cast(float, h1 << 16);
    h2 = s12_bf16_rn(r1);
    h3 = s12_bf16_rn(r1 - __builtin_bit_cast(float, h2 << 16));
}
// the split of split_conv3x3s2_kernel (v_cvt_pk_bf16_f32, exact residuals): the normalised activations, two at a time
__device__ __forceinline__ float s12_hi(unsigned w) { return __builtin_bit_cast(float, w & 0xffff0000u); }
__device__ __forceinline__ float s12_lo(unsigned w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ void s12_split_pair(f32x2 v, unsigned &w1, unsigned &w2, unsigned &w3)
{
    w1 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    const f32x2 r = v - f32x2{ s12_lo(w1), s12_hi(w1) };
    w2 = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
    const f32x2 r2 = r - f32x2{ s12_lo(w2), s12_hi(w2) };
    w3 = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2));
}

struct Stem12Args {
    const float *img;            // [B][3][H][W]
    const u32x4 *w1;             // conv1 weight fragments [3 planes][3 window rows][64 lanes] x 16 B (networks._Plan.conv1_fragments)
    const float *b1;             // [32]
    const float *coef;           // [B][32][2] {scale, shift} of conv1's GroupNorm (XL_OP_GN_FINAL)
    const u32x4 *w2;             // conv2 weight fragments [18 K-steps][3 planes][2 column blocks][64 lanes] x 16 B (PAIR: 2 planes {hi, lo} fp16)
    const float *aScale;         // PAIR: {s, 1 / s} of the activations (conv1's normalised output); conv2's inverse weight scale: one float behind w2
    const float *b2;             // [64]
    float *out;                  // [B][Ho][Wo][64], pixel stride ldOut
    int B, H, W, Ho, Wo, ldOut, tilesX, tilesY;
    float normLo;                // 0 (ReLU) or -inf
    int *queue;                  // [2] = {next tile, workgroups done}: tiles are handed out dynamically (both zero before and after a launch)
    long long *clk;              // diagnostics (XL_STEM12_CLK=1): per-wave shader-clock sums of the phases of a tile, else NULL
    // GroupNorm partial sums of the output (stats == nullptr: none), 32 groups of 2 channels: [B][nchunks][32][2] fp64 {sum, sum of
    // squares}; chunk = (tile within the image) * (TH / 2) + (the wave's 32-pixel block): one writer per entry, every entry written
    double *stats; int nchunks;
};


// PAIR (round 5): conv2 as three fp16 passes (csrc/xl_gemm_pair.hip's arithmetic, term for term that of pair_conv3x3s2_kernel<64>:
// the raw conv2 output stays bitwise what the two-kernel path produces): conv1's normalised output times the plan's scale goes
// into the patch as {hi, lo'} (144 bytes per pixel), conv2's weights arrive as {hi, lo} fragments of the scaled matrix, hs = hi 2^-11
// in registers.  conv1 itself - its operand is the IMAGE, for which there is no bound - keeps the three bf16 terms.
template <int TH, bool PAIR = false, int WGS = (TH == 8 ? 1 : 2)>
__global__ __launch_bounds__(256, WGS)
void stem12_kernel(Stem12Args a)
{
    typedef S12<TH, PAIR> K;
    constexpr int kPixB = K::kPixB, NPL = PAIR ? 2 : 3, kPitch = K::kPitch;
    constexpr int kTH = TH, kHH = K::kHH, kNPix = K::kNPix, kBlocks = K::kBlocks, kPatchBytes = K::kPatchBytes;
    constexpr int kHaloPix = K::kHaloPix, kE = K::kE, kPB = K::kPB, kNJ = kPB == 4 ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
    unsigned char *sPatch = dsm;
    u32x2 *sH = reinterpret_cast<u32x2 *>(dsm + kPatchBytes);
    float *sTab = reinterpret_cast<float *>(dsm + kPatchBytes + K::kHaloBytes);   // scale[32], shift[32] of the tile's image; conv2 bias[64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, kh = lane >> 5;
    const long long HW = (long long)a.H * a.W;
    const int tilesPerImage = a.tilesX * a.tilesY;
    const int total = a.B * tilesPerImage;

    // conv1 weight fragments and biases (registers for the whole kernel)
    bf16x8 wf[3][3];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) wf[p][dy] = __builtin_bit_cast(bf16x8, a.w1[(p * 3 + dy) * 64 + lane]);
    // accumulator element r of a lane: pixel lane & 31, channel 8 (r >> 2) + 4 kh + (r & 3) (+ 32 j for conv2's column block j)
    // conv2: wave -> (32-pixel block pxb: tile rows 2 pxb, 2 pxb + 1; first column block jb; kNJ column blocks of 32 channels)
    const int pxb = wave % kPB, jb = (wave / kPB) * kNJ;
    if (tid < 64) { if (!PAIR) sTab[64 + tid] = a.b2[tid]; }          // (visible after the first barrier)
    else if (tid < 96) sTab[64 + tid] = a.b1[tid - 64];               // conv1 bias[32] at sTab[128]

    // image halo of a tile: pixel i = tid + 256 e of the 19 x 36 window, three channels each, fetched one tile ahead
    float pre[kE][3];
    auto prefetch = [&](int t) {
        const int n = t / tilesPerImage, tt = t - n * tilesPerImage;
        const int ty = tt / a.tilesX, tx = tt - ty * a.tilesX;
        const float *img = a.img + (long long)n * 3 * HW;
#pragma unroll
        for (int e = 0; e < kE; ++e) {
            const int i = tid + 256 * e;
            const int r = i / kHW, c = i - r * kHW;
            const int y = 2 * kTH * ty - 2 + r, x = 2 * kTW * tx - 2 + c;
            const bool inb = (t < total) & (i < kHaloPix) & ((unsigned)y < (unsigned)a.H) & ((unsigned)x < (unsigned)a.W);
            const long long off = inb ? (long long)y * a.W + x : 0;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) pre[e][ch] = inb ? img[ch * HW + off] : 0.f;
        }
    };

    // conv2: my output pixel inside the tile and the byte offset of its patch record (tap (0, 0), plane 0)
    const int oyl = 2 * pxb + (fr >> 4), oxl = fr & 15;
    const unsigned aBase = (unsigned)((2 * oyl) * kPitch + oxl * kPixB + kh * 16);
    float aS = 1.f, unscale = 1.f;
    if constexpr (PAIR) {
        aS = a.aScale[0];
        const float uInv = reinterpret_cast<const float *>(a.w2 + 18 * 2 * 2 * 64)[0];     // (behind the fragments)
        unscale = a.aScale[1] * uInv;
        // biases of conv2 in the scaled domain (bias * s * weight scale: powers of two)
        if (tid < 64) sTab[64 + tid] = a.b2[tid] * (aS * (1.f / uInv));
    }

    const __amdgpu_buffer_rsrc_t srdW2 = __builtin_amdgcn_make_buffer_rsrc((void *)a.w2, 0, 18 * NPL * 2 * 1024, 0x00020000);
    long long cPh[6] = { 0, 0, 0, 0, 0, 0 }, cT = 0, nTiles = 0;
    auto stamp = [&](int ph) { if (a.clk) { const long long now = clock64(); cPh[ph] += now - cT; cT = now; } };
    // Tiles come from a queue (one atomic per tile, fetched two tiles ahead), not from a static stride: the solver of the batch
    // before runs on a side stream under the stem and holds CUs (LDS, registers) for a millisecond - with a static partition the
    // workgroups that do not fit beside it start when others END, and the launch takes twice as long (measured: 2.2 vs 1.26 ms).
    int *sQ = reinterpret_cast<int *>(sTab + 192);
    if (tid == 0) { sQ[0] = atomicAdd(a.queue, 1); sQ[1] = atomicAdd(a.queue, 1); }
    __syncthreads();
    int t = sQ[0], tNext = sQ[1];
    __syncthreads();                                                  // (sQ[0] is rewritten in the first tile)
    prefetch(t);
    if (a.clk) cT = clock64();
    while (t < total) {
        ++nTiles;
        if (tid == 0) sQ[0] = atomicAdd(a.queue, 1);                  // the tile after the next one (read behind barrier 1)
        const int n = t / tilesPerImage, tt = t - n * tilesPerImage;
        const int ty = tt / a.tilesX, tx = tt - ty * a.tilesX;
        const int oy0 = kTH * ty, ox0 = kTW * tx;

        // ---- 1. halo -> LDS, split ({R | G << 16, B} per pixel and plane)
#pragma unroll
        for (int e = 0; e < kE; ++e) {
            const int i = tid + 256 * e;
            if (i < kHaloPix) {
                unsigned h[3][3];
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) s12_split3(pre[e][ch], h[0][ch], h[1][ch], h[2][ch]);
#pragma unroll
                for (int p = 0; p < 3; ++p) sH[p * kHaloPix + i] = u32x2{ h[p][0] | (h[p][1] << 16), h[p][2] };
            }
        }
        // {scale, shift} of the 32 conv1 channels for this image (a table in LDS: 32 registers fewer than per-lane copies)
        if (tid < 32) {
            const f32x2 c2 = *reinterpret_cast<const f32x2 *>(a.coef + ((long long)n * 32 + tid) * 2);
            sTab[tid] = c2[0] * aS; sTab[32 + tid] = c2[1] * aS;       // (PAIR: {scale, shift} s - fmaf(x, scale s, shift s) = s fmaf(x, scale, shift))
        }
        stamp(0);                                                      // halo split + coefficient loads
        __syncthreads();
        stamp(1);                                                      // barrier 1
        const int tAfter = sQ[0];
        // conv2's weight fragments of the first three K-steps: in flight under conv1
        constexpr int D = kNJ == 1 ? 6 : 3;                 // K-steps of weight fragments in flight (6 kNJ MFMAs = 192 kNJ cycles each)
        u32x4 fbr[D][NPL][kNJ];
        auto load_b = [&](int kk, int slot) {
#pragma unroll
            for (int p = 0; p < NPL; ++p)
#pragma unroll
                for (int j = 0; j < kNJ; ++j)          // (buffer load: one lane-offset register + a scalar offset per fragment)
                    fbr[slot][p][j] = __builtin_amdgcn_raw_buffer_load_b128(srdW2, (int)(lane * 16), (int)(((kk * NPL + p) * 2 + jb + j) * 1024), 0);
        };
#pragma unroll
        for (int kk = 0; kk < D; ++kk) load_b(kk, kk);

        // ---- 2. conv1 on the patch: blocks of 32 patch pixels (linear index L = row * 33 + de-interleaved column).
        // (round 6: the image fragments of a wave's NEXT block are read while the MFMAs of the current one execute and in front of its
        //  conversion arithmetic - a block used to be one dependent chain reads -> 18 MFMAs -> conversion -> writes)
        auto block_pixel = [&](int blk, bool &valid, int &pr, int &idx, int &x) {
            const int L = blk * 32 + fr;
            valid = L < kNPix;
            const int Lc = valid ? L : kNPix - 1;
            pr = Lc / kPW; idx = Lc - pr * kPW;
            x = idx < kEven ? 2 * idx : 2 * (idx - kEven) + 1;                    // patch column
        };
        auto load_pf = [&](bf16x8 (&pf)[3][3], int pr, int x) {
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const u32x2 *src = sH + p * kHaloPix + (pr + dy) * kHW + x + 2 * kh;
                    const u32x2 v0 = src[0], v1 = src[1];
                    pf[p][dy] = __builtin_bit_cast(bf16x8, u32x4{ v0[0], v0[1], v1[0], v1[1] });
                }
        };
        bf16x8 pfN[3][3];
        {
            bool v0; int pr0, idx0, x0;
            block_pixel(wave, v0, pr0, idx0, x0);
            load_pf(pfN, pr0, x0);
        }
#pragma unroll 1
        for (int blk = wave; blk < kBlocks; blk += 4) {
            bool valid; int pr, idx, x;
            block_pixel(blk, valid, pr, idx, x);
            f32x16 acc;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 b4 = *reinterpret_cast<const f32x4 *>(sTab + 128 + 8 * q + 4 * kh);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[4 * q + e] = b4[e];
            }
            bf16x8 pf[3][3];
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) pf[p][dy] = pfN[p][dy];
            constexpr int PW_[6] = { 2, 1, 0, 1, 0, 0 }, PP_[6] = { 0, 1, 2, 0, 1, 0 };   // smallest terms first (conv1_mfma_kernel)
#pragma unroll
            for (int tm = 0; tm < 6; ++tm)
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[PW_[tm]][dy], pf[PP_[tm]][dy], acc, 0, 0, 0);
            if (blk + 4 < kBlocks) {
                bool vN; int prN, idxN, xN;
                block_pixel(blk + 4, vN, prN, idxN, xN);
                load_pf(pfN, prN, xN);
            }
            // GroupNorm + ReLU; a conv1 pixel outside the image is conv2's zero padding
            const int iy = 2 * oy0 - 1 + pr, ix = 2 * ox0 - 1 + x;
            const bool inimg = ((unsigned)iy < (unsigned)a.H) & ((unsigned)ix < (unsigned)a.W);
            const float lo = inimg ? a.normLo : 0.f, hi = inimg ? __builtin_inff() : 0.f;
            unsigned char *dst = sPatch + pr * kPitch + idx * kPixB + kh * 8;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 sc4 = *reinterpret_cast<const f32x4 *>(sTab + 8 * q + 4 * kh), sh4 = *reinterpret_cast<const f32x4 *>(sTab + 32 + 8 * q + 4 * kh);
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = __builtin_amdgcn_fmed3f(fmaf(acc[4 * q + e], sc4[e], sh4[e]), lo, hi);
                unsigned wa[3], wb[3];
                if constexpr (PAIR) {
                    const f32x2 va = f32x2{ v[0], v[1] }, vb = f32x2{ v[2], v[3] };
                    const f16x2 ha = __builtin_convertvector(va, f16x2), hb = __builtin_convertvector(vb, f16x2);
                    wa[0] = __builtin_bit_cast(unsigned, ha); wb[0] = __builtin_bit_cast(unsigned, hb);
                    wa[1] = __builtin_bit_cast(unsigned, __builtin_convertvector((va - __builtin_convertvector(ha, f32x2)) * 2048.f, f16x2));
                    wb[1] = __builtin_bit_cast(unsigned, __builtin_convertvector((vb - __builtin_convertvector(hb, f32x2)) * 2048.f, f16x2));
                } else {
                    s12_split_pair(f32x2{ v[0], v[1] }, wa[0], wa[1], wa[2]);
                    s12_split_pair(f32x2{ v[2], v[3] }, wb[0], wb[1], wb[2]);
                }
                if (valid) {
#pragma unroll
                    for (int p = 0; p < NPL; ++p) *reinterpret_cast<u32x2 *>(dst + p * 64 + q * 16) = u32x2{ wa[p], wb[p] };
                }
            }
        }
        stamp(2);                                                      // conv1 on my blocks of the patch
        __syncthreads();
        stamp(3);                                                      // barrier 2

        // ---- 3. conv2 on the patch; the image halo of my next tile is fetched meanwhile
        prefetch(tNext);
        f32x16 acc2[kNJ];
#pragma unroll
        for (int j = 0; j < kNJ; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 b4 = *reinterpret_cast<const f32x4 *>(sTab + 64 + 32 * (jb + j) + 8 * q + 4 * kh);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc2[j][4 * q + e] = b4[e];
            }
        // (round 6: the activation fragments of K-step kk + 1 are read in front of the MFMAs of step kk - the step used to open with
        //  its own two reads and wait out their latency with the matrix pipe idle: one accumulator, nothing else to issue)
        auto tap_off = [&](int kk) {
            const int tap = kk >> 1, c = kk & 1, ky = tap / 3, kx = tap - 3 * ky;
            return (unsigned)(ky * kPitch + ((kx & 1) * kEven + (kx >> 1)) * kPixB + c * 32);
        };
        f16x8 faN[2];
        if constexpr (PAIR) {
#pragma unroll
            for (int p = 0; p < 2; ++p) faN[p] = *reinterpret_cast<const f16x8 *>(sPatch + aBase + tap_off(0) + p * 64);
        }
#pragma unroll
        for (int kk = 0; kk < 18; ++kk) {
            const unsigned off = tap_off(kk);
            const int slot = kk % D;
            if constexpr (PAIR) {
                f16x8 fa[2];
#pragma unroll
                for (int p = 0; p < 2; ++p) fa[p] = faN[p];
                if (kk + 1 < 18) {
#pragma unroll
                    for (int p = 0; p < 2; ++p) faN[p] = *reinterpret_cast<const f16x8 *>(sPatch + aBase + tap_off(kk + 1) + p * 64);
                }
                // hs x lo', lo x hi, hi x hi: pair_conv3x3s2_kernel's terms and order
#pragma unroll
                for (int j = 0; j < kNJ; ++j) {
                    const f16x8 hs = __builtin_bit_cast(f16x8, fbr[slot][0][j]) * (_Float16)0.00048828125f;
                    acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hs, fa[1], acc2[j], 0, 0, 0);
                }
#pragma unroll
                for (int j = 0; j < kNJ; ++j)
                    acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fbr[slot][1][j]), fa[0], acc2[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < kNJ; ++j)
                    acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fbr[slot][0][j]), fa[0], acc2[j], 0, 0, 0);
            } else {
            bf16x8 fa[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) fa[p] = *reinterpret_cast<const bf16x8 *>(sPatch + aBase + off + p * 64);
            constexpr int PU[6] = { 2, 1, 0, 1, 0, 0 }, PV[6] = { 0, 1, 2, 0, 1, 0 };   // (weights, activations): split_conv3x3s2's order
#pragma unroll
            for (int tm = 0; tm < 6; ++tm)
#pragma unroll
                for (int j = 0; j < kNJ; ++j)
                    acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fbr[slot][PU[tm]][j]), fa[PV[tm]], acc2[j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (kk + D < 18) load_b(kk + D, slot);
            __builtin_amdgcn_sched_barrier(0);
        }

        if constexpr (PAIR) {
#pragma unroll
            for (int j = 0; j < kNJ; ++j) acc2[j] *= unscale;         // (exact)
        }
        stamp(4);                                                      // conv2
        // ---- 4. store: pixel (oy0 + oyl, ox0 + oxl), channels 32 j + 8 q + 4 kh + {0..3}
        const int oy = oy0 + oyl, ox = ox0 + oxl;
        if (a.stats != nullptr) {
            // GroupNorm partial sums of conv2's output (round 4: no statistics pass over the 64-channel half-resolution tensor):
            // 32 groups of 2 channels - accumulator registers (2 u, 2 u + 1) of a lane are group 16 (jb + j) + 4 (u >> 1) + 2 kh +
            // (u & 1); fp32 over the pair, the fp32 tree of xl_half_wave_sum over the 32 pixels, one fp64 entry per (tile, block)
            const bool live = oy < a.Ho && ox < a.Wo;
            double *o = a.stats + ((long long)n * a.nchunks + tt * kPB + pxb) * 64;
            typedef double f64x2 __attribute__((ext_vector_type(2)));
            // (all the trees first - DPP only, no LDS traffic - then ONE predicated block of stores by a lane that holds the totals)
            float s1[kNJ][8], s2[kNJ][8];
#pragma unroll
            for (int j = 0; j < kNJ; ++j)
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float v0 = acc2[j][2 * u], v1 = acc2[j][2 * u + 1];
                    s1[j][u] = xl_half_wave_sum(live ? v0 + v1 : 0.f);
                    s2[j][u] = xl_half_wave_sum(live ? fmaf(v1, v1, v0 * v0) : 0.f);
                }
            // lanes 16-23 / 48-55 hold the totals of their half: lane 16 + u (48 + u) stores group-pair u - one 16-byte store
            // instruction per column block instead of one per group (the per-lane choice is 7 selects per value)
            if ((fr >> 3) == 2) {
                const int us = fr & 7;
#pragma unroll
                for (int j = 0; j < kNJ; ++j) {
                    float p1 = s1[j][0], p2 = s2[j][0];
#pragma unroll
                    for (int u = 1; u < 8; ++u) { p1 = us == u ? s1[j][u] : p1; p2 = us == u ? s2[j][u] : p2; }
                    *reinterpret_cast<f64x2 *>(o + 2 * (16 * (jb + j) + 4 * (us >> 1) + 2 * kh + (us & 1))) = f64x2{ (double)p1, (double)p2 };
                }
            }
        }
        if (oy < a.Ho && ox < a.Wo) {
            float *o = a.out + (((long long)n * a.Ho + oy) * a.Wo + ox) * a.ldOut + 4 * kh;
#pragma unroll
            for (int j = 0; j < kNJ; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<f32x4 *>(o + 32 * (jb + j) + 8 * q) = f32x4{ acc2[j][4 * q], acc2[j][4 * q + 1], acc2[j][4 * q + 2], acc2[j][4 * q + 3] };
        }
        stamp(5);                                                      // stores issued
        t = tNext; tNext = tAfter;
    }
    // the last workgroup to finish leaves the queue as it found it (every workgroup made its last fetch before it counts itself done)
    if (tid == 0) {
        __threadfence();
        if (atomicAdd(a.queue + 1, 1) == (int)gridDim.x - 1) { a.queue[0] = 0; a.queue[1] = 0; __threadfence(); }
    }
    if (a.clk && lane == 0) {
        long long *c = a.clk + ((long long)blockIdx.x * 4 + wave) * 8;
        for (int i = 0; i < 6; ++i) c[i] = cPh[i];
        c[6] = nTiles;
    }
}

}  // namespace

// XL_OP_STEM12: in = image [B,3,Hi,Wi] NCHW; w = conv1 weight fragments; bias = conv1 bias[32]; aux2 = {scale, shift} pairs
// [B][32][2] of conv1's GroupNorm; aux = conv2 weight fragments [18][3][2][64][8] bf16; stats2 = conv2 bias[64] (read only);
// out = raw conv2 output [B,Ho,Wo,64] NHWC (ld_out); flags & XL_GN_RELU_IN: ReLU after the GroupNorm; out2 = two ints, zero (the
// tile queue of the launch; left zero again - one buffer per op and stream); stats (optional; groups = 32, nchunks = tiles per
// image * tile rows / 2): GroupNorm partial sums of the output, [B][nchunks][32][2] fp64, every entry written (XL_OP_GN_FINAL
// with reserved_i = 0 sums all nchunks of them).
int xl_run_stem12(const xl_op &op, hipStream_t st)
{
    if (op.Cin != 3 || op.Cout != 64 || op.Ho != (op.Hi - 1) / 2 + 1 || op.Wo != (op.Wi - 1) / 2 + 1 || op.ld_out < 64 || (op.ld_out & 3) ||
        !op.in || !op.w || !op.bias || !op.aux || !op.aux2 || !op.stats2 || !op.out2 || !op.out || op.B < 1 ||
        (((uintptr_t)op.out | (uintptr_t)op.w | (uintptr_t)op.aux | (uintptr_t)op.aux2 | (uintptr_t)op.stats) & 15) ||
        (op.stats && op.groups != 32))
        return XL_ERR_ARG;
    Stem12Args a;
    a.img = (const float *)op.in; a.w1 = (const u32x4 *)op.w; a.b1 = (const float *)op.bias; a.coef = (const float *)op.aux2;
    a.w2 = (const u32x4 *)op.aux; a.b2 = (const float *)op.stats2; a.out = (float *)op.out; a.queue = (int *)op.out2;
    const bool pair = (op.flags & XL_CONV_PAIR_F16) != 0;             // aux = conv2 {hi, lo} fragments [18][2][2][64][8] fp16 + the inverse scale
    if (pair && !op.scale) return XL_ERR_ARG;
    a.aScale = (const float *)op.scale;
    a.stats = (double *)op.stats; a.nchunks = op.nchunks;
    a.B = op.B; a.H = op.Hi; a.W = op.Wi; a.Ho = op.Ho; a.Wo = op.Wo; a.ldOut = op.ld_out;
    // tile rows: 4 (two workgroups per CU, 71 KB each - default) or 8 (one of 133 KB; XL_STEM12_TILE=8)
    static const int tileRows = getenv("XL_STEM12_TILE") && atoi(getenv("XL_STEM12_TILE")) == 8 ? 8 : 4;
    a.tilesX = (op.Wo + kTW - 1) / kTW; a.tilesY = (op.Ho + tileRows - 1) / tileRows;
    a.normLo = (op.flags & XL_GN_RELU_IN) ? 0.f : -__builtin_inff();
    const long long total = (long long)op.B * a.tilesX * a.tilesY;
    if (total >= 0x7fffffffLL || (op.stats && op.nchunks != a.tilesX * a.tilesY * (tileRows / 2))) return XL_ERR_ARG;
    const size_t lds = (tileRows == 8 ? (pair ? S12<8, true>::kPatchBytes : S12<8>::kPatchBytes) + S12<8>::kHaloBytes
                                      : (pair ? S12<4, true>::kPatchBytes : S12<4>::kPatchBytes) + S12<4>::kHaloBytes) + 960;
    // (tables: 192 floats + the queue's two ints = 776 bytes.  960, not 1024: LDS is handed out in 512-byte granules, and three
    //  workgroups of the pair form fit a CU only up to 106 granules = 54272 bytes each - 43776 + 9504 + 960 = 54240)
    const void *fn = tileRows == 8 ? (pair ? reinterpret_cast<const void *>(stem12_kernel<8, true>) : reinterpret_cast<const void *>(stem12_kernel<8>))
                                   : (pair ? reinterpret_cast<const void *>(stem12_kernel<4, true>) : reinterpret_cast<const void *>(stem12_kernel<4>));
    static XlLdsLimit configured[4];
    int cfgDev;
    const int cslot = (tileRows == 8 ? 1 : 0) + (pair ? 2 : 0);
    if (configured[cslot].needs(lds, &cfgDev)) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return XL_ERR_HIP;
        configured[cslot].done(lds, cfgDev);
    }
    // (XL_STEM12_WGS=3, pair form with 4-row tiles: three workgroups per CU - 53 KB of LDS and at most 168 registers each)
    static const bool wg3 = getenv("XL_STEM12_WGS") && atoi(getenv("XL_STEM12_WGS")) == 3;
    const bool three = wg3 && pair && tileRows == 4 && 3 * lds <= 160 * 1024;
    if (three) {
        static XlLdsLimit configured3;
        if (configured3.needs(lds, &cfgDev)) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(stem12_kernel<4, true, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return XL_ERR_HIP;
            configured3.done(lds, cfgDev);
        }
    }
    int grid = tileRows == 8 ? 256 : three ? 768 : 512;               // persistent: one / two (three) workgroups per CU
    if (grid > total) grid = (int)total;
    static const bool clkDbg = getenv("XL_STEM12_CLK") != nullptr;
    a.clk = nullptr;
    if (clkDbg && hipMalloc(&a.clk, sizeof(long long) * 32 * grid) != hipSuccess) return XL_ERR_HIP;
    if (tileRows == 8) {
        if (pair) hipLaunchKernelGGL((stem12_kernel<8, true>), dim3(grid), dim3(256), lds, st, a);
        else hipLaunchKernelGGL((stem12_kernel<8, false>), dim3(grid), dim3(256), lds, st, a);
    } else {
        if (three) hipLaunchKernelGGL((stem12_kernel<4, true, 3>), dim3(grid), dim3(256), lds, st, a);
        else if (pair) hipLaunchKernelGGL((stem12_kernel<4, true>), dim3(grid), dim3(256), lds, st, a);
        else hipLaunchKernelGGL((stem12_kernel<4, false>), dim3(grid), dim3(256), lds, st, a);
    }
    if (clkDbg) {
        std::vector<long long> h((size_t)32 * grid);
        if (hipStreamSynchronize(st) == hipSuccess && hipMemcpy(h.data(), a.clk, sizeof(long long) * 32 * grid, hipMemcpyDeviceToHost) == hipSuccess) {
            static const char *names[6] = { "halo split", "barrier 1", "conv1", "barrier 2", "conv2", "stores" };
            for (int w = 0; w < 4; ++w) {
                double v[6] = { 0 }, tiles = 0;
                for (int b = 0; b < grid; ++b) { for (int i = 0; i < 6; ++i) v[i] += (double)h[((size_t)b * 4 + w) * 8 + i]; tiles += (double)h[((size_t)b * 4 + w) * 8 + 6]; }
                fprintf(stderr, "[stem12 clk] wave %d, ticks per tile:", w);
                for (int i = 0; i < 6; ++i) fprintf(stderr, " %s %.0f", names[i], v[i] / tiles);
                fprintf(stderr, "\n");
            }
        }
        (void)hipFree(a.clk);
    }
    return XL_OK;
}
