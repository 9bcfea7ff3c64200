// crossloc_hip: the stride-2 3x3 stem convolutions (conv3 64->128, conv4 128->256, and conv2 32->64 when the fused stem is off;
// networks.py:191-201 of the reference) of inference plans as fp16 PAIRS, three matrix-pipe passes (round 5): the loop of
// csrc/xl_stem_split.hip - persistent workgroups walk tiles of 256 (128) output pixels x all Cout channels, the weights stream by
// LDS-DMA through a ring of three stages, every thread gathers 8 channels of ONE source pixel of its row per K-step (the pixel
// moves with the tap, out of the image = an out-of-range buffer offset = zero), normalises and converts them on their way into
// LDS - with the arithmetic of csrc/xl_gemm_pair.hip: an activation (times the plan's power-of-two scale) is {hi, lo' = (a - hi)
// 2^11}, a weight (times its matrix's own power of two) {hi, lo} with hs = hi 2^-11 derived in registers; products hs x lo',
// lo x hi, hi x hi on v_mfma_f32_32x32x16_f16, fp32 accumulation, exact un-scaling in the epilogue.  Weights: [Cout][9 Cin / 16][2][16]
// fp16, K ordered tap-major, + 2 floats (xl_cnn_pair_weight with taps = 9).  Both operands: 64-byte LDS rows, slot s of row r at
// s ^ swz(r).  A weight stage is Cout x 64 bytes = 16 / 8 / 4 DMA instructions: every wave issues two per K-step (straight-line
// code, the same counted waits in all waves; with Cout 64 those of waves 2 and 3 land in a scratch KB).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>

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

constexpr int kUnit = 64;                               // bytes per row and K-step, both operands: 2 planes x 16 fp16
__device__ __forceinline__ int swz(int row) { return ((row >> 2) & 3) ^ (((row >> 1) & 1) << 1); }      // (csrc/xl_gemm_pair.hip)
__device__ __forceinline__ f16x8 scale_hs(f16x8 hi) { return hi * (_Float16)0.00048828125f; }            // hi * 2^-11

struct PairStemArgs {
    const float *in; const unsigned char *u; const float *bias; float *out;
    const float *uInv; const float *aScale;          // inverse weight scale (one float); {s, 1 / s} of the activations
    const float *coef; float normLo;                 // NORM: [B][Cin][2] {scale, shift}; lower clamp (0 = ReLU, -inf = none)
    int B, Hi, Wi, Cin, Ho, Wo, ldIn, ldOut, M, nbm;
    int nbn;                                         // column tiles of NT channels (1, or 2 when Cout = 256 runs on 128-wide tiles)
    // GroupNorm partial sums of the output (stats == nullptr: none), G = Cout / 2 (Cout 64), / 4 (128), / 8 (256) groups: [B][nchunks][G][2] fp64
    // {sum, sum of squares}; chunk = (tile index within the image) * WM + (the wave's row block wm): one writer per entry,
    // every entry of a tile that overlaps the image is written.  nchunks >= (ceil(Ho*Wo / BM) + 1) * WM.
    double *stats; int G, nchunks;
};


// CPT = channels of a K-step a thread converts: 8 (two threads per row, tiles of 32 NW rows) or 16 (one thread per row, tiles of
// 64 NW rows: a wave then owns 64 rows - twice the MFMAs per byte of LDS traffic, what the 64-column layer is bound by)
template <int NT, bool NORM, int NW, int WPS = (NT == 64 ? 3 : 2), int CPT = 8>    // Cout; normalise on load; waves per workgroup; per SIMD
__global__ __launch_bounds__(64 * NW, WPS)                          // waves per SIMD: 3 workgroups of 4 waves per CU / 2 of 4 / 1 of 8
void pair_conv3x3s2_kernel(PairStemArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
    typedef __attribute__((address_space(3))) void lds_void;
    constexpr int NTH = 64 * NW, BM = NTH * CPT / 16;                 // threads; rows per tile (a thread = CPT channels of one row)
    constexpr int NH = CPT / 8, NL = CPT / 4;                         // 8-channel halves / 16-byte loads per thread and K-step
    constexpr int kAStage = BM * kUnit;                               // one activation stage
    constexpr int WN = NT / 64, WM = NW / WN, RI = BM / WM / 32;      // waves across columns / rows; 32-row blocks per wave
    static_assert(WN * WM == NW && RI >= 1 && RI * WM * 32 == BM, "tile shape");
    constexpr int kW = NT * kUnit;                                    // one weight stage
    constexpr int kA = 3 * kW;                                        // activation stages (behind the three weight stages)
    constexpr int kCoef = kA + 2 * kAStage;                           // coefficient tables of two tiles, 2 KB each (Cin <= 128)
    constexpr int kBias = kCoef + 4096;                               // bias[Cout <= 256]
    constexpr int kScratch = kBias + 1024;                            // 1 KB the DMA instructions of idle waves write zeros into
    constexpr int NDMA = NT * kUnit / 1024;                           // DMA instructions per weight stage
    constexpr int NS = 8 * RI;                                        // stores per wave and tile
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    // every wave issues two DMA instructions per K-step - the K-step is straight-line code with the same vmcnt arithmetic
    // in all waves; those of the waves beyond the stage (Cout 64: waves 2, 3) read out of range and land in a scratch KB
    const bool dmaWave = NDMA >= 2 * NW || wave * 2 < NDMA;
    const int dmaBase = __builtin_amdgcn_readfirstlane(dmaWave ? wave * 2 * 1024 : kScratch);
    const int dmaStage = __builtin_amdgcn_readfirstlane(dmaWave ? kW : 0), dmaQ = __builtin_amdgcn_readfirstlane(dmaWave ? 1024 : 0);

    const int total = a.nbm * a.nbn;
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3, nloc = gridDim.x >> 3;
    const int q8 = total >> 3, r8 = total & 7;
    const int runStart = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int runLen = q8 + (xcd < r8 ? 1 : 0);
    const int myCount = runLen > local ? (runLen - local + nloc - 1) / nloc : 0;
    if (myCount == 0) return;
    auto tile_m0 = [&](int i) { return ((runStart + local + i * nloc) / a.nbn) * BM; };
    auto tile_n0 = [&](int i) { return ((runStart + local + i * nloc) % a.nbn) * NT; };
    auto tile_rows = [&](int m0) { const int rows = a.M - m0; return rows < BM ? rows : BM; };

    constexpr unsigned OOB = 0x80000000u;
    const int HWo = a.Ho * a.Wo;
    const int nch = a.Cin >> 4;                                       // 16-channel chunks per tap
    const int nk = 9 * nch;
    const long long rowU = (long long)9 * a.Cin * 4;                  // bytes per weight row
    const float aS = a.aScale[0], aInv = a.aScale[1];
    const long long imgIn = (long long)a.Hi * a.Wi * a.ldIn;          // floats per input image

    // ---- stream two K-steps ahead of the multiplies
    const __amdgpu_buffer_rsrc_t srdU = __builtin_amdgcn_make_buffer_rsrc((void *)a.u, 0, (int)(a.nbn * NT * rowU), 0x00020000);
    // the descriptor of the stream's input window, as scalars (the descriptor is rebuilt from them at each load: values that
    // live in SGPRs - a descriptor carried across the loop in vector registers costs a readfirstlane loop per load)
    int inLo = 0, inHi = 0, inBytes = 0;
    auto srd_in = [&]() {
        const unsigned long long p = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(inHi) << 32) |
                                     (unsigned)__builtin_amdgcn_readfirstlane(inLo);
        return __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, __builtin_amdgcn_readfirstlane(inBytes), 0x00020000);
    };
    const int arow = CPT == 8 ? tid >> 1 : tid, ahalf = CPT == 8 ? tid & 1 : 0;
    unsigned gB[2];
    int dTile = 0, dK = 0, dChunk = 0, dDy = 0, dDx = 0;
    int pY = 0, pX = 0;                                               // 2 oy - 1, 2 ox - 1 of my row in the stream's tile
    unsigned pBase = OOB;                                             // byte offset of my row's image inside srdIn (OOB: no row)
    unsigned gTap = OOB;                                              // ... of the source pixel of the stream's tap (+ my half)
    auto set_tap = [&]() {
        const int iy = pY + dDy, ix = pX + dDx;
        const bool inb = (pBase != OOB) & ((unsigned)iy < (unsigned)a.Hi) & ((unsigned)ix < (unsigned)a.Wi);
        gTap = inb ? pBase + (unsigned)((iy * a.Wi + ix) * a.ldIn * 4 + ahalf * 32) : OOB;
    };
    auto set_dma_tile = [&](int i) {
        pBase = OOB;
        if (i < myCount) {
            const int m0 = __builtin_amdgcn_readfirstlane(tile_m0(i));
            const int nLo = __builtin_amdgcn_readfirstlane(m0 / HWo);
            const int left = a.B - nLo < 2 ? a.B - nLo : 2;           // a tile touches at most two images (Ho*Wo >= 256)
            const unsigned long long p = (unsigned long long)(a.in + nLo * imgIn);
            inLo = __builtin_amdgcn_readfirstlane((int)(unsigned)p);
            inHi = __builtin_amdgcn_readfirstlane((int)(unsigned)(p >> 32));
            inBytes = __builtin_amdgcn_readfirstlane((int)(left * imgIn * 4));
            const int m = m0 + arow;
            if (m < a.M) {
                const int n = m / HWo, p = m - n * HWo;
                const int oy = p / a.Wo, ox = p - oy * a.Wo;
                pY = 2 * oy - 1; pX = 2 * ox - 1;
                pBase = (unsigned)((n - nLo) * imgIn * 4);
            }
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {                                 // 16 rows x 4 slots per instruction
            const int row = (wave * 2 + q) * 16 + (lane >> 2);
            gB[q] = (i < myCount && dmaWave) ? (unsigned)((long long)(tile_n0(i) + row) * rowU + (((lane & 3) ^ swz(row)) * 16)) : OOB;
        }
        dDy = 0; dDx = 0; dChunk = 0;
        set_tap();
    };
    auto dma_instr = [&](int q, int stage) {                           // (waves that stream weights only)
        const int dst = dmaBase + __builtin_amdgcn_readfirstlane(stage) * dmaStage + q * dmaQ;     // (scalar arithmetic, no branch)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srdU, (lds_void *)(dsm + dst), 16,
                                                 (int)gB[q], __builtin_amdgcn_readfirstlane(dK) * kUnit, 0, 0);
    };
    u32x4 rA[2][NL];                                                   // [K-step parity][four channels of mine]
    unsigned mOK[2] = { 0u, 0u };                                      // [K-step parity] my source pixel is inside the image
    auto load_a = [&](auto parTag) {
        constexpr int P = decltype(parTag)::value;
        const __amdgpu_buffer_rsrc_t srdIn = srd_in();
#pragma unroll
        for (int l = 0; l < NL; ++l)
            rA[P][l] = __builtin_amdgcn_raw_buffer_load_b128(srdIn, (int)(gTap + 16u * l), __builtin_amdgcn_readfirstlane(dChunk) * 64, 0);
        mOK[P] = gTap != OOB ? 0xffffffffu : 0u;
    };
    auto advance_dma = [&]() {
        dK = __builtin_amdgcn_readfirstlane(dK + 1);
        if (dK == nk) { dK = 0; dTile = __builtin_amdgcn_readfirstlane(dTile + 1); set_dma_tile(dTile); return; }
        dChunk = __builtin_amdgcn_readfirstlane(dChunk + 1);
        if (dChunk == nch) {
            dChunk = 0;
            dDx = __builtin_amdgcn_readfirstlane(dDx + 1);
            if (dDx == 3) { dDx = 0; dDy = __builtin_amdgcn_readfirstlane(dDy + 1); }
            set_tap();
        }
    };

    // ---- conversion, one K-step ahead of the multiplies
    unsigned wOff[NH][2];
#pragma unroll
    for (int hh = 0; hh < NH; ++hh)
#pragma unroll
        for (int p = 0; p < 2; ++p) wOff[hh][p] = (unsigned)(kA + arow * kUnit + (((2 * p + ahalf + hh) ^ swz(arow)) * 16));
    int cTile = 0, cK = 0, cChunk = 0;
    unsigned cCoef = 0;                                              // LDS offset of my row's {scale, shift} run
    auto set_conv_tile = [&](int i) {
        if (NORM && i < myCount) {
            const int m0 = tile_m0(i);
            const int nLo = m0 / HWo;
            const int split = (nLo + 1) * HWo - m0;                  // first tile row of the second image
            cCoef = (unsigned)(kCoef + (i & 1) * 2048 + (arow >= split ? a.Cin * 8 : 0) + ahalf * 64);
        }
    };
    auto convert = [&](auto parTag) {                                  // registers of parity P -> activation stage P
        constexpr int P = decltype(parTag)::value;
        const float clampLo = mOK[P] != 0u ? a.normLo : 0.f, clampHi = mOK[P] != 0u ? __builtin_inff() : 0.f;
#pragma unroll
        for (int hh = 0; hh < NH; ++hh) {
        unsigned w[2][4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x4 x = __builtin_bit_cast(f32x4, rA[P][2 * hh + h]);
            if constexpr (NORM) {
                // (the table in LDS holds {scale, shift} * s: fmaf(x, scale s, shift s) = s fmaf(x, scale, shift) to the bit)
                const f32x4 c0 = *reinterpret_cast<const f32x4 *>(dsm + cCoef + cChunk * 128 + hh * 64 + h * 32);
                const f32x4 c1 = *reinterpret_cast<const f32x4 *>(dsm + cCoef + cChunk * 128 + hh * 64 + h * 32 + 16);
                // one rounding per element, then ONE v_med3_f32 that is both the lower clamp and the padding mask: in-image pixels
                // clamp to [normLo, +inf), the zero padding of the convolution to [0, 0]
                const f32x2 lo = f32x2{ fmaf(x[0], c0[0], c0[1]), fmaf(x[1], c0[2], c0[3]) };
                const f32x2 hi = f32x2{ fmaf(x[2], c1[0], c1[1]), fmaf(x[3], c1[2], c1[3]) };
                x = f32x4{ __builtin_amdgcn_fmed3f(lo[0], clampLo, clampHi), __builtin_amdgcn_fmed3f(lo[1], clampLo, clampHi),
                           __builtin_amdgcn_fmed3f(hi[0], clampLo, clampHi), __builtin_amdgcn_fmed3f(hi[1], clampLo, clampHi) };
            } else x *= aS;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const f32x2 v = f32x2{ x[2 * e], x[2 * e + 1] };
                const f16x2 vh = __builtin_convertvector(v, f16x2);
                const f16x2 vl = __builtin_convertvector((v - __builtin_convertvector(vh, f32x2)) * 2048.f, f16x2);
                w[0][2 * h + e] = __builtin_bit_cast(unsigned, vh);
                w[1][2 * h + e] = __builtin_bit_cast(unsigned, vl);
            }
        }
#pragma unroll
        for (int p = 0; p < 2; ++p)
            *reinterpret_cast<u32x4 *>(dsm + P * kAStage + wOff[hh][p]) = u32x4{ w[p][0], w[p][1], w[p][2], w[p][3] };
        }
    };
    auto advance_conv = [&]() {
        cK = __builtin_amdgcn_readfirstlane(cK + 1);
        if (cK == nk) { cK = 0; cChunk = 0; cTile = __builtin_amdgcn_readfirstlane(cTile + 1); set_conv_tile(cTile); return; }
        cChunk = __builtin_amdgcn_readfirstlane(cChunk + 1);
        if (cChunk == nch) cChunk = 0;
    };
    // coefficient table of tile i: the {scale, shift} pairs of its (at most two) images, 4 Cin floats, into table i & 1
    const __amdgpu_buffer_rsrc_t srdCoef = __builtin_amdgcn_make_buffer_rsrc((void *)a.coef, 0, NORM ? a.B * a.Cin * 8 : 0, 0x00020000);
    auto load_table = [&](int i) -> u32x4 {
        unsigned off = OOB;
        if (i < myCount && tid < a.Cin) off = (unsigned)(((long long)(tile_m0(i) / HWo) * a.Cin * 2 + tid * 4) * 4);
        return __builtin_amdgcn_raw_buffer_load_b128(srdCoef, (int)off, 0, 0);
    };
    auto store_table = [&](int i, u32x4 v) {
        if (tid < a.Cin) *reinterpret_cast<f32x4 *>(dsm + kCoef + (i & 1) * 2048 + tid * 16) = __builtin_bit_cast(f32x4, v) * aS;
    };

    // ---- fragments
    const int fr = lane & 31, kh = lane >> 5;
    unsigned slot[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) slot[p] = (unsigned)(((2 * p + kh) ^ swz(fr)) * 16);
    const unsigned frA = (unsigned)(kA + (wm * (32 * RI) + fr) * kUnit), frB = (unsigned)((wn * 64 + fr) * kUnit);
    f16x8 fa[2][RI], fb[2][2], fbs[2];
    f32x16 acc[RI][2];
    auto ldA = [&](int stage, int p, int i) { return *reinterpret_cast<const f16x8 *>(dsm + stage * kAStage + frA + i * 32 * kUnit + slot[p]); };
    auto ldB = [&](int stage, int p, int j) { return *reinterpret_cast<const f16x8 *>(dsm + stage * kW + frB + j * 32 * kUnit + slot[p]); };
    auto mma = [&](const f16x8 (&b)[2], const f16x8 (&v)[RI]) {
#pragma unroll
        for (int i = 0; i < RI; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[j], v[i], acc[i][j], 0, 0, 0);
    };
    const int rhalf = kh * 4;
    auto init_acc = [&](int n0) {                                      // accumulators start at the bias
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 b = *reinterpret_cast<const f32x4 *>(dsm + kBias + (n0 + wn * 64 + j * 32 + rhalf + 8 * q) * 4);
#pragma unroll
                for (int i = 0; i < RI; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][j][4 * q + e] = b[e];
            }
    };

    // ---- prologue: bias and the first two coefficient tables into LDS, steps 0 and 1 of the stream, step 0 converted
    // (accumulators start at the bias in the scaled domain: bias * s * weight scale, powers of two)
    const float biasMul = aS * (1.f / a.uInv[0]);
    for (int i = tid; i < a.nbn * NT; i += NTH) reinterpret_cast<float *>(dsm + kBias)[i] = a.bias[i] * biasMul;
    if constexpr (NORM) {
        const u32x4 t0 = load_table(0), t1 = load_table(1);
        store_table(0, t0);
        store_table(1, t1);
    }
    set_dma_tile(0);
    set_conv_tile(0);
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    load_a(P0{});
    dma_instr(0, 0); dma_instr(1, 0);
    advance_dma();
    __builtin_amdgcn_s_waitcnt(0x0070);                               // everything landed
    __syncthreads();                                                  // tables and bias visible
    convert(P0{});
    advance_conv();
    load_a(P1{});
    dma_instr(0, 1); dma_instr(1, 1);
    advance_dma();
    __builtin_amdgcn_s_waitcnt(0x0070 | (2 + NL));                    // my writes of step 0; stage 0 of the ring landed before
    __builtin_amdgcn_s_barrier();
    int sc = 0, sd = 2;
    int statStores = 0;                                               // statistics stores of the last epilogue (uniform)
    init_acc(tile_n0(0));
    // one K-step; FIRST: the first step of a tile that follows another one (NS stores of its epilogue are in flight)
    auto step = [&](auto firstTag, auto parTag) __attribute__((always_inline)) {
        constexpr int sa = decltype(parTag)::value;                   // parity of the K-step
        const int next = sc == 2 ? 0 : sc + 1;
        load_a(parTag);                                               // step kk + 2: two steps until its conversion
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[0][j] = ldB(sc, 0, j);
#pragma unroll
        for (int i = 0; i < RI; ++i) fa[1][i] = ldA(sa, 1, i);
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[1][j] = ldB(sc, 1, j);
#pragma unroll
        for (int i = 0; i < RI; ++i) fa[0][i] = ldA(sa, 0, i);
#pragma unroll
        for (int j = 0; j < 2; ++j) fbs[j] = scale_hs(fb[0][j]);
        mma(fbs, fa[1]); dma_instr(0, sd);                             // hs x lo'
        __builtin_amdgcn_sched_barrier(0);
        // lo x hi with the conversion of step kk + 1 threaded through it
        mma(fb[1], fa[0]);
        convert(std::integral_constant<int, sa ^ 1>{});               // (the compiler counts vmcnt for rA)
        {
            constexpr int nM = 2 * RI;                                // MFMAs of the term
            constexpr int valu = (NORM ? 56 : 40) * NH / nM;
#pragma unroll
            for (int g = 0; g < nM; ++g) {
                if constexpr (NORM) { if (g == 0 || g == nM / 2) __builtin_amdgcn_sched_group_barrier(0x100, 2 * NH, 0); }
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, valu, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x200, 2 * NH, 0);                 // the LDS writes
        }
        __builtin_amdgcn_sched_barrier(0);
        // the weights of step kk + 1 have landed: younger are 1 DMA and the NL loads of step kk + 2 - and, in the first step
        // of a tile, the NS stores of the tile before; lgkmcnt(0): my activation writes are done
        if constexpr (decltype(firstTag)::value) {
            constexpr int S1 = NT == 64 ? 16 : 8;                      // statistics stores per slot (2 column blocks x 16 / NR groups)
            constexpr int W0 = 1 + NL + NS, W1 = W0 + S1, W2 = W0 + 2 * S1;
            static_assert(W2 < 64, "vmcnt is a 6-bit counter");
            if (statStores == 0) __builtin_amdgcn_s_waitcnt(0x0070 | (W0 & 15) | ((W0 >> 4) << 14));
            else if (statStores == S1) __builtin_amdgcn_s_waitcnt(0x0070 | (W1 & 15) | ((W1 >> 4) << 14));
            else __builtin_amdgcn_s_waitcnt(0x0070 | (W2 & 15) | ((W2 >> 4) << 14));
        }
        else __builtin_amdgcn_s_waitcnt(0x0070 | (1 + NL));
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        mma(fb[0], fa[0]); dma_instr(1, sd);                           // hi x hi
        advance_conv();                                               // (the branches of the two streams' bookkeeping end the step)
        advance_dma();
        sc = next;
        sd = sd == 2 ? 0 : sd + 1;
        __builtin_amdgcn_sched_barrier(0);                            // (the vmcnt arithmetic above assumes this issue order)
    };
    auto epilogue = [&](int ti) __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
        // ---- tile ti (swapped operands): row = lane & 31, channels (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
        const int m0 = tile_m0(ti), n0 = tile_n0(ti);
        {
            const float inv = aInv * a.uInv[0];                        // un-scale (exact)
#pragma unroll
            for (int i = 0; i < RI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] *= inv;
        }
        if constexpr (NORM) {                                          // (waits for everything older than the table)
            const u32x4 tab = load_table(ti + 2);
            store_table(ti + 2, tab);
        }
        // (every wave issues exactly NS stores per tile - the vmcnt arithmetic of the next step counts them: rows past the
        //  end of the tile fall outside the descriptor)
        const __amdgpu_buffer_rsrc_t srdO = __builtin_amdgcn_make_buffer_rsrc((void *)(a.out + (long long)m0 * a.ldOut), 0,
                                                                              tile_rows(m0) * a.ldOut * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < RI; ++i) {
            const unsigned rowOff = (unsigned)((wm * (32 * RI) + i * 32 + fr) * a.ldOut * 4);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = n0 + wn * 64 + j * 32 + rhalf + 8 * q;
                    const unsigned off = rowOff + (unsigned)n * 4u;
                    const f32x4 v = f32x4{ acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3] };
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), srdO, (int)off, 0, 0);
                }
        }
        if (a.stats != nullptr) {
            // GroupNorm partial sums of the output (round 4: the statistics passes over the stem tensors are gone).  Groups
            // of CPG = Cout / G channels (32 groups: 2, 4, 8 channels for conv2..conv4).  A lane's
            // accumulator r holds channel 8 (r >> 2) + 4 kh + (r & 3) of its 32-column block: a group of 2 / 4 channels lies
            // within one lane, one of 8 / 16 spans the two halves of the wave.  A tile touches at most two images (Ho*Wo >=
            // 256 >= BM): slot 0 = rows before `split`, slot 1 = the rest.  Fixed order: per lane fp32 over the group's
            // channels and its RI rows, the fp32 DPP tree of xl_half_wave_sum / xl_wave_sum_top over the 32 rows (x 2 halves), one fp64 entry
            // per (image, tile, wm, group) - one writer, every entry of a tile that overlaps the image written.
            // Its stores come AFTER the tile's NS output stores (the trees run under those) and are counted by the vmcnt wait of
            // the next tile's first step (statStores: 0, S1 or 2 S1 more instructions in flight) - issued before them they were
            // the OLDEST thing that wait covers, i.e. a store acknowledgement per tile on the critical path (+12 % on conv3).
            const int nLo = m0 / HWo;
            const int split = (nLo + 1) * HWo - m0;
            const bool two = split < BM && nLo + 1 < a.B;              // (uniform)
            const int kT = m0 / BM - (int)(((long long)nLo * HWo) / BM);
            double *oLo = a.stats + ((long long)nLo * a.nchunks + kT * WM + wm) * a.G * 2;
            double *oHi = a.stats + ((long long)(nLo + 1) * a.nchunks + wm) * a.G * 2;
            typedef double f64x2 __attribute__((ext_vector_type(2)));
            auto sums = [&](auto cpgTag) __attribute__((always_inline)) {
                constexpr int CPG = decltype(cpgTag)::value;
                constexpr int NR = CPG >= 16 ? 8 : (CPG >= 4 ? 4 : 2);   // accumulator registers per group and lane
                constexpr bool FULL = CPG >= 8;                          // the group spans both halves of the wave
                // (all the trees first, then one predicated block of stores)
                constexpr int NU = 16 / NR;
                float s0[2][NU], q0[2][NU], s1[2][NU], q1[2][NU];
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int u = 0; u < NU; ++u) {
                        float a0 = 0.f, b0 = 0.f, a1 = 0.f, b1 = 0.f;
#pragma unroll
                        for (int i = 0; i < RI; ++i) {
                            const int row = wm * (32 * RI) + i * 32 + fr;
                            float t = 0.f, tt = 0.f;
#pragma unroll
                            for (int e = 0; e < NR; ++e) {
                                const float v = acc[i][j][NR * u + e];
                                t += v;
                                tt = fmaf(v, v, tt);
                            }
                            const bool hi = row >= split, live = m0 + row < a.M;
                            a0 += (live && !hi) ? t : 0.f; b0 += (live && !hi) ? tt : 0.f;
                            a1 += (live && hi) ? t : 0.f;  b1 += (live && hi) ? tt : 0.f;
                        }
                        s0[j][u] = FULL ? xl_wave_sum_top(a0) : xl_half_wave_sum(a0);
                        q0[j][u] = FULL ? xl_wave_sum_top(b0) : xl_half_wave_sum(b0);
                        if (two) {
                            a1 = FULL ? xl_wave_sum_top(a1) : xl_half_wave_sum(a1);
                            b1 = FULL ? xl_wave_sum_top(b1) : xl_half_wave_sum(b1);
                        }
                        s1[j][u] = a1; q1[j][u] = b1;
                    }
                statStores = two ? 4 * NU : 2 * NU;
                if (FULL ? lane == 63 : fr == 31) {                    // (a lane that holds the totals)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int u = 0; u < NU; ++u) {
                            const int cb = n0 + wn * 64 + j * 32;
                            const int g = CPG >= 16 ? (cb >> 4) + u : CPG == 8 ? (cb >> 3) + u : CPG == 4 ? (cb >> 2) + 2 * u + kh
                                                                                               : (cb >> 1) + 4 * (u >> 1) + 2 * kh + (u & 1);
                            *reinterpret_cast<f64x2 *>(oLo + 2 * g) = f64x2{ (double)s0[j][u], (double)q0[j][u] };
                            if (two) *reinterpret_cast<f64x2 *>(oHi + 2 * g) = f64x2{ (double)s1[j][u], (double)q1[j][u] };
                        }
                }
            };
            // (only the group sizes the stem has - 32 groups: Cout 64 -> 2, 128 -> 4, 256 -> 8, also as two 128-column tiles - are
            //  instantiated: each one costs registers in a kernel that has none to spare; the launcher rejects the rest)
            const int cpg = a.nbn * NT / a.G;
            if constexpr (NT == 64) sums(std::integral_constant<int, 2>{});
            else if constexpr (NT == 256) sums(std::integral_constant<int, 8>{});
            else if (cpg == 4) sums(std::integral_constant<int, 4>{});
            else sums(std::integral_constant<int, 8>{});
        }
        if (ti + 1 < myCount) init_acc(tile_n0(ti + 1));
        __builtin_amdgcn_sched_barrier(0);
    };
    auto tile_steps = [&](auto firstTag) __attribute__((always_inline)) {
        step(firstTag, P0{});
        step(std::false_type{}, P1{});
        for (int kk = 2; kk < nk; kk += 2) {
            step(std::false_type{}, P0{});
            step(std::false_type{}, P1{});
        }
    };
    tile_steps(std::false_type{});
    for (int ti = 1; ti < myCount; ++ti) {
        epilogue(ti - 1);
        tile_steps(std::true_type{});
    }
    epilogue(myCount - 1);
    __builtin_amdgcn_s_waitcnt(0x0F70);                             // vmcnt(0)
}

template <int NT, int NW, int PER_CU, int CPT = 8>
int launch_pair_stem(PairStemArgs a, bool norm, hipStream_t st)
{
    constexpr int BM = 4 * NW * CPT;
    const size_t lds = 3 * NT * kUnit + 2 * BM * kUnit + 4096 + 1024 + 1024;
    static XlLdsLimit configured[2];
    int cfgDev;
    constexpr int WPS = NW * PER_CU / 4 < 2 ? 2 : NW * PER_CU / 4;
    const void *fn = norm ? reinterpret_cast<const void *>(pair_conv3x3s2_kernel<NT, true, NW, WPS, CPT>)
                          : reinterpret_cast<const void *>(pair_conv3x3s2_kernel<NT, false, NW, WPS, CPT>);
    if (configured[norm].needs(lds, &cfgDev)) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return XL_ERR_HIP;
        configured[norm].done(lds, cfgDev);
    }
    a.nbm = (a.M + BM - 1) / BM;
    constexpr int WM = NW / (NT / 64);
    if (a.stats && (a.nchunks < ((a.Ho * a.Wo + BM - 1) / BM + 1) * WM || a.G * (NT == 64 ? 2 : NT == 256 || a.nbn == 2 ? 8 : 4) != a.nbn * NT)) return XL_ERR_ARG;
    int grid = 256 * PER_CU;                                          // persistent: PER_CU workgroups per CU (LDS- and register-bound)
    if (grid > ((a.nbm * a.nbn + 7) & ~7)) grid = (a.nbm * a.nbn + 7) & ~7;
    if (norm) hipLaunchKernelGGL((pair_conv3x3s2_kernel<NT, true, NW, WPS, CPT>), dim3(grid), dim3(64 * NW), lds, st, a);
    else hipLaunchKernelGGL((pair_conv3x3s2_kernel<NT, false, NW, WPS, CPT>), dim3(grid), dim3(64 * NW), lds, st, a);
    return XL_OK;
}

}  // namespace

// XL_OP_CONV with ksize 3, stride 2 and XL_CONV_SPLIT_BF16 | XL_CONV_SPLIT_IL | XL_CONV_PAIR_F16: in fp32 NHWC [B,Hi,Wi,Cin] (ld_in), out fp32
// NHWC [B,Ho,Wo,Cout] (ld_out), w = [Cout][9 Cin / 16][2][16] fp16 + 2 floats (K = tap * Cin + c, tap = 3 ky + kx), bias, scale; optionally
// XL_CONV_NORM_IN (aux2 = [B][Cin][2] coefficients, XL_CONV_NORM_RELU).  Cin in {32, 64, 128}, Cout in {64, 128, 256},
// Ho*Wo >= 256.  stats (optional, groups = 32, nchunks): GroupNorm partial sums of the output, see StemArgs - chunk =
// tile * WM + wm with (rows per tile, WM) = (128, 4) for Cout 64, (128, 2) for Cout 128, (256, 2) for Cout 256 and (128, 2)
// for its latency form; XL_OP_GN_FINAL sums them with reserved_i = rows per tile, stride = WM.
int xl_run_pair_stem(const xl_op &op, hipStream_t st)
{
    const long long M = (long long)op.B * op.Ho * op.Wo;
    const bool norm = (op.flags & XL_CONV_NORM_IN) != 0;
    if (op.ksize != 3 || op.stride != 2 || (op.Cin != 32 && op.Cin != 64 && op.Cin != 128) ||
        (op.Cout != 64 && op.Cout != 128 && op.Cout != 256) || op.Ho != (op.Hi - 1) / 2 + 1 || op.Wo != (op.Wi - 1) / 2 + 1 ||
        op.Ho * op.Wo < 256 || op.ld_in < op.Cin || op.ld_out < op.Cout || (op.ld_in & 3) || (op.ld_out & 3) || !op.bias ||
        (op.flags & (XL_CONV_ACCUMULATE | XL_CONV_DGRAD)) || !op.in || !op.w || !op.out || (op.stats && ((uintptr_t)op.stats & 15)) ||
        (((uintptr_t)op.in | (uintptr_t)op.out | (uintptr_t)op.w) & 15) || M >= 0x7fffffffLL - 256 || !op.scale ||
        2LL * op.Hi * op.Wi * op.ld_in * 4 >= 0x7fffffffLL || 256LL * op.ld_out * 4 >= 0x7fffffffLL || (norm && !op.aux2))
        return XL_ERR_ARG;
    PairStemArgs a;
    a.uInv = reinterpret_cast<const float *>((const unsigned char *)op.w + (long long)op.Cout * 9 * op.Cin * 4) + 1;
    a.aScale = (const float *)op.scale;
    a.in = (const float *)op.in; a.u = (const unsigned char *)op.w; a.bias = (const float *)op.bias; a.out = (float *)op.out;
    a.coef = (const float *)op.aux2;
    a.normLo = (op.flags & XL_CONV_NORM_RELU) ? 0.f : -__builtin_inff();
    a.B = op.B; a.Hi = op.Hi; a.Wi = op.Wi; a.Cin = op.Cin; a.Ho = op.Ho; a.Wo = op.Wo;
    a.ldIn = op.ld_in; a.ldOut = op.ld_out; a.M = (int)M; a.nbm = 0; a.nbn = 1;
    a.stats = (double *)op.stats; a.G = op.groups; a.nchunks = op.nchunks;
    static const char *form = getenv("XL_STEM_FORM");                // measurement switch: "8x2" = 8-wave workgroups, two per CU
    if (op.Cout == 64) return form && !strcmp(form, "8x2") ? launch_pair_stem<64, 8, 2>(a, norm, st)
                            : form && !strcmp(form, "c16") ? launch_pair_stem<64, 4, 2, 16>(a, norm, st) : launch_pair_stem<64, 4, 3>(a, norm, st);
    if (op.Cout == 128) return form && !strcmp(form, "8x2") ? launch_pair_stem<128, 8, 1>(a, norm, st) : launch_pair_stem<128, 4, 2>(a, norm, st);
    if (op.reserved_i == 128) {                                       // latency form (the host asks when 256-row tiles cannot fill
        a.nbn = 2;                                                    // the chip): 128 x 128 tiles, two column tiles per row tile
        return launch_pair_stem<128, 4, 2>(a, norm, st);
    }
    return launch_pair_stem<256, 8, 1>(a, norm, st);
}
