// crossloc_hip: the split-pipe GEMMs with HALF the matrix-pipe work (round 5): fp16 pairs and triples, three passes.
//
//   M_z[t][o] = sum_c V_z[t][c] * U_z[o][c]
//
// csrc/xl_gemm_split.hip writes an fp32 operand as three bf16 terms and multiplies six term pairs.  fp16 carries 11
// significand bits against bf16's 8, so TWO fp16 terms hold 22 bits and the products that matter are three:
//
//     a = ah + al,   b = bh + bl,        a b  =  ah bh  +  ah bl  +  al bh   ( + al bl, below 2^-22 of ah bh: dropped )
//
// on v_mfma_f32_32x32x16_f16 - the same 2.5 PFLOP/s pipe as the bf16 instruction, an fp16 x fp16 product is exact in fp32,
// accumulation in fp32.  What fp16 lacks is bf16's exponent range, and the two operands deal with that differently:
//   * weights (static, packed once per weight version, csrc/xl_pack.hip): every matrix is scaled by the power of two that
//     puts its largest element into [2^14, 2^15) and stored as the pair {hi = fp16(w), lo = fp16(w - hi)}; the kernels derive
//     hs = hi * 2^-11 from the hi fragment in registers (v_pk_mul_f16, exact down to fp16's subnormals);
//   * activations are scaled by ONE power of two per plan (xl_op.scale, chosen from a bound on the network's activations
//     so that nothing can overflow) and stored as the PAIR {hi = fp16(a), lo' = fp16((a - hi) * 2^11)}: the low term is kept
//     2^11 times too large, i.e. in fp16's normal range whenever hi is - its 11 bits survive for any |a| in [2^-13, 65504],
//     so a loose bound (and with it a small scale) costs no accuracy - and its product is taken with `hs`, the weight's
//     high term scaled down by the same 2^11:      a b  =  hi * bh  +  hi * bl  +  lo' * hs.
// Measured error of the dominant launch against a float64 product of the same fp32 operands: see bench.py
// (`split_gemm_err_vs_f64`, beside the fp32-MFMA kernel's own).
//
// Layouts (K-step = 16 channels = one k-depth of the MFMA):
//     activations  [Z][rows][C/8][2][8] fp16       64 bytes per row and K-step  (4 bytes per element: what fp32 costs)
//     weights      [Z][rows][C/16][2][16] fp16     the same, then 2 Z floats: scratch, inverse scales
// Kernels:
//   pair_gemm_kernel      - both operands arrive in that form (V written by the Winograd input transform) and reach LDS by
//                           DMA: no conversion, no register staging, no LDS writes from the ALU side;
//   pair_conv1x1_kernel   - activations arrive as fp32 (1x1 layers; normalise-on-load, residual-on-load), the pairs are
//                           formed on the way into LDS as csrc/xl_gemm_split.hip forms its bf16 terms.
// Both are the 256 x 256 persistent loops of csrc/xl_gemm_split.hip (one workgroup per CU, 8 waves of 128 x 64, ONE bare
// s_barrier per K-step behind a counted vmcnt wait) with half the MFMAs per K-step - and therefore half the time to hide a
// memory access in: the LDS-DMA ring has FOUR stages of 32 KB (three K-steps ahead) where the bf16 kernels have three of 48.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
#include <vector>

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

constexpr int kPA = 64;                                  // bytes per activation row and K-step: {hi, lo'} x 16 fp16
constexpr int kPB = 64;                                  // bytes per weight row and K-step: {hi, lo} x 16 fp16
constexpr unsigned OOB = 0x80000000u;

// LDS rows of both operands are 4 slots of 16 bytes (hi k0-7, hi k8-15, lo k0-7, lo k8-15), slot s of row r at
// physical slot s ^ swz(r).  ds_read_b128 is serviced in four groups of 16 lanes ({0-3, 12-15, 20-27}, ...); the 16 rows of a
// group must hit 16 different 16-byte columns of the 256-byte bank row: (4 r + slot) mod 16, i.e. rows equal mod 4 need
// different swz - bits 2-3 of the row.  Bit 1 is folded into the upper slot bit for the converting kernel's ds_write_b128
// (groups of 8 consecutive lanes = 4 rows x 2 halves over 128 bytes of banks: rows r and r + 2 would collide).
__device__ __forceinline__ int swz(int row) { return ((row >> 2) & 3) ^ (((row >> 1) & 1) << 1); }

struct PairArgs {
    const unsigned char *v, *u;  // [Z][T][C/16][2][16] fp16, [Z][N][C/16][2][16] fp16
    float *out;                  // [Z][T][N] fp32
    const float *uInv;           // [Z] inverse weight scales
    const float *aScale;         // {s, 1 / s} of the activations
    int T, C, N, Z, nbm, nbn;
    long long *clk;              // diagnostics (XL_PAIR_CLK=1): per-wave shader-tick sums of the four phases of a K-step
    int var;                     // measurement switch (XL_PAIR_VAR=4: the epilogue stores as the accumulators lie, 32-byte pieces)
};

// Epilogue stores (round 6).  A lane of a 32 x 32 accumulator block (swapped operands) holds 4 x 4 consecutive channels of ONE
// row, 8 apart (the other 4 of each 8 in lane + 32): stored as they lie, an instruction writes 32 bytes into each of 32 rows - 32
// requests of half a 64-byte L2 request each, and the CU's store path takes ~440 ticks per instruction (14000 per 256 x 256 tile,
// XL_PAIR_CLK).  Neighbouring lanes (rows m, m + 1) trade one channel quad of each pair (quad_perm [1,0,3,2]): the first
// instruction of a pair then writes 64 contiguous bytes into each of the 16 EVEN rows, the second into the odd rows - 7900 ticks
// per tile, dominant launch 1.65 -> 1.56 ms on random data.  Same values, same number of stores.
// pair PQ of the block: o0 = (row m & ~1, channels 16 PQ + 8 (m & 1) + 4 (lane >> 5) ..+3), o1 = the same channels of row m | 1.
template <int PQ>
__device__ __forceinline__ void lane_pair_exchange(const f32x16 &c, bool odd, f32x4 &o0, f32x4 &o1)
{
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float x0 = c[8 * PQ + e], x1 = c[8 * PQ + 4 + e];
        const float give = odd ? x0 : x1;
        const float recv = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, give), 0xB1, 0xF, 0xF, true));
        o0[e] = odd ? recv : x0;
        o1[e] = odd ? x1 : recv;
    }
}

__device__ __forceinline__ f16x8 scale_hs(f16x8 hi) { return hi * (_Float16)0.00048828125f; }      // hi * 2^-11: 4 x v_pk_mul_f16

// vmcnt bookkeeping (in order, per wave): a wave issues 4 DMA instructions per K-step (2 per operand), all in front of the step's
// barrier, into the stage that was multiplied in the step before.  At the barrier of step s the operands of step s + 1 must
// have landed - issued in step s - 2; younger are the 4 DMAs of step s - 1 and the 4 of step s: vmcnt(8), and vmcnt(40) in the
// first TWO steps of a tile that follows another one (the 32 stores of its epilogue lie between).
// DBG (XL_PAIR_DBG, measurement only - the results are garbage): 1 = no DMA after the prologue, 2 = no fragment reads in the loop,
// 4 = no barrier / vmcnt wait, 8 = no epilogue stores (round 6): what each piece costs under the chip's power limit.
// Round 6 (random data, 95 frames, one box): full kernel 1.65 ms; no stores 1.35; no DMA 1.36; MFMAs + stores only 1.20; MFMAs only
// 1.00.  The stores' 0.30 ms was the store PATTERN as much as the bytes: see lane_pair_exchange (1.65 -> 1.56).
// WM = 1 (round 6, XL_PAIR_PP=1: "ping-pong"): the workgroup is ONE row of four waves - 128 x 256 tiles, 256 threads, a ring of THREE
// stages of 24 KB - and two of them share a CU, each with its own barrier: a SIMD holds one wave of each, and while one workgroup
// stores a tile, waits at its barrier or issues its DMAs, the other one multiplies.  Price: the weights' tile is fetched by twice
// as many workgroups (6 DMA instructions per wave and K-step for the same 24 MFMAs instead of 4).
template <int CT, int DBG = 0, int WM = 2>                             // compile-time channel count (0: a.C)
__global__ __launch_bounds__(256 * WM, 2 / WM)
void pair_gemm_kernel(PairArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
    typedef __attribute__((address_space(3))) void lds_void;
    constexpr int BMT = 128 * WM;                                     // tile rows
    constexpr int kOpA = BMT * kPA, kOpB = 256 * kPB, kStage = kOpA + kOpB;     // 16 (8) + 16 KB per stage
    constexpr int NST = WM == 2 ? 4 : 3;
    constexpr int NA = 2, NB = 4 / WM, ND = NA + NB;                  // DMA instructions per wave and K-step: activations, weights
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = WM == 2 ? wave >> 2 : 0, wn = wave & 3;            // WM x 4 waves of 128 x 64

    // tiles of this workgroup: XCD x (= block % 8) owns a contiguous run of the (z, m-tile, n-tile) order, its workgroups
    // take every nloc-th tile of the run, so the workgroups of an XCD work on neighbouring tiles at any time
    const int total = a.nbm * a.nbn * a.Z;
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3, nloc = gridDim.x >> 3;
    const int q8 = total >> 3, r8 = total & 7;
    const int runStart = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int runLen = q8 + (xcd < r8 ? 1 : 0);
    const int myCount = runLen > local ? (runLen - local + nloc - 1) / nloc : 0;
    if (myCount == 0) return;

    const long long rowB = (long long)a.C * 4;                        // bytes per operand row (both operands)
    const int nk = CT ? CT / 16 : a.C / 16;

    // ---- operand stream: an instruction moves 16 rows x 4 slots; wave w fills rows 32 w .. 32 w + 31 of both operands.  A lane's
    // offset inside a tile never changes (row and slot of the 16 x 4 piece); the tile is the descriptor's base, and rows past the
    // end of an operand - or a tile past the end of my list - fall outside the descriptor's extent and read as zero
    __amdgpu_buffer_rsrc_t srdV, srdU;
    unsigned lOffA[NA], lOffB[NB];
#pragma unroll
    for (int q = 0; q < NA; ++q) {
        const int row = (wave * NA + q) * 16 + (lane >> 2);
        lOffA[q] = (unsigned)(row * (int)rowB + (((lane & 3) ^ swz(row)) * 16));
    }
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        const int row = (wave * NB + q) * 16 + (lane >> 2);
        lOffB[q] = (unsigned)(row * (int)rowB + (((lane & 3) ^ swz(row)) * 16));
    }
    int dTile = 0, dK = 0;                                            // position of the stream: tile of my list, K-step
    auto set_dma_tile = [&](int i) {
        int t = runStart + local + (i < myCount ? i : 0) * nloc;
        const int z = t / (a.nbm * a.nbn);
        t -= z * (a.nbm * a.nbn);
        const int mt = t / a.nbn, nt = t - mt * a.nbn;
        const int m0 = mt * BMT, n0 = nt * 256;
        const int rowsA = i < myCount ? (a.T - m0 < BMT ? a.T - m0 : BMT) : 0, rowsB = i < myCount ? 256 : 0;
        srdV = __builtin_amdgcn_make_buffer_rsrc((void *)(a.v + ((long long)z * a.T + m0) * rowB), 0, (int)(rowsA * rowB), 0x00020000);
        srdU = __builtin_amdgcn_make_buffer_rsrc((void *)(a.u + ((long long)z * a.N + n0) * rowB), 0, (int)(rowsB * rowB), 0x00020000);
    };
    bool dmaOn = true;
    auto dma_instr = [&](int q, int stage) {                          // instruction q of ND of the stream's current step
        if ((DBG & 1) && !dmaOn) return;
        unsigned char *base = dsm + stage * kStage;
        if (q < NA) __builtin_amdgcn_raw_ptr_buffer_load_lds(srdV, (lds_void *)(base + (wave * NA + q) * 1024), 16, (int)lOffA[q], dK * kPA, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(srdU, (lds_void *)(base + kOpA + (wave * NB + q - NA) * 1024), 16, (int)lOffB[q - NA], dK * kPB, 0, 0);
    };
    auto advance_dma = [&]() {
        if (++dK == nk) { dK = 0; ++dTile; set_dma_tile(dTile); }
    };

    // ---- fragments: lane -> row lane & 31 of a 32-row block, k-half lane >> 5 (8 fp16 = one 16-byte slot)
    const int fr = lane & 31, kh = lane >> 5;
    // (a K-step's four slots: activations hi(k 0-7) lo'(k 0-7) hi(k 8-15) lo'(k 8-15) - what the input transform's quads write -,
    //  weights hi(k 0-7) hi(k 8-15) lo(k 0-7) lo(k 8-15))
    unsigned slot[2], slotA[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        slot[p] = (unsigned)(((2 * p + kh) ^ swz(fr)) * 16);
        slotA[p] = (unsigned)(((2 * kh + p) ^ swz(fr)) * 16);
    }
    const unsigned frA = (unsigned)((wm * 128 + fr) * kPA), frB = (unsigned)(kOpA + (wn * 64 + fr) * kPB);
    f16x8 fa[2][4], fb[2][2], fbs[2];                                 // [hi, lo'][row block], [hi, lo][column block], hs
    f16x8 faN[4], fbN[2];
    f32x16 acc[4][2];
    bool readOn = true;
    auto ldA = [&](const unsigned char *sb, int p, int i) {
        if ((DBG & 2) && !readOn) return faN[i];
        return *reinterpret_cast<const f16x8 *>(sb + frA + i * 32 * kPA + slotA[p]);
    };
    auto ldB = [&](const unsigned char *sb, int p, int j) {
        if ((DBG & 2) && !readOn) return fbN[j];
        return *reinterpret_cast<const f16x8 *>(sb + frB + j * 32 * kPB + slot[p]);
    };
    auto mma = [&](const f16x8 (&b)[2], const f16x8 (&v)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[j], v[i], acc[i][j], 0, 0, 0);
    };

    // ---- prologue: steps 0 .. NST - 2 of the stream
    set_dma_tile(0);
#pragma unroll
    for (int st = 0; st < NST - 1; ++st) {
#pragma unroll
        for (int q = 0; q < ND; ++q) dma_instr(q, st);
        advance_dma();
    }
    constexpr int kYoung = ND * (NST - 2);                            // DMAs younger than the step a barrier needs: 8 (6)
    __builtin_amdgcn_s_waitcnt(0x0F70 | kYoung);
    // (a bare s_barrier: the workgroup fence of __syncthreads() makes the compiler wait for EVERY outstanding LDS-DMA)
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int j = 0; j < 2; ++j) fbN[j] = ldB(dsm, 0, j);
#pragma unroll
    for (int i = 0; i < 4; ++i) faN[i] = ldA(dsm, 1, i);
    int sc = 0, sd = NST - 1;                                         // stage being multiplied / being filled
    const int rhalf = (lane >> 5) * 4;
    const float aInv = a.aScale[1];
    dmaOn = false; readOn = false;
    long long cPre = 0, cVm = 0, cBar = 0, cTail = 0, cT = 0, cEpi = 0, cVm2 = 0;
    if (a.clk) cT = clock64();
    for (int ti = 0; ti < myCount; ++ti) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int kk = 0; kk < nk; ++kk) {
            const unsigned char *sb = dsm + sc * kStage;
            const int next = sc + 1 == NST ? 0 : sc + 1;
            // The two waves of a SIMD (wm = 0: the older one, which wins the matrix pipe's arbitration) take their four DMA
            // instructions at opposite ends of the phase: an LDS-DMA costs its wave 100-200 ticks of issue in which it multiplies
            // nothing, and with both waves in the same order those windows coincide (measured, XL_PAIR_CLK: the phase took the
            // younger waves 1837 ticks for 1024 of MFMA work per SIMD).  Now one wave multiplies while the other issues.
            if (wm) {
#pragma unroll
                for (int q = 0; q < ND; ++q) dma_instr(q, sd);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) { fb[0][j] = fbN[j]; fbs[j] = scale_hs(fbN[j]); }
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[1][i] = faN[i];
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[1][j] = ldB(sb, 1, j);
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[0][i] = ldA(sb, 0, i);
            mma(fbs, fa[1]);                                            // hs x lo'
            mma(fb[1], fa[0]);                                          // lo x hi
            __builtin_amdgcn_sched_barrier(0);
            if (!wm) {
#pragma unroll
                for (int q = 0; q < ND; ++q) dma_instr(q, sd);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (a.clk) { const long long t = clock64(); cPre += t - cT; cT = t; }
            if (!(DBG & 4)) {
                // (vmcnt is a 6-bit field: bits 3:0 and 15:14 of the immediate)
                constexpr int kS = kYoung + 32, kS1 = kYoung + 1;
                if ((DBG & 8) && kk < NST - 2 && ti > 0) __builtin_amdgcn_s_waitcnt(0x0F70 | kS1);          // (one store per tile)
                else if (kk < NST - 2 && ti > 0) __builtin_amdgcn_s_waitcnt(0x0F70 | (kS & 15) | ((kS >> 4) << 14));   // + the 32 stores of the tile before
                else __builtin_amdgcn_s_waitcnt(0x0F70 | kYoung);
            }
            if (a.clk) { const long long t = clock64(); cVm += t - cT; if (kk < 2 && ti > 0) cVm2 += t - cT; cT = t; }
            if (!(DBG & 4)) __builtin_amdgcn_s_barrier();
            if (a.clk) { const long long t = clock64(); cBar += t - cT; cT = t; }
            __builtin_amdgcn_sched_barrier(0);
            const unsigned char *sn = dsm + next * kStage;
#pragma unroll
            for (int j = 0; j < 2; ++j) fbN[j] = ldB(sn, 0, j);
#pragma unroll
            for (int i = 0; i < 4; ++i) faN[i] = ldA(sn, 1, i);
            mma(fb[0], fa[0]);                                          // hi x hi
            advance_dma();
            sc = next;
            sd = sd + 1 == NST ? 0 : sd + 1;
            if (a.clk) { const long long t = clock64(); cTail += t - cT; cT = t; }
        }
        // ---- epilogue of tile ti (swapped operands): row = lane & 31, channels (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
        int t = runStart + local + ti * nloc;
        const int z = t / (a.nbm * a.nbn);
        t -= z * (a.nbm * a.nbn);
        const int mt = t / a.nbn, nt = t - mt * a.nbn;
        const int m0 = mt * BMT, n0 = nt * 256;
        const float inv = aInv * a.uInv[z];                              // (powers of two: the un-scaling is exact)
        const __amdgpu_buffer_rsrc_t srdO = __builtin_amdgcn_make_buffer_rsrc((void *)(a.out + (long long)z * a.T * a.N), 0, (int)((long long)a.T * a.N * 4), 0x00020000);
        if (!(a.var & 4) && !(DBG & 8)) {
            // (lane_pair_exchange: 64-byte pieces, even rows then odd rows)
            const bool odd = lane & 1;
            const int mE = m0 + wm * 128 + ((lane & 31) & ~1), cL = n0 + wn * 64 + 8 * (lane & 1) + rhalf;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned rowOff0 = (unsigned)((long long)(mE + i * 32) * a.N * 4) + (unsigned)cL * 4u;
                const unsigned rowOff1 = rowOff0 + (unsigned)a.N * 4u;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    f32x4 o[4];
                    const f32x16 c = acc[i][j] * inv;
                    lane_pair_exchange<0>(c, odd, o[0], o[1]);
                    lane_pair_exchange<1>(c, odd, o[2], o[3]);
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o[q]), srdO,
                                                               (int)(((q & 1) ? rowOff1 : rowOff0) + (unsigned)(j * 32 + (q >> 1) * 16) * 4u), 0, 0);
                }
            }
        } else
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // (exactly 32 stores per wave and tile, counted by the vmcnt arithmetic above: rows past T fall outside the
            //  descriptor, N is a multiple of 256)
            const int m = m0 + wm * 128 + i * 32 + (lane & 31);
            const unsigned rowOff = (unsigned)((long long)m * a.N * 4);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = n0 + wn * 64 + j * 32 + rhalf + 8 * q;
                    const unsigned off = rowOff + (unsigned)n * 4u;
                    const f32x4 v = f32x4{ acc[i][j][4 * q] * inv, acc[i][j][4 * q + 1] * inv, acc[i][j][4 * q + 2] * inv, acc[i][j][4 * q + 3] * inv };
                    if (DBG & 8) continue;                                   // (measurement: no stores)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), srdO, (int)off, 0, 0);
                }
        }
        if (DBG & 8) {                                                   // keep the accumulators alive: one store per tile
            f32x4 v = f32x4{ 0.f, 0.f, 0.f, 0.f };
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r & 3] += acc[i][j][r];
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), srdO, (int)((unsigned)((long long)(m0 + wm * 128 + (lane & 31)) * a.N * 4) + (unsigned)(n0 + wn * 64 + rhalf) * 4u), 0, 0);
        }
        if (a.clk) { const long long t = clock64(); cEpi += t - cT; cT = t; }
    }
    if (a.clk && lane == 0) {
        long long *c = a.clk + ((long long)blockIdx.x * 8 + wave) * 8;
        c[0] = cPre; c[1] = cVm; c[2] = cBar; c[3] = cTail; c[4] = (long long)myCount * nk; c[5] = cEpi; c[6] = cVm2; c[7] = myCount;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                             // vmcnt(0)
}

// ---------------------------------------------------------------------------------------------- fp32 activations, pairs formed in the kernel
//
// out[m][n] = bias[n] + sum_c f(in[m][c]) * W[n][c] - the 1x1 layers (f = identity, the producer's deferred GroupNorm(+ReLU), or
// its whole GroupNorm + ReLU + residual + ReLU epilogue), the GroupNorm partial sums of the output in the epilogue, and the batched
// GEMMs of a Winograd layer whose V arrives as fp32 (XL_CONV_SPLIT_ACT).  split_conv1x1_body of csrc/xl_gemm_split.hip with two
// activation planes instead of three and three term groups instead of six: a thread loads 8 channels of one row two K-steps
// ahead, applies scale (folded into the {scale, shift} table for the normalising forms), forms {hi, lo'} with v_cvt_pk_f16_f32
// (the residual a - hi is exact in fp32) and writes 2 x 16 bytes into the activation stage of the next K-step under the MFMAs of
// the second term group.  The weights run THREE K-steps ahead (ring of 4 stages by LDS-DMA), the activation loads two.
// LDS: weights 4 x BN x 64 | activations 2 x BM x 64 | statistics staging 64 NTH (overlaps activation stage 1) | fp64 partials
// 16 NTH | coefficient tables 2 x 8 KB | bias 4 KB | 1 KB scratch.
struct PairConvArgs {
    const float *in; const unsigned char *u; const float *bias; float *out;
    const float *coef; float normLo;                 // NORM: [B][C][2] {scale, shift}; lower clamp (0 = ReLU, -inf = none)
    double *stats; int HW, G, nchunks, B;            // statistics of the output (stats == nullptr: none); 16 channels per group
    int M, C, N, ldIn, ldOut, nbm, nbn;
    int tpi;                                         // > 0: tiles start at image boundaries, tpi = ceil(HW / BM) per image; 0: dense
    int Z; long long zIn, zOut;                      // Z > 1: Z independent products (in / out advance by zIn / zOut floats, the weights by N rows)
    int tile0, ntiles;                               // the launch's range of the (z, m-tile, n-tile) order
    const float *res; int ldRes;                     // RES (with NORM): a residual added behind the normalisation, then ReLU
    const float *uInv; const float *aScale;          // [Z] inverse weight scales; {s, 1 / s} of the activations
    int amaxShift;                                   // >= 0 (XL_CONV_PAIR_AMAX): aScale points at the float bits of max |operand source| and the
                                                     // scale is derived here: max 2^e in [2^(14 - shift), 2^(15 - shift))
    int var;                                         // measurement switches (unused)
};

template <bool NORM, bool ACC, int NW, int ZB, int BN, bool RES>               // ACC: out += result
__device__ __forceinline__ void pair_conv1x1_body(const PairConvArgs &a)
{
    static_assert(!RES || NORM, "the residual form is a normalise-on-load form");
    constexpr int NLD = RES ? 4 : 2;                                          // activation loads per thread and K-step
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
    typedef __attribute__((address_space(3))) void lds_void;
    constexpr int NTH = 64 * NW, BM = NTH / 2;                                // threads; tile rows (a thread = 8 channels of a row)
    constexpr int WN = BN / 64, WM = NW / WN, RI = BM / WM / 32;              // waves across columns / rows; 32-row blocks per wave
    static_assert(RI >= 1 && RI * WM * 32 == BM && WN * WM == NW, "tile shape");
    constexpr int NDMA = BN * kPB / 1024;                                     // DMA instructions per weight stage
    constexpr int PARTS = NTH * 8 / BN;                                       // statistics: threads per (group, slot)
    constexpr int kW = BN * kPB, kAS = BM * kPA;                              // one weight stage / one activation stage
    constexpr int NWS = 4;                                                    // weight stages
    constexpr int kCvA = NWS * kW, kCvStage1 = kCvA + kAS;                    // activation stages; statistics staging: 64 NTH bytes from stage 1
    constexpr int kCvPart = kCvStage1 + 64 * NTH;                             // fp64 partials [NTH / 16][16][2]
    constexpr int kCvCoef = kCvPart + 16 * NTH;                               // coefficient tables of two tiles, 8 KB each (C <= 512)
    constexpr int kCvBias = kCvCoef + 16384;                                  // bias[N <= 1024]
    constexpr int NS = 8 * RI;                                                // stores per wave and tile
    static_assert(64 * NTH >= kAS, "the statistics staging covers activation stage 1");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    // every wave issues two DMA instructions per K-step (straight-line code, the same counted waits in all waves); with 8 waves
    // on 128 columns those of waves 4 .. 7 read out of range and write their zeros into a scratch KB
    constexpr bool ALLDMA = NDMA >= 2 * NW;
    constexpr int kCvScratch = kCvBias + 4096;
    const bool dmaWave = ALLDMA || wave * 2 < NDMA;
    const int dmaBase = __builtin_amdgcn_readfirstlane(dmaWave ? wave * 2 * 1024 : kCvScratch);
    const int dmaStage = __builtin_amdgcn_readfirstlane(dmaWave ? kW : 0), dmaQ = __builtin_amdgcn_readfirstlane(dmaWave ? 1024 : 0);

    const int total = a.ntiles;
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3, nloc = gridDim.x >> 3;
    const int q8 = total >> 3, r8 = total & 7;
    const int runStart = a.tile0 + (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8);
    const int runLen = q8 + (xcd < r8 ? 1 : 0);
    const int myCount = runLen > local ? (runLen - local + nloc - 1) / nloc : 0;
    if (myCount == 0) return;
    auto tile_z = [&](int i) { return (runStart + local + i * nloc) / (a.nbm * a.nbn); };
    auto tile_at = [&](int i, int &m0, int &n0) {
        int t = runStart + local + i * nloc;
        t -= (t / (a.nbm * a.nbn)) * (a.nbm * a.nbn);
        const int mt = t / a.nbn;
        n0 = (t - mt * a.nbn) * BN;
        if (a.tpi) {
            const int n = mt / a.tpi;
            m0 = n * a.HW + (mt - n * a.tpi) * BM;
        } else m0 = mt * BM;
    };
    auto tile_rows = [&](int m0) {
        int rows = a.M - m0;
        if (a.tpi) rows = (m0 / a.HW + 1) * a.HW - m0;
        return rows < BM ? rows : BM;
    };

    const long long rowU = (long long)a.C * 4;
    const int nk = ZB == 2 ? 32 : a.C / 16;
    float aS, aInv;
    if (a.amaxShift >= 0) {
        const unsigned bits = reinterpret_cast<const unsigned *>(a.aScale)[0];
        int e = 0;
        if (bits >> 23) e = 14 - ((int)(bits >> 23) - 127) - a.amaxShift;
        e = e > 120 ? 120 : (e < -120 ? -120 : e);
        aS = __builtin_bit_cast(float, (unsigned)(e + 127) << 23);
        aInv = __builtin_bit_cast(float, (unsigned)(127 - e) << 23);
    } else { aS = a.aScale[0]; aInv = a.aScale[1]; }

    // ---- streams: weights by LDS-DMA three K-steps ahead of the multiplies (2 instructions per wave and step), activations into
    // registers two steps ahead (row tid >> 1, channels 8 (tid & 1) .. + 7 of the step; two register sets, by the parity of the step)
    __amdgpu_buffer_rsrc_t srdU = __builtin_amdgcn_make_buffer_rsrc((void *)a.u, 0, (int)(a.N * rowU), 0x00020000);
    __amdgpu_buffer_rsrc_t srdIn = __builtin_amdgcn_make_buffer_rsrc((void *)a.in, 0, 0, 0x00020000);
    __amdgpu_buffer_rsrc_t srdRes = __builtin_amdgcn_make_buffer_rsrc((void *)a.in, 0, 0, 0x00020000);
    const int arow = tid >> 1, ahalf = tid & 1;
    // (a lane's offsets inside a tile never change: the tile is the descriptor's base, and a tile past the end of my list gets
    //  an empty descriptor - every load of it reads zero)
    unsigned gB[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {                                     // 16 rows x 4 slots per instruction
        const int row = (wave * 2 + q) * 16 + (lane >> 2);
        gB[q] = dmaWave ? (unsigned)(row * (int)rowU + (((lane & 3) ^ swz(row)) * 16)) : OOB;
    }
    const unsigned gA = (unsigned)(arow * a.ldIn * 4 + ahalf * 32), gR = (unsigned)(arow * a.ldRes * 4 + ahalf * 32);
    int dTile = 0, dK = 0;                                            // position of the weight stream: tile of my list, K-step
    int aTile = 0, aK = 0;                                            // ... of the activation loads
    auto set_w_tile = [&](int i) {
        int m0, n0;
        tile_at(i < myCount ? i : 0, m0, n0);
        const int z = tile_z(i < myCount ? i : 0);
        srdU = __builtin_amdgcn_make_buffer_rsrc((void *)(a.u + ((long long)z * a.N + n0) * rowU), 0, i < myCount ? (int)(BN * rowU) : 0, 0x00020000);
    };
    auto set_a_tile = [&](int i) {
        int m0, n0;
        tile_at(i < myCount ? i : 0, m0, n0);
        const int rows = i < myCount ? tile_rows(m0) : 0;              // rows past the tile fall outside the descriptor
        srdIn = __builtin_amdgcn_make_buffer_rsrc((void *)(a.in + tile_z(i < myCount ? i : 0) * a.zIn + (long long)m0 * a.ldIn), 0, rows * a.ldIn * 4, 0x00020000);
        if constexpr (RES)
            srdRes = __builtin_amdgcn_make_buffer_rsrc((void *)(a.res + (long long)m0 * a.ldRes), 0, rows * a.ldRes * 4, 0x00020000);
    };
    auto dma_instr = [&](int q, int stage) {
        const int dst = ALLDMA ? stage * kW + (wave * 2 + q) * 1024 : dmaBase + stage * dmaStage + q * dmaQ;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srdU, (lds_void *)(dsm + dst), 16, (int)gB[q], dK * kPB, 0, 0);
    };
    u32x4 rA[2][2];                                                    // [K-step parity][half of my 8 channels]
    u32x4 rR[2][RES ? 2 : 1];                                          // ... of the residual
    auto load_a = [&](auto parTag) {
        constexpr int P = decltype(parTag)::value;
        rA[P][0] = __builtin_amdgcn_raw_buffer_load_b128(srdIn, (int)gA, aK * 64, 0);
        rA[P][1] = __builtin_amdgcn_raw_buffer_load_b128(srdIn, (int)(gA + 16u), aK * 64, 0);
        if constexpr (RES) {
            rR[P][0] = __builtin_amdgcn_raw_buffer_load_b128(srdRes, (int)gR, aK * 64, 0);
            rR[P][1] = __builtin_amdgcn_raw_buffer_load_b128(srdRes, (int)(gR + 16u), aK * 64, 0);
        }
    };
    auto advance_dma = [&]() {
        if (++dK == nk) { dK = 0; ++dTile; set_w_tile(dTile); }
    };
    auto advance_a = [&]() {
        if (++aK == nk) { aK = 0; ++aTile; set_a_tile(aTile); }
    };

    // ---- conversion, one K-step ahead of the multiplies
    unsigned wOff[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) wOff[p] = (unsigned)(kCvA + arow * kPA + (((2 * p + ahalf) ^ swz(arow)) * 16));
    int cTile = 0, cK = 0;
    unsigned cCoef = 0;                                              // LDS offset of my row's {scale, shift} run
    auto set_conv_tile = [&](int i) {
        if (NORM && i < myCount) {
            int m0, n0;
            tile_at(i, m0, n0);
            const int nLo = m0 / a.HW;
            const int split = (nLo + 1) * a.HW - m0;                 // first tile row of the second image
            cCoef = (unsigned)(kCvCoef + (i & 1) * 8192 + (arow >= split ? a.C * 8 : 0) + ahalf * 64);
        }
    };
    auto convert = [&](auto parTag) {                                  // registers of parity P -> activation stage P
        constexpr int P = decltype(parTag)::value;
        unsigned w[2][4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x4 x = __builtin_bit_cast(f32x4, rA[P][h]);
            if constexpr (NORM) {
                // (the table in LDS holds {scale, shift} * s: fmaf(x, scale s, shift s) = s fmaf(x, scale, shift) to the bit, s a power of two)
                const f32x4 c0 = *reinterpret_cast<const f32x4 *>(dsm + cCoef + cK * 128 + h * 32);
                const f32x4 c1 = *reinterpret_cast<const f32x4 *>(dsm + cCoef + cK * 128 + h * 32 + 16);
                x = f32x4{ fmaxf(fmaf(x[0], c0[0], c0[1]), a.normLo), fmaxf(fmaf(x[1], c0[2], c0[3]), a.normLo),
                           fmaxf(fmaf(x[2], c1[0], c1[1]), a.normLo), fmaxf(fmaf(x[3], c1[2], c1[3]), a.normLo) };
                if constexpr (RES) {                                   // gn_apply's epilogue: v += residual; v = max(v, 0)
                    const f32x4 r = __builtin_bit_cast(f32x4, rR[P][h]);
                    x = f32x4{ fmaxf(fmaf(r[0], aS, x[0]), 0.f), fmaxf(fmaf(r[1], aS, x[1]), 0.f), fmaxf(fmaf(r[2], aS, x[2]), 0.f), fmaxf(fmaf(r[3], aS, x[3]), 0.f) };
                }
            } else x *= aS;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const f32x2 v = f32x2{ x[2 * e], x[2 * e + 1] };
                const f16x2 hi = __builtin_convertvector(v, f16x2);
                const f16x2 lo = __builtin_convertvector((v - __builtin_convertvector(hi, f32x2)) * 2048.f, f16x2);
                w[0][2 * h + e] = __builtin_bit_cast(unsigned, hi);
                w[1][2 * h + e] = __builtin_bit_cast(unsigned, lo);
            }
        }
#pragma unroll
        for (int p = 0; p < 2; ++p)
            *reinterpret_cast<u32x4 *>(dsm + P * kAS + wOff[p]) = u32x4{ w[p][0], w[p][1], w[p][2], w[p][3] };
    };
    auto advance_conv = [&]() {
        if (++cK == nk) { cK = 0; ++cTile; set_conv_tile(cTile); }
    };
    // coefficient table of tile i: the {scale, shift} pairs of its (at most two) images, 4 C floats, into table i & 1 - times s
    const __amdgpu_buffer_rsrc_t srdCoef = __builtin_amdgcn_make_buffer_rsrc((void *)a.coef, 0, NORM ? a.B * a.C * 8 : 0, 0x00020000);
    constexpr int TPT = 512 / NTH;                                     // 16-byte table entries per thread (C <= 512)
    struct Tab { u32x4 v[TPT]; };
    auto load_table = [&](int i) -> Tab {
        Tab t;
#pragma unroll
        for (int e = 0; e < TPT; ++e) {
            const int idx = tid + e * NTH;
            unsigned off = OOB;
            if (i < myCount && idx < a.C) {
                int m0, n0;
                tile_at(i, m0, n0);
                off = (unsigned)(((long long)(m0 / a.HW) * a.C * 2 + idx * 4) * 4);
            }
            t.v[e] = __builtin_amdgcn_raw_buffer_load_b128(srdCoef, (int)off, 0, 0);
        }
        return t;
    };
    auto store_table = [&](int i, const Tab &t) {
#pragma unroll
        for (int e = 0; e < TPT; ++e) {
            const int idx = tid + e * NTH;
            if (idx < a.C) *reinterpret_cast<f32x4 *>(dsm + kCvCoef + (i & 1) * 8192 + idx * 16) = __builtin_bit_cast(f32x4, t.v[e]) * aS;
        }
    };

    // ---- fragments
    const int fr = lane & 31, kh = lane >> 5;
    unsigned slot[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) slot[p] = (unsigned)(((2 * p + kh) ^ swz(fr)) * 16);
    const unsigned frA = (unsigned)(kCvA + (wm * (32 * RI) + fr) * kPA), frB = (unsigned)((wn * 64 + fr) * kPB);
    f16x8 fa[2][RI], fb[2][2], fbs[2];
    f32x16 acc[RI][2];
    auto ldA = [&](int stage, int p, int i) { return *reinterpret_cast<const f16x8 *>(dsm + stage * kAS + frA + i * 32 * kPA + slot[p]); };
    auto ldB = [&](int stage, int p, int j) { return *reinterpret_cast<const f16x8 *>(dsm + stage * kW + frB + j * 32 * kPB + slot[p]); };
    auto mma = [&](const f16x8 (&b)[2], const f16x8 (&v)[RI]) {
#pragma unroll
        for (int i = 0; i < RI; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[j], v[i], acc[i][j], 0, 0, 0);
    };
    auto mma_col = [&](const f16x8 &b, const f16x8 (&v)[RI], int j) {   // one 32-column block
#pragma unroll
        for (int i = 0; i < RI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, v[i], acc[i][j], 0, 0, 0);
    };
    const int rhalf = kh * 4;
    // accumulators start at the bias, in the scaled domain: bias * s * (weight scale) - powers of two, so the sum is s sW times
    // what the unscaled order of operations gives, to the bit
    const float biasMul = a.Z > 1 ? 0.f : aS * (1.f / a.uInv[0]);
    auto init_acc = [&](int n0) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 b = *reinterpret_cast<const f32x4 *>(dsm + kCvBias + (n0 + wn * 64 + j * 32 + rhalf + 8 * q) * 4);
#pragma unroll
                for (int i = 0; i < RI; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][j][4 * q + e] = b[e];
            }
    };

    // ---- prologue: bias and the first two coefficient tables into LDS, steps 0 and 1 of the stream, step 0 converted
    for (int i = tid; i < a.nbn * BN; i += NTH) reinterpret_cast<float *>(dsm + kCvBias)[i] = (a.bias && i < a.N) ? a.bias[i] * biasMul : 0.f;
    if constexpr (NORM) {
        const Tab t0 = load_table(0), t1 = load_table(1);
        store_table(0, t0);
        store_table(1, t1);
    }
    set_w_tile(0);
    set_a_tile(0);
    set_conv_tile(0);
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    load_a(P0{});
    advance_a();
    dma_instr(0, 0); dma_instr(1, 0);
    advance_dma();
    __builtin_amdgcn_s_waitcnt(0x0070);                               // everything landed
    __syncthreads();                                                  // tables and bias visible
    convert(P0{});
    advance_conv();
    load_a(P1{});
    advance_a();
    dma_instr(0, 1); dma_instr(1, 1);
    advance_dma();
    dma_instr(0, 2); dma_instr(1, 2);
    advance_dma();
    // (once per launch: everything issued so far has landed, so the counted waits of the first steps - which assume the steady
    //  state's issue order - cannot under-wait; lgkmcnt(0): my writes of step 0)
    __builtin_amdgcn_s_waitcnt(0x0070);
    __builtin_amdgcn_s_barrier();
    fb[0][0] = ldB(0, 0, 0);
#pragma unroll
    for (int i = 0; i < RI; ++i) fa[1][i] = ldA(0, 1, i);
    int sc = 0, sd = 3;
    {
        int m0, n0;
        tile_at(0, m0, n0);
        init_acc(n0);
    }
    // one K-step; FIRST: one of the first two steps of a tile that follows another one (NS stores of its epilogue are in flight).
    // The weights of step kk + 1 were issued in step kk - 2; younger at the barrier of step kk are the NLD loads and 2 DMAs of
    // step kk - 1 and of step kk
    auto step = [&](auto firstTag, auto parTag) __attribute__((always_inline)) {
        constexpr int sa = decltype(parTag)::value;                   // parity of the K-step
        const int next = (sc + 1) & (NWS - 1);
        // (hs x lo' first, column block 0 first: its operands - the weights' hi fragment of block 0 and the activations' lo' fragments
        //  of this step - were read behind the barrier of the step before, into registers that were dead by then, so the matrix
        //  pipe has work while this step's other fragments arrive)
        fb[0][1] = ldB(sc, 0, 1);
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[1][j] = ldB(sc, 1, j);
#pragma unroll
        for (int i = 0; i < RI; ++i) fa[0][i] = ldA(sa, 0, i);
        fbs[0] = scale_hs(fb[0][0]);
        mma_col(fbs[0], fa[1], 0);                                     // hs x lo'
        fbs[1] = scale_hs(fb[0][1]);
        mma_col(fbs[1], fa[1], 1);
        load_a(parTag);                                               // step kk + 2: two steps until its conversion
        advance_a();
        dma_instr(0, sd);
        __builtin_amdgcn_sched_barrier(0);
        // lo x hi with the conversion of step kk + 1 threaded through it: one MFMA (8 passes, 32 cycles of the pipe) covers the issue
        // of a few VALU instructions of the same wave; the LDS writes go out before the last MFMAs of the group
        mma(fb[1], fa[0]);
        convert(std::integral_constant<int, sa ^ 1>{});               // (the compiler counts vmcnt for rA)
        {
            constexpr int nM = 2 * RI, valu = (RES ? 64 : (NORM ? 48 : 40)) / nM;
#pragma unroll
            for (int g = 0; g < nM; ++g) {
                if constexpr (NORM) { if (g == 0 || g == nM / 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); }  // coefficient reads of 4 channels
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, valu, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);                      // the two LDS writes
        }
        dma_instr(1, sd);
        __builtin_amdgcn_sched_barrier(0);
        advance_conv();
        constexpr int YOUNG = 2 * NLD + 4 + (decltype(firstTag)::value ? NS : 0);
        __builtin_amdgcn_s_waitcnt(0x0070 | (YOUNG & 15) | ((YOUNG >> 4) << 14));     // + lgkmcnt(0): my activation writes are done
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        mma_col(fb[0][0], fa[0], 0);                                   // hi x hi, column block 0 ...
        __builtin_amdgcn_sched_barrier(0);
        fb[0][0] = ldB(next, 0, 0);                                    // ... whose weight fragment then makes room for the next step's
#pragma unroll
        for (int i = 0; i < RI; ++i) fa[1][i] = ldA(sa ^ 1, 1, i);
        mma_col(fb[0][1], fa[0], 1);
        advance_dma();
        sc = next;
        sd = (sd + 1) & (NWS - 1);
        __builtin_amdgcn_sched_barrier(0);                            // (the vmcnt arithmetic above assumes this issue order)
    };
    auto epilogue = [&](int ti) __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
        // ---- tile ti (swapped operands): row = lane & 31, channels (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
        int m0, n0;
        tile_at(ti, m0, n0);
        {
            const float inv = aInv * a.uInv[a.Z > 1 ? tile_z(ti) : 0];    // un-scale (exact)
#pragma unroll
            for (int i = 0; i < RI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] *= inv;
        }
        if (a.stats != nullptr) {
            // GroupNorm partial sums of the output, as split_conv1x1_body: slot 0 = rows before `split`, slot 1 = the rest; per lane
            // fp32 over its rows x 8 channels of a group, fp64 over the lanes holding the group; one writer per (image, tile, group)
            const int nLo = m0 / a.HW;
            const int split = (nLo + 1) * a.HW - m0;
            f32x2 *sS = reinterpret_cast<f32x2 *>(dsm + kCvStage1);    // [4 pieces][2 slots][NTH threads]
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int gh = 0; gh < 2; ++gh) {
                    float s[2] = { 0.f, 0.f }, ss[2] = { 0.f, 0.f };
#pragma unroll
                    for (int i = 0; i < RI; ++i) {
                        const int row = wm * (32 * RI) + i * 32 + fr;
                        float t = 0.f, tt = 0.f;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float v = acc[i][j][8 * gh + e];
                            t += v;
                            tt = fmaf(v, v, tt);
                        }
                        const bool hi = row >= split, live = m0 + row < a.M && !(a.tpi && hi);
                        s[0] += (live && !hi) ? t : 0.f; ss[0] += (live && !hi) ? tt : 0.f;
                        s[1] += (live && hi) ? t : 0.f;  ss[1] += (live && hi) ? tt : 0.f;
                    }
#pragma unroll
                    for (int sl = 0; sl < 2; ++sl) sS[((j * 2 + gh) * 2 + sl) * NTH + tid] = f32x2{ s[sl], ss[sl] };
                }
            __builtin_amdgcn_s_waitcnt(0x0070 | 0xC00F);               // lgkmcnt(0)
            __builtin_amdgcn_s_barrier();
            {
                const int gs = tid / PARTS, part = tid % PARTS;        // (group of the tile, slot) x PARTS parts (8 lane blocks x WM)
                const int gt = gs >> 1, sl = gs & 1;
                const int src = ((part >> 3) * WN + (gt >> 2)) * 64 + (part & 7) * 8;
                const f32x2 *o = sS + ((gt & 3) * 2 + sl) * NTH + src;
                double s1 = 0.0, s2 = 0.0;
#pragma unroll
                for (int e = 0; e < 8; ++e) { s1 += (double)o[e][0]; s2 += (double)o[e][1]; }
                double *sC = reinterpret_cast<double *>(dsm + kCvPart);
                sC[tid * 2] = s1; sC[tid * 2 + 1] = s2;
            }
            __builtin_amdgcn_s_waitcnt(0x0070 | 0xC00F);
            __builtin_amdgcn_s_barrier();
            if (tid < 2 * (BN / 16)) {
                const int gt = tid >> 1, sl = tid & 1;
                const int n = nLo + sl;
                const int firstRow = sl ? split : 0;
                const int g = (n0 >> 4) + gt;
                if (n < a.B && m0 + firstRow < a.M && (sl == 0 || (split < BM && !a.tpi)) && g < a.G) {
                    const double *sC = reinterpret_cast<const double *>(dsm + kCvPart) + tid * (2 * PARTS);
                    double s1 = 0.0, s2 = 0.0;
                    for (int e = 0; e < PARTS; ++e) { s1 += sC[2 * e]; s2 += sC[2 * e + 1]; }
                    const int k = a.tpi ? (m0 - n * a.HW) / BM                            // tile index within the image
                                        : m0 / BM - (int)(((long long)n * a.HW) / BM);
                    double *o = a.stats + (((long long)n * a.nchunks + k) * a.G + g) * 2;
                    o[0] = s1; o[1] = s2;
                }
            }
        }
        if constexpr (NORM) {                                          // (waits for everything older than the table)
            const Tab tab = load_table(ti + 2);
            store_table(ti + 2, tab);
        }
        // (every wave issues exactly NS stores per tile - the vmcnt arithmetic of the next step counts them)
        const __amdgpu_buffer_rsrc_t srdO = __builtin_amdgcn_make_buffer_rsrc((void *)(a.out + tile_z(ti) * a.zOut + (long long)m0 * a.ldOut), 0,
                                                                              tile_rows(m0) * a.ldOut * 4, 0x00020000);
        {
            // (lane_pair_exchange: 64-byte pieces, even rows then odd rows; rows past the tile fall outside the descriptor)
            const bool odd = lane & 1;
            const unsigned colL = (unsigned)(n0 + wn * 64 + 8 * (lane & 1) + rhalf) * 4u;
#pragma unroll
            for (int i = 0; i < RI; ++i) {
                const unsigned rowOff0 = (unsigned)((wm * (32 * RI) + i * 32 + (fr & ~1)) * a.ldOut * 4) + colL;
                const unsigned rowOff1 = rowOff0 + (unsigned)a.ldOut * 4u;
                f32x4 old[2][4];
                if constexpr (ACC) {                                   // (8 loads in flight per 32-row block; rows past the tile read 0)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            old[j][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                srdO, (int)(((q & 1) ? rowOff1 : rowOff0) + (unsigned)(j * 32 + (q >> 1) * 16) * 4u), 0, 0));
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    f32x4 o[4];
                    lane_pair_exchange<0>(acc[i][j], odd, o[0], o[1]);
                    lane_pair_exchange<1>(acc[i][j], odd, o[2], o[3]);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4 v = o[q];
                        if constexpr (ACC) v += old[j][q];
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), srdO,
                                                               (int)(((q & 1) ? rowOff1 : rowOff0) + (unsigned)(j * 32 + (q >> 1) * 16) * 4u), 0, 0);
                    }
                }
            }
        }
        if (ti + 1 < myCount) {
            tile_at(ti + 1, m0, n0);
            init_acc(n0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto tile_steps = [&](auto firstTag) __attribute__((always_inline)) {
        step(firstTag, P0{});
        step(firstTag, P1{});
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

template <bool NORM, bool ACC = false, int NW = 8, int ZB = 0, int BN = (NW == 8 ? 256 : 128)>
__global__ __launch_bounds__(64 * NW)
void pair_conv1x1_kernel(PairConvArgs a)
{
    pair_conv1x1_body<NORM, ACC, NW, ZB, BN, false>(a);
}

template <int NW = 8, int BN = (NW == 8 ? 256 : 128)>
__global__ __launch_bounds__(64 * NW)
void pair_conv1x1_res_kernel(PairConvArgs a)
{
    pair_conv1x1_body<true, false, NW, 0, BN, true>(a);
}

}  // namespace

// XL_OP_CONV with XL_CONV_SPLIT_BF16 | XL_CONV_SPLIT_IL | XL_CONV_PAIR_F16, nchunks2 = Z > 1, no XL_CONV_SPLIT_ACT: in = V as
// activation pairs [Z][T][Cin/16][2][16] fp16, w = weight pairs [Z][Cout][Cin/16][2][16] fp16 + 2 Z floats, out = M fp32
// [Z][T][Cout], scale = {s, 1 / s}.
static int xl_run_pair_gemm(const xl_op &op, hipStream_t st)
{
    const int T = op.B * op.Ho * op.Wo, Z = op.nchunks2;
    if (op.ksize != 1 || op.stride != 1 || Z < 1 || op.Cin % 16 != 0 || op.Cout % 256 != 0 || op.ld_in != op.Cin || op.ld_out != op.Cout ||
        op.bias || op.stats || (op.flags & (XL_CONV_ACCUMULATE | XL_CONV_NORM_IN | XL_CONV_M_TILE_MAJOR | XL_CONV_PAIR_AMAX)) || !op.in || !op.w || !op.out || !op.scale ||
        (((uintptr_t)op.in | (uintptr_t)op.w | (uintptr_t)op.out) & 15))
        return XL_ERR_ARG;
    // 32-bit offsets inside one GEMM's operands / result (each z has its own buffer descriptor)
    if ((long long)(T + 256) * op.Cout * 4 >= 0xffffffffLL || (long long)T * op.Cin * 4 >= 0x7fffffffLL || (long long)op.Cout * op.Cin * 4 >= 0x7fffffffLL ||
        (long long)T * op.Cout * 4 >= 0x7fffffffLL) return XL_ERR_ARG;
    PairArgs a;
    a.v = (const unsigned char *)op.in; a.u = (const unsigned char *)op.w; a.out = (float *)op.out;
    a.uInv = reinterpret_cast<const float *>(a.u + (long long)Z * op.Cout * op.Cin * 4) + Z;
    a.aScale = (const float *)op.scale;
    a.T = T; a.C = op.Cin; a.N = op.Cout; a.Z = Z;
    static const bool pingPong = getenv("XL_PAIR_PP") && atoi(getenv("XL_PAIR_PP")) != 0;
    const bool pp = pingPong && op.Cin == 512;
    a.nbm = pp ? (T + 127) / 128 : (T + 255) / 256; a.nbn = op.Cout / 256;
    a.clk = nullptr;
    static const int pvar = getenv("XL_PAIR_VAR") ? atoi(getenv("XL_PAIR_VAR")) : 0;
    a.var = pvar;
    const size_t lds = pp ? 3 * (size_t)(128 * kPA + 256 * kPB)     // 72 KB: two workgroups per CU
                          : 4 * (size_t)(256 * kPA + 256 * kPB);    // 128 KB: one workgroup per CU
    auto kernel = pp ? pair_gemm_kernel<512, 0, 1> : op.Cin == 512 ? pair_gemm_kernel<512> : pair_gemm_kernel<0>;
    static const int dbg = getenv("XL_PAIR_DBG") ? atoi(getenv("XL_PAIR_DBG")) : 0;
    if (dbg && op.Cin == 512 && !pp)
        kernel = dbg == 1 ? pair_gemm_kernel<512, 1> : dbg == 2 ? pair_gemm_kernel<512, 2> : dbg == 3 ? pair_gemm_kernel<512, 3> : dbg == 4 ? pair_gemm_kernel<512, 4>
               : dbg == 7 ? pair_gemm_kernel<512, 7> : dbg == 8 ? pair_gemm_kernel<512, 8> : dbg == 15 ? pair_gemm_kernel<512, 15> : pair_gemm_kernel<512, 5>;
    static XlLdsLimit configured[4];
    int cfgDev;
    const int slot = pp ? 3 : dbg ? 2 : op.Cin == 512 ? 1 : 0;
    if (configured[slot].needs(lds, &cfgDev)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return XL_ERR_HIP;
        configured[slot].done(lds, cfgDev);
    }
    const int nwg = a.nbm * a.nbn * Z;
    int grid = pp ? 512 : 256;
    if (grid > ((nwg + 7) & ~7)) grid = (nwg + 7) & ~7;
    static const bool clkDbg = getenv("XL_PAIR_CLK") != nullptr;
    if (clkDbg && hipMalloc(&a.clk, sizeof(long long) * 64 * grid) != hipSuccess) return XL_ERR_HIP;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(pp ? 256 : 512), lds, st, a);
    if (clkDbg) {
        std::vector<long long> h((size_t)64 * grid);
        if (hipStreamSynchronize(st) == hipSuccess && hipMemcpy(h.data(), a.clk, sizeof(long long) * 64 * grid, hipMemcpyDeviceToHost) == hipSuccess) {
            double v[2][4] = { { 0 } }; double steps = 0, epi = 0, vm2 = 0, tiles = 0;
            for (int i = 0; i < 8 * grid; ++i) { const int r = (i & 7) < 4 ? 0 : 1; for (int c = 0; c < 4; ++c) v[r][c] += h[8 * (size_t)i + c]; steps += h[8 * (size_t)i + 4];
                                                 epi += h[8 * (size_t)i + 5]; vm2 += h[8 * (size_t)i + 6]; tiles += h[8 * (size_t)i + 7]; }
            fprintf(stderr, "[pair clk] per wave and tile: epilogue (un-scale + store issue) %.0f ticks, DMA wait of the first two K-steps behind it %.0f\n", epi / tiles, vm2 / tiles);
            steps /= 2;                                         // per wave group
            for (int r = 0; r < 2; ++r)
                fprintf(stderr, "[pair clk] waves %d-%d, ticks per K-step: reads + terms 0-1 + DMA issue %.0f, wait for DMAs %.0f, barrier %.0f, prefetch + term 2 %.0f\n",
                        4 * r, 4 * r + 3, v[r][0] / steps, v[r][1] / steps, v[r][2] / steps, v[r][3] / steps);
        }
        (void)hipFree(a.clk);
    }
    return XL_OK;
}


// fp32 activations: the launcher of csrc/xl_gemm_split.hip's 1x1 / XL_CONV_SPLIT_ACT forms with the pair kernels (same tile forms,
// same statistics rows: op.reserved_i as there)
template <int NW, int BN>
static int launch_pair_conv1x1(const xl_op &op, PairConvArgs a, bool norm, int Z, hipStream_t st, int tileFrom = 0, int tileCount = -1)
{
    constexpr int BM = 32 * NW;
    const long long M = a.M;
    const bool perImage = op.reserved_i < 0;         // tiles start at image boundaries (reserved_i = -256 / -128)
    a.tpi = perImage ? (a.HW + BM - 1) / BM : 0;
    a.nbm = perImage ? op.B * a.tpi : (int)((M + BM - 1) / BM);
    a.nbn = (op.Cout + BN - 1) / BN;
    const size_t lds = 4 * BN * kPB + BM * kPA + 64 * (64 * NW) + 16 * (64 * NW) + 16384 + 4096 + 1024;
    const bool accumulate = (op.flags & XL_CONV_ACCUMULATE) != 0;
    const bool resid = norm && a.res != nullptr;
    const bool dominant = Z > 1 && op.Cin == 512 && op.Cout == 512 && !accumulate && !norm;
    const void *fn = accumulate ? reinterpret_cast<const void *>(pair_conv1x1_kernel<false, true, NW, 0, BN>)
                   : resid ? reinterpret_cast<const void *>(pair_conv1x1_res_kernel<NW, BN>)
                   : norm ? reinterpret_cast<const void *>(pair_conv1x1_kernel<true, false, NW, 0, BN>)
                   : dominant ? reinterpret_cast<const void *>(pair_conv1x1_kernel<false, false, NW, 2, BN>)
                   : Z > 1 ? reinterpret_cast<const void *>(pair_conv1x1_kernel<false, false, NW, 1, BN>)
                           : reinterpret_cast<const void *>(pair_conv1x1_kernel<false, false, NW, 0, BN>);
    static XlLdsLimit configured[6];
    int cfgDev;
    const int slot = accumulate ? 2 : (resid ? 5 : (norm ? 1 : (dominant ? 4 : (Z > 1 ? 3 : 0))));
    if (configured[slot].needs(lds, &cfgDev)) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return XL_ERR_HIP;
        configured[slot].done(lds, cfgDev);
    }
    const int nwg = tileCount < 0 ? a.nbm * a.nbn * Z : tileCount;
    a.tile0 = tileFrom; a.ntiles = nwg;
    int grid = 256;                                   // persistent: one workgroup per CU
    if (grid > ((nwg + 7) & ~7)) grid = (nwg + 7) & ~7;
    if (accumulate) hipLaunchKernelGGL((pair_conv1x1_kernel<false, true, NW, 0, BN>), dim3(grid), dim3(64 * NW), lds, st, a);
    else if (resid) hipLaunchKernelGGL((pair_conv1x1_res_kernel<NW, BN>), dim3(grid), dim3(64 * NW), lds, st, a);
    else if (norm) hipLaunchKernelGGL((pair_conv1x1_kernel<true, false, NW, 0, BN>), dim3(grid), dim3(64 * NW), lds, st, a);
    else if (dominant) hipLaunchKernelGGL((pair_conv1x1_kernel<false, false, NW, 2, BN>), dim3(grid), dim3(64 * NW), lds, st, a);
    else if (Z > 1) hipLaunchKernelGGL((pair_conv1x1_kernel<false, false, NW, 1, BN>), dim3(grid), dim3(64 * NW), lds, st, a);
    else hipLaunchKernelGGL((pair_conv1x1_kernel<false, false, NW, 0, BN>), dim3(grid), dim3(64 * NW), lds, st, a);
    return XL_OK;
}

static int xl_run_pair_conv1x1(const xl_op &op, hipStream_t st)
{
    const long long M = (long long)op.B * op.Ho * op.Wo;
    const int HW = op.Ho * op.Wo;
    const bool norm = (op.flags & XL_CONV_NORM_IN) != 0;
    const bool small = op.reserved_i == 128 || op.reserved_i == -128;
    const int BM = small ? 128 : 256;
    const int Z = op.nchunks2 > 1 ? op.nchunks2 : 1;
    if (Z > 1 && (op.bias || op.stats || norm || op.reserved_i < 0 || op.ld_in != op.Cin || op.ld_out != op.Cout)) return XL_ERR_ARG;
    if (op.ksize != 1 || op.stride != 1 || op.Cin % 32 != 0 || op.Cout % 256 != 0 || op.Cout > 1024 || op.ld_in < op.Cin ||
        op.ld_out < op.Cout || (op.ld_in & 3) || (op.ld_out & 3) || ((op.flags & XL_CONV_ACCUMULATE) && (norm || Z > 1 || op.stats)) || !op.in ||
        !op.w || !op.out || !op.scale || (((uintptr_t)op.in | (uintptr_t)op.out | (uintptr_t)op.w) & 15) || M >= 0x7fffffffLL ||
        (long long)op.Cout * op.Cin * 4 >= 0x7fffffffLL || 256LL * op.ld_in * 4 >= 0x7fffffffLL || 256LL * op.ld_out * 4 >= 0x7fffffffLL)
        return XL_ERR_ARG;
    if (norm && (!op.aux2 || op.Cin > 512 || HW < BM)) return XL_ERR_ARG;
    if (op.stats && (op.groups <= 0 || op.Cout != 16 * op.groups || HW < BM || op.nchunks < (HW + BM - 1) / BM + 1)) return XL_ERR_ARG;
    if (op.reserved_i < 0 && HW < BM) return XL_ERR_ARG;
    PairConvArgs a;
    a.in = (const float *)op.in; a.u = (const unsigned char *)op.w; a.bias = (const float *)op.bias; a.out = (float *)op.out;
    a.coef = (const float *)op.aux2;
    a.normLo = (op.flags & XL_CONV_NORM_RELU) ? 0.f : -__builtin_inff();
    a.stats = (double *)op.stats; a.HW = HW; a.G = op.groups; a.nchunks = op.nchunks; a.B = op.B;
    a.M = (int)M; a.C = op.Cin; a.N = op.Cout; a.ldIn = op.ld_in; a.ldOut = op.ld_out;
    a.Z = Z; a.zIn = M * op.ld_in; a.zOut = M * op.ld_out;
    a.res = nullptr; a.ldRes = 0;
    a.uInv = reinterpret_cast<const float *>(a.u + (long long)Z * op.Cout * op.Cin * 4) + Z;
    a.aScale = (const float *)op.scale;
    a.amaxShift = (op.flags & XL_CONV_PAIR_AMAX) ? (Z > 1 ? 8 : 0) : -1;
    a.var = 0;
    if (op.flags & XL_CONV_NORM_ADD) {
        if (!norm || !(op.flags & XL_CONV_NORM_RELU) || !op.aux || op.ld_aux < op.Cin || (op.ld_aux & 3) || ((uintptr_t)op.aux & 15) || Z > 1 ||
            256LL * op.ld_aux * 4 >= 0x7fffffffLL) return XL_ERR_ARG;
        a.res = (const float *)op.aux; a.ldRes = op.ld_aux;
    }
    if (op.flags & XL_CONV_M_TILE_MAJOR) {
        if (Z < 2 || op.ld_out != op.Cout || 256LL * Z * op.Cout * 4 >= 0x7fffffffLL) return XL_ERR_ARG;
        a.zOut = op.Cout; a.ldOut = Z * op.Cout;
    }
    a.tpi = 0; a.nbm = 0; a.nbn = 0;
    if (small) return launch_pair_conv1x1<4, 128>(op, a, norm, Z, st);
    if (op.reserved_i == 192) return launch_pair_conv1x1<8, 128>(op, a, norm, Z, st);
    if (op.reserved_i == 384) {
        const int nbig = (int)((M + 255) / 256) * (op.Cout / 256) * Z, full = nbig / 256 * 256, rem = nbig - full;
        if (full > 0 && rem > 0 && 2 * rem <= 256) {
            const int rc = launch_pair_conv1x1<8, 256>(op, a, norm, Z, st, 0, full);
            return rc != XL_OK ? rc : launch_pair_conv1x1<8, 128>(op, a, norm, Z, st, 2 * full, 2 * rem);
        }
        if (full == 0 && 2 * rem <= 256) return launch_pair_conv1x1<8, 128>(op, a, norm, Z, st);
    }
    return launch_pair_conv1x1<8, 256>(op, a, norm, Z, st);
}

int xl_run_pair(const xl_op &op, hipStream_t st)
{
    if (!(op.flags & XL_CONV_SPLIT_IL)) return XL_ERR_ARG;
    if (op.nchunks2 > 1 && !(op.flags & XL_CONV_SPLIT_ACT)) return xl_run_pair_gemm(op, st);
    return xl_run_pair_conv1x1(op, st);
}
