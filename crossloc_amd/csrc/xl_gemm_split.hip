// crossloc_hip: batched GEMM with fp32 operands split into three bf16 terms (opt-in, inference plans).
//
//   M_z[t][o] = sum_c V_z[t][c] * U_z[o][c]          z = 0 .. Z-1 (the frequencies of a Winograd layer)
//
// Every fp32 value a is stored as a1 + a2 + a3 with a1 = bf16(a), a2 = bf16(a - a1), a3 = bf16(a - a1 - a2): 24
// mantissa bits, exact.  The six term pairs a1b1, a1b2, a2b1, a1b3, a2b2, a3b1 are multiplied on the bf16 matrix pipe
// (v_mfma_f32_32x32x16_bf16: a bf16 x bf16 product is exact in fp32) and accumulated in fp32; the pairs left out are
// below 2^-24 of the leading one, the order of fp32's own rounding.  Six passes on a pipe that sustains 16x the fp32
// MFMA rate (2482 vs 154.6 TFLOP/s measured) = 2.7x the fp32 MFMA ceiling for the same result to fp32 accuracy.
//
// Operand layout: three planes each, plane p of V = [Z][T][C] bf16 at vPlane*p, of U = [Z][N][C] bf16 at uPlane*p.
// Workgroup = 128 tiles x 128 output channels, 4 waves of 64 x 64; K-step = 32 channels: global -> registers ->
// LDS (one buffer; the loads of the next step are in flight during the MFMAs of the current one).  LDS rows are 64 B
// (32 bf16), 16-byte slots XOR-swizzled by (row >> 1) & 3 so that the 32 rows a fragment read touches spread over all
// banks.  The weight fragment is the MFMA row operand: an accumulator quad is 4 consecutive output channels of one
// tile row, stored with one dwordx4.
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
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct SplitArgs {
    const uint16_t *v, *u;      // plane 0 of the activations / weights
    float *out;                 // [Z][T][N] fp32
    long long vPlane, uPlane;   // elements between planes
    int T, C, N, Z, nbm, nbn;
    unsigned vBytes, uBytes, outBytes;     // extents of one plane / of the output for the buffer descriptors
};

__device__ __forceinline__ int xcd_remap(int b, int nwg)
{
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = b & 7, local = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
}

constexpr int kBM = 128, kBN = 128, kBK = 32;           // tile; K-step in channels
constexpr int kRowB = kBK * 2;                          // bytes per LDS row (64)
constexpr int kPlaneA = kBM * kRowB, kPlaneB = kBN * kRowB;

__global__ __launch_bounds__(256, 2)
void split_gemm_kernel(SplitArgs a)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[3 * kPlaneA + 3 * kPlaneB];     // 48 KB
    unsigned char *sA = smem, *sB = smem + 3 * kPlaneA;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    int tile = xcd_remap(blockIdx.x, a.nbm * a.nbn * a.Z);
    const int z = tile / (a.nbm * a.nbn);
    tile -= z * (a.nbm * a.nbn);
    const int mt = tile / a.nbn, nt = tile - mt * a.nbn;
    const int m0 = mt * kBM, n0 = nt * kBN;

    constexpr unsigned OOB = 0x80000000u;
    __amdgpu_buffer_rsrc_t srdV[3], srdU[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        srdV[p] = __builtin_amdgcn_make_buffer_rsrc((void *)(a.v + p * a.vPlane), 0, (int)a.vBytes, 0x00020000);
        srdU[p] = __builtin_amdgcn_make_buffer_rsrc((void *)(a.u + p * a.uPlane), 0, (int)a.uBytes, 0x00020000);
    }
    // loader: thread -> (row = tid >> 2 (+64), 16-byte slot = tid & 3) of a 128 x 64 B plane tile, 2 loads per plane
    const int lrow = tid >> 2, lslot = tid & 3;
    unsigned gA[2], gB[2], sOff[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int row = lrow + 64 * h;
        const int m = m0 + row, n = n0 + row;
        gA[h] = m < a.T ? (unsigned)(((long long)z * a.T + m) * a.C * 2 + lslot * 16) : OOB;
        gB[h] = n < a.N ? (unsigned)(((long long)z * a.N + n) * a.C * 2 + lslot * 16) : OOB;
        sOff[h] = (unsigned)(row * kRowB + ((lslot ^ ((row >> 1) & 3)) * 16));
    }
    u32x4 rA[3][2], rB[3][2];
    auto load_regs = [&](int kk) {
        const int kb = kk * kRowB;                               // byte offset of the K-step inside a row
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                rA[p][h] = __builtin_amdgcn_raw_buffer_load_b128(srdV[p], (int)gA[h], kb, 0);
                rB[p][h] = __builtin_amdgcn_raw_buffer_load_b128(srdU[p], (int)gB[h], kb, 0);
            }
    };
    auto store_lds = [&]() {
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                *reinterpret_cast<u32x4 *>(sA + p * kPlaneA + sOff[h]) = rA[p][h];
                *reinterpret_cast<u32x4 *>(sB + p * kPlaneB + sOff[h]) = rB[p][h];
            }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment addressing: lane -> row lane & 31 of a 32-row block, k-half lane >> 5 (8 bf16 = one 16-byte slot)
    const int fr = lane & 31, kh = lane >> 5;
    unsigned fOffA[2][2], fOffB[2][2];                            // [32-row block][16-channel chunk of the K-step]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int ra = wm * 64 + i * 32 + fr, rb = wn * 64 + i * 32 + fr;
            fOffA[i][c] = (unsigned)(ra * kRowB + (((2 * c + kh) ^ ((ra >> 1) & 3)) * 16));
            fOffB[i][c] = (unsigned)(rb * kRowB + (((2 * c + kh) ^ ((rb >> 1) & 3)) * 16));
        }

    const int nk = a.C / kBK;
    load_regs(0);
    for (int kk = 0; kk < nk; ++kk) {
        if (kk) __syncthreads();                                  // every wave has read the previous K-step
        store_lds();
        __syncthreads();
        if (kk + 1 < nk) load_regs(kk + 1);
        // fragments of both 16-channel chunks are read up front (the second set lands under the MFMAs of the first);
        // MFMAs in term-major order: consecutive instructions hit different accumulators, smallest terms first
        bf16x8 fa[2][3][2], fb[2][3][2];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    fa[c][p][i] = *reinterpret_cast<const bf16x8 *>(sA + p * kPlaneA + fOffA[i][c]);
                    fb[c][p][i] = *reinterpret_cast<const bf16x8 *>(sB + p * kPlaneB + fOffB[i][c]);
                }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            constexpr int PU[6] = { 2, 1, 0, 1, 0, 0 }, PV[6] = { 0, 1, 2, 0, 1, 0 };   // u3v1 u2v2 u1v3 u2v1 u1v2 u1v1
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[c][PU[t]][j], fa[c][PV[t]][i], acc[i][j], 0, 0, 0);
        }
    }

    // epilogue (swapped operands): tile row = lane & 31, channels (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of a 32-block
    const __amdgpu_buffer_rsrc_t srdO = __builtin_amdgcn_make_buffer_rsrc((void *)a.out, 0, (int)a.outBytes, 0x00020000);
    const int rhalf = (lane >> 5) * 4;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + wm * 64 + i * 32 + (lane & 31);
        const unsigned rowOff = m < a.T ? (unsigned)((((long long)z * a.T + m) * a.N) * 4) : OOB;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + wn * 64 + j * 32 + rhalf + 8 * q;
                const unsigned off = (rowOff != OOB && n < a.N) ? rowOff + (unsigned)n * 4u : OOB;
                const f32x4 v = f32x4{ acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3] };
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), srdO, (int)off, 0, 0);
            }
    }
}


// ---------------------------------------------------------------------------------------------- 256 x 256 persistent form
//
// What bounds the 128 x 128 form above is operand delivery, not the matrix pipe: a tile moves 48 KB of split operands out
// of L2 per 1.05 MFLOP (fp32-equivalent), in 64-byte requests: 10 GB and 1.9e8 L2 requests per 512-channel layer launch
// (TCC_REQ), 8 TB/s at the measured 1.27 ms while the MFMA pipe is 58 % busy, and a third of the LDS cycles are bank
// conflicts of the ds_write staging.  This form:
//   * 256 x 256 tiles (8 waves of 128 x 64): half the operand bytes and L2 requests per FLOP;
//   * an operand layout with the three planes of a 16-channel chunk next to each other,
//         V[z][t][c / 16][plane][c % 16]            (96 contiguous bytes per row and K-step)
//     so a K-step of 16 channels - exactly one k-depth of v_mfma_f32_32x32x16_bf16 - is one piece per row;
//   * operands by LDS-DMA into a ring of 3 stages of (256 + 256) rows x 96 B = 48 KB (144 KB: one workgroup per CU), the
//     stream two K-steps ahead of the multiplies.  LDS rows are 6 slots of 16 bytes, rotated by one slot on rows with bit 3
//     set (rows r and r + 8 would otherwise hit the same banks: 96 r mod 256) - applied to the source offset of the lane
//     that fills a slot and to the fragment reads: SQ_LDS_BANK_CONFLICT = 0;
//   * persistent: ONE workgroup per CU walks its tiles and the operand stream runs across tile boundaries.  A
//     launch-per-tile version of the same loop lived 178k shader ticks per tile of which 98k are MFMA time: prologue 13k,
//     epilogue (until the stores are acknowledged) 19k, dispatch of the next workgroup 19k (XL_SPLIT_CLK);
//   * ONE barrier per K-step, a bare s_barrier behind a counted s_waitcnt: __syncthreads() carries a workgroup fence for
//     which the compiler waits for EVERY outstanding LDS-DMA (vmcnt(0)) - with it the prefetch depth is zero whatever the
//     ring size, which is why rings of 2, 3 and 4 + 2 stages all measured 1.18 ms before this was found in the ISA;
//   * the 6 DMA instructions a wave issues per step sit one behind each of the six term groups (an LDS-DMA costs its wave
//     100-200 ticks of issue in which it multiplies nothing; issued together at the top of a step all eight waves stall
//     at once), and the first term's operands of the next step are prefetched behind the barrier, so the matrix pipe
//     never waits for LDS data right after it.
// Measured (512 -> 512 layer, 44 frames, 64 GEMMs): 1.05-1.11 ms stand-alone on random data = 200-211 TFLOP/s
// fp32-equivalent (1.20-1.27 PFLOP/s on the bf16 pipe), 1.03 ms inside the network; fp32-MFMA kernel 1.69 ms.  With every
// DMA and LDS read removed the same MFMA + barrier stream takes 0.92 ms.  What the kernel runs into is the power limit, not
// a schedule: PMC of the profiled (serialised) launches shows the matrix pipe 64 % busy at an average 1.94 GHz, the
// back-to-back launches inside the network are 6 % slower again, and changes that remove stalls without removing work
// leave the time where it is - e.g. issue priorities that fall through a barrier-to-barrier cycle (s_setprio 3 3 2 2 1 1
// over the six term groups) make the two waves of a SIMD advance in step and cut the older waves' barrier wait from 1800
// to 700 of a K-step's 4300 ticks (XL_SPLIT_CLK), yet 1.171 vs 1.169 ms.  busy x clock stays near 1.2 GHz-equivalent of the
// 2.4 GHz the datasheet peak assumes; the lever left is energy per FLOP (fewer LDS reads per MFMA: 128 x 128 per wave).
constexpr int kIUnit = 96;                              // bytes per row and K-step: 3 planes x 16 bf16
constexpr int kIOperand = 256 * kIUnit;                 // one operand of one stage: 24 KB

struct SplitArgs2 {
    const uint16_t *v, *u;      // interleaved layout, [Z][T][C/16][3][16] and [Z][N][C/16][3][16]
    float *out;                 // [Z][T][N] fp32
    int T, C, N, Z, nbm, nbn;
    long long *clk;             // diagnostics (XL_SPLIT_CLK=1): per-wave shader-tick sums of the four phases of a K-step
};

// vmcnt bookkeeping (in order, per wave): at the barrier of step s the operands of step s+1 must have landed; younger
// than those are the 5 DMAs of step s+2 issued so far - and, in the first step of a tile, the 32 stores of the tile before.
template <int CT>                                                      // compile-time channel count (0: a.C)
__global__ __launch_bounds__(512)
void split_gemm_persist_kernel(SplitArgs2 a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
    typedef __attribute__((address_space(3))) void lds_void;
    constexpr int kStage = 2 * kIOperand;                             // activations then weights: 48 KB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;                          // 2 x 4 waves of 128 x 64

    // tiles of this workgroup: XCD x (= block % 8) owns a contiguous run of the (z, m-tile, n-tile) order, its
    // workgroups take every nloc-th tile of the run, so the workgroups of an XCD work on neighbouring tiles at any time
    const int total = a.nbm * a.nbn * a.Z;
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3, nloc = gridDim.x >> 3;
    const int q8 = total >> 3, r8 = total & 7;
    const int runStart = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int runLen = q8 + (xcd < r8 ? 1 : 0);
    const int myCount = runLen > local ? (runLen - local + nloc - 1) / nloc : 0;
    if (myCount == 0) return;

    constexpr unsigned OOB = 0x80000000u;
    const long long rowB = (long long)a.C * 6;                        // bytes per operand row
    const int nk = CT ? CT / 16 : a.C / 16;

    // ---- operand stream
    // every wave issues 6 of the 48 DMA instructions of a step (3 of each operand), one after each term group.  (Letting
    // waves 0-3 - the older ones on their SIMDs, which win the MFMA arbitration and then sit at the barrier - issue all
    // of them was measured slower: 1.09 vs 1.05 ms.)
    __amdgpu_buffer_rsrc_t srdV, srdU;
    unsigned gA[3], gB[3];
    int dTile = 0, dK = 0;                                            // position of the stream: tile of my list, K-step
    auto set_dma_tile = [&](int i) {
        if (i < myCount) {
            int t = runStart + local + i * nloc;
            const int z = t / (a.nbm * a.nbn);
            t -= z * (a.nbm * a.nbn);
            const int mt = t / a.nbn, nt = t - mt * a.nbn;
            const int m0 = mt * 256, n0 = nt * 256;
            srdV = __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)a.v + (long long)z * a.T * rowB), 0, (int)(a.T * rowB), 0x00020000);
            srdU = __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)a.u + (long long)z * a.N * rowB), 0, (int)(a.N * rowB), 0x00020000);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int sl = (wave * 3 + q) * 64 + lane;
                const int row = sl / 6, phys = sl - row * 6;
                int logical = phys - ((row >> 3) & 1);
                if (logical < 0) logical += 6;
                gA[q] = (m0 + row < a.T) ? (unsigned)((long long)(m0 + row) * rowB + logical * 16) : OOB;
                gB[q] = (n0 + row < a.N) ? (unsigned)((long long)(n0 + row) * rowB + logical * 16) : OOB;
            }
        } else {                                                      // past my last tile: zero-fill, same instruction count
#pragma unroll
            for (int q = 0; q < 3; ++q) { gA[q] = OOB; gB[q] = OOB; }
        }
    };
    auto dma_instr = [&](int q, int stage) {                          // instruction q of 6 of the stream's current step
        unsigned char *base = dsm + stage * kStage + wave * 3 * 1024;
        const int kb = dK * kIUnit;
        if (q < 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(srdV, (lds_void *)(base + q * 1024), 16, (int)gA[q], kb, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(srdU, (lds_void *)(base + kIOperand + (q - 3) * 1024), 16, (int)gB[q - 3], kb, 0, 0);
    };
    auto advance_dma = [&]() {
        if (++dK == nk) { dK = 0; ++dTile; set_dma_tile(dTile); }
    };

    // ---- fragments (as in the kernel above)
    const int fr = lane & 31, kh = lane >> 5;
    unsigned slotOff[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        int ph = 2 * p + kh + ((fr >> 3) & 1);
        if (ph >= 6) ph -= 6;
        slotOff[p] = (unsigned)(ph * 16);
    }
    const unsigned frA = (unsigned)((wm * 128 + fr) * kIUnit), frB = (unsigned)(kIOperand + (wn * 64 + fr) * kIUnit);
    bf16x8 fa[3][4], fb[3][2];
    bf16x8 faN[4], fbN[2];
    f32x16 acc[4][2];
    auto ldA = [&](const unsigned char *sb, int p, int i) { return *reinterpret_cast<const bf16x8 *>(sb + frA + i * 32 * kIUnit + slotOff[p]); };
    auto ldB = [&](const unsigned char *sb, int p, int j) { return *reinterpret_cast<const bf16x8 *>(sb + frB + j * 32 * kIUnit + slotOff[p]); };
    auto mma_term = [&](int pu, int pv) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[pu][j], fa[pv][i], acc[i][j], 0, 0, 0);
    };

    // ---- prologue: steps 0 and 1 of the stream
    set_dma_tile(0);
#pragma unroll
    for (int q = 0; q < 6; ++q) dma_instr(q, 0);
    advance_dma();
#pragma unroll
    for (int q = 0; q < 6; ++q) dma_instr(q, 1);
    advance_dma();
    __builtin_amdgcn_s_waitcnt(0x0F70 | 6);
    // (a bare s_barrier: the workgroup fence of __syncthreads() makes the compiler wait for EVERY outstanding LDS-DMA,
    //  vmcnt(0), which puts the memory latency of the stream back on the critical path; what has to be ordered is
    //  covered by the counted vmcnt wait above and by program order)
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int j = 0; j < 2; ++j) fbN[j] = ldB(dsm, 2, j);
#pragma unroll
    for (int i = 0; i < 4; ++i) faN[i] = ldA(dsm, 0, i);
    int sc = 0, sd = 2;                                               // stage being multiplied / being filled
    const int rhalf = (lane >> 5) * 4;
    long long cPre = 0, cVm = 0, cBar = 0, cTail = 0, cEpi = 0, cT = 0;
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
            const int next = sc == 2 ? 0 : sc + 1;
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[2][j] = fbN[j];
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[0][i] = faN[i];
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[1][j] = ldB(sb, 1, j);
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[1][i] = ldA(sb, 1, i);
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[0][j] = ldB(sb, 0, j);
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[2][i] = ldA(sb, 2, i);
            mma_term(2, 0); dma_instr(0, sd);
            mma_term(1, 1); dma_instr(1, sd);
            mma_term(0, 2); dma_instr(2, sd);
            mma_term(1, 0); dma_instr(3, sd);
            mma_term(0, 1); dma_instr(4, sd);
            __builtin_amdgcn_sched_barrier(0);
            if (a.clk) { const long long t = clock64(); cPre += t - cT; cT = t; }
            // younger than the operands of step s+1: the 5 DMAs of step s+2 issued so far (and, in the first step of a
            // tile, the 32 stores of the tile before: vmcnt(37) = 0b100101, high bits in [15:14])
            if (kk == 0 && ti > 0) __builtin_amdgcn_s_waitcnt(0x8F70 | 5);
            else __builtin_amdgcn_s_waitcnt(0x0F70 | 5);
            if (a.clk) { const long long t = clock64(); cVm += t - cT; cT = t; }
            __builtin_amdgcn_s_barrier();
            if (a.clk) { const long long t = clock64(); cBar += t - cT; cT = t; }
            __builtin_amdgcn_sched_barrier(0);
            const unsigned char *sn = dsm + next * kStage;
#pragma unroll
            for (int j = 0; j < 2; ++j) fbN[j] = ldB(sn, 2, j);
#pragma unroll
            for (int i = 0; i < 4; ++i) faN[i] = ldA(sn, 0, i);
            mma_term(0, 0); dma_instr(5, sd);
            advance_dma();
            sc = next;
            sd = sd == 2 ? 0 : sd + 1;
            if (a.clk) { const long long t = clock64(); cTail += t - cT; cT = t; }
        }
        // ---- epilogue of tile ti (swapped operands): row = lane & 31, channels (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
        int t = runStart + local + ti * nloc;
        const int z = t / (a.nbm * a.nbn);
        t -= z * (a.nbm * a.nbn);
        const int mt = t / a.nbn, nt = t - mt * a.nbn;
        const int m0 = mt * 256, n0 = nt * 256;
        const __amdgpu_buffer_rsrc_t srdO = __builtin_amdgcn_make_buffer_rsrc((void *)(a.out + (long long)z * a.T * a.N), 0, (int)((long long)a.T * a.N * 4), 0x00020000);
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
                    const f32x4 v = f32x4{ acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3] };
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), srdO, (int)off, 0, 0);
                }
        }
    }
    if (a.clk && lane == 0) {
        long long *c = a.clk + ((long long)blockIdx.x * 8 + wave) * 8;
        c[0] = cPre; c[1] = cVm; c[2] = cBar; c[3] = cTail; c[4] = (long long)myCount * nk;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                             // vmcnt(0)
}


// ---- the same loop with FOUR waves of 128 x 128 (one per SIMD, 256 accumulator registers each) instead of eight of 128 x 64:
// 24 fragment reads per 96 MFMAs instead of 18 per 48 - a third fewer LDS reads per MFMA, the "energy per FLOP" lever the
// round-2 review asked to try (XL_GEMM_PERSIST_WAVES=4 with XL_WINO_V_SPLIT=1; measured A/B in profiles/r3_gemm_ab.*).
// vmcnt bookkeeping: younger than the operands of step s+1 are the 11 DMAs of step s+2 issued so far and, in the first step
// of a tile, the 64 stores of the tile before - 75 > the 63 the counter can express: vmcnt(63) then also waits for the
// oldest 12 of those stores (stricter than needed, correct).
template <int CT>
__global__ __launch_bounds__(256)
void split_gemm_persist4_kernel(SplitArgs2 a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
    typedef __attribute__((address_space(3))) void lds_void;
    constexpr int kStage = 2 * kIOperand;                             // activations then weights: 48 KB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;                          // 2 x 2 waves of 128 x 128

    // tiles of this workgroup: XCD x (= block % 8) owns a contiguous run of the (z, m-tile, n-tile) order, its
    // workgroups take every nloc-th tile of the run, so the workgroups of an XCD work on neighbouring tiles at any time
    const int total = a.nbm * a.nbn * a.Z;
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3, nloc = gridDim.x >> 3;
    const int q8 = total >> 3, r8 = total & 7;
    const int runStart = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int runLen = q8 + (xcd < r8 ? 1 : 0);
    const int myCount = runLen > local ? (runLen - local + nloc - 1) / nloc : 0;
    if (myCount == 0) return;

    constexpr unsigned OOB = 0x80000000u;
    const long long rowB = (long long)a.C * 6;                        // bytes per operand row
    const int nk = CT ? CT / 16 : a.C / 16;

    // ---- operand stream
    // every wave issues 6 of the 48 DMA instructions of a step (3 of each operand), one after each term group.  (Letting
    // waves 0-3 - the older ones on their SIMDs, which win the MFMA arbitration and then sit at the barrier - issue all
    // of them was measured slower: 1.09 vs 1.05 ms.)
    __amdgpu_buffer_rsrc_t srdV, srdU;
    unsigned gA[6], gB[6];
    int dTile = 0, dK = 0;                                            // position of the stream: tile of my list, K-step
    auto set_dma_tile = [&](int i) {
        if (i < myCount) {
            int t = runStart + local + i * nloc;
            const int z = t / (a.nbm * a.nbn);
            t -= z * (a.nbm * a.nbn);
            const int mt = t / a.nbn, nt = t - mt * a.nbn;
            const int m0 = mt * 256, n0 = nt * 256;
            srdV = __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)a.v + (long long)z * a.T * rowB), 0, (int)(a.T * rowB), 0x00020000);
            srdU = __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)a.u + (long long)z * a.N * rowB), 0, (int)(a.N * rowB), 0x00020000);
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                const int sl = (wave * 6 + q) * 64 + lane;
                const int row = sl / 6, phys = sl - row * 6;
                int logical = phys - ((row >> 3) & 1);
                if (logical < 0) logical += 6;
                gA[q] = (m0 + row < a.T) ? (unsigned)((long long)(m0 + row) * rowB + logical * 16) : OOB;
                gB[q] = (n0 + row < a.N) ? (unsigned)((long long)(n0 + row) * rowB + logical * 16) : OOB;
            }
        } else {                                                      // past my last tile: zero-fill, same instruction count
#pragma unroll
            for (int q = 0; q < 6; ++q) { gA[q] = OOB; gB[q] = OOB; }
        }
    };
    auto dma_instr = [&](int q, int stage) {                          // instruction q of 12 of the stream's current step
        unsigned char *base = dsm + stage * kStage + wave * 6 * 1024;
        const int kb = dK * kIUnit;
        if (q < 6) __builtin_amdgcn_raw_ptr_buffer_load_lds(srdV, (lds_void *)(base + q * 1024), 16, (int)gA[q], kb, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(srdU, (lds_void *)(base + kIOperand + (q - 6) * 1024), 16, (int)gB[q - 6], kb, 0, 0);
    };
    auto advance_dma = [&]() {
        if (++dK == nk) { dK = 0; ++dTile; set_dma_tile(dTile); }
    };

    // ---- fragments (as in the kernel above)
    const int fr = lane & 31, kh = lane >> 5;
    unsigned slotOff[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        int ph = 2 * p + kh + ((fr >> 3) & 1);
        if (ph >= 6) ph -= 6;
        slotOff[p] = (unsigned)(ph * 16);
    }
    const unsigned frA = (unsigned)((wm * 128 + fr) * kIUnit), frB = (unsigned)(kIOperand + (wn * 128 + fr) * kIUnit);
    bf16x8 fa[3][4], fb[3][4];
    bf16x8 faN[4], fbN[4];
    f32x16 acc[4][4];
    auto ldA = [&](const unsigned char *sb, int p, int i) { return *reinterpret_cast<const bf16x8 *>(sb + frA + i * 32 * kIUnit + slotOff[p]); };
    auto ldB = [&](const unsigned char *sb, int p, int j) { return *reinterpret_cast<const bf16x8 *>(sb + frB + j * 32 * kIUnit + slotOff[p]); };
    auto mma_term = [&](int pu, int pv) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[pu][j], fa[pv][i], acc[i][j], 0, 0, 0);
    };

    // ---- prologue: steps 0 and 1 of the stream
    set_dma_tile(0);
#pragma unroll
    for (int q = 0; q < 12; ++q) dma_instr(q, 0);
    advance_dma();
#pragma unroll
    for (int q = 0; q < 12; ++q) dma_instr(q, 1);
    advance_dma();
    __builtin_amdgcn_s_waitcnt(0x0F70 | 12);
    // (a bare s_barrier: the workgroup fence of __syncthreads() makes the compiler wait for EVERY outstanding LDS-DMA,
    //  vmcnt(0), which puts the memory latency of the stream back on the critical path; what has to be ordered is
    //  covered by the counted vmcnt wait above and by program order)
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int j = 0; j < 4; ++j) fbN[j] = ldB(dsm, 2, j);
#pragma unroll
    for (int i = 0; i < 4; ++i) faN[i] = ldA(dsm, 0, i);
    int sc = 0, sd = 2;                                               // stage being multiplied / being filled
    const int rhalf = (lane >> 5) * 4;
    long long cPre = 0, cVm = 0, cBar = 0, cTail = 0, cEpi = 0, cT = 0;
    if (a.clk) cT = clock64();
    for (int ti = 0; ti < myCount; ++ti) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int kk = 0; kk < nk; ++kk) {
            const unsigned char *sb = dsm + sc * kStage;
            const int next = sc == 2 ? 0 : sc + 1;
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[2][j] = fbN[j];
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[0][i] = faN[i];
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[1][j] = ldB(sb, 1, j);
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[1][i] = ldA(sb, 1, i);
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[0][j] = ldB(sb, 0, j);
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[2][i] = ldA(sb, 2, i);
            mma_term(2, 0); dma_instr(0, sd); dma_instr(1, sd);
            mma_term(1, 1); dma_instr(2, sd); dma_instr(3, sd);
            mma_term(0, 2); dma_instr(4, sd); dma_instr(5, sd);
            mma_term(1, 0); dma_instr(6, sd); dma_instr(7, sd);
            mma_term(0, 1); dma_instr(8, sd); dma_instr(9, sd); dma_instr(10, sd);
            __builtin_amdgcn_sched_barrier(0);
            if (a.clk) { const long long t = clock64(); cPre += t - cT; cT = t; }
            // younger than the operands of step s+1: the 5 DMAs of step s+2 issued so far (and, in the first step of a
            // tile, the 32 stores of the tile before: vmcnt(37) = 0b100101, high bits in [15:14])
            if (kk == 0 && ti > 0) __builtin_amdgcn_s_waitcnt(0xCF7F);         // vmcnt(63), see the header
            else __builtin_amdgcn_s_waitcnt(0x0F70 | 11);
            if (a.clk) { const long long t = clock64(); cVm += t - cT; cT = t; }
            __builtin_amdgcn_s_barrier();
            if (a.clk) { const long long t = clock64(); cBar += t - cT; cT = t; }
            __builtin_amdgcn_sched_barrier(0);
            const unsigned char *sn = dsm + next * kStage;
#pragma unroll
            for (int j = 0; j < 4; ++j) fbN[j] = ldB(sn, 2, j);
#pragma unroll
            for (int i = 0; i < 4; ++i) faN[i] = ldA(sn, 0, i);
            mma_term(0, 0); dma_instr(11, sd);
            advance_dma();
            sc = next;
            sd = sd == 2 ? 0 : sd + 1;
            if (a.clk) { const long long t = clock64(); cTail += t - cT; cT = t; }
        }
        // ---- epilogue of tile ti (swapped operands): row = lane & 31, channels (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
        int t = runStart + local + ti * nloc;
        const int z = t / (a.nbm * a.nbn);
        t -= z * (a.nbm * a.nbn);
        const int mt = t / a.nbn, nt = t - mt * a.nbn;
        const int m0 = mt * 256, n0 = nt * 256;
        const __amdgpu_buffer_rsrc_t srdO = __builtin_amdgcn_make_buffer_rsrc((void *)(a.out + (long long)z * a.T * a.N), 0, (int)((long long)a.T * a.N * 4), 0x00020000);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // (exactly 64 stores per wave and tile: rows past T fall outside the descriptor, N is a multiple of 256)
            const int m = m0 + wm * 128 + i * 32 + (lane & 31);
            const unsigned rowOff = (unsigned)((long long)m * a.N * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = n0 + wn * 128 + j * 32 + rhalf + 8 * q;
                    const unsigned off = rowOff + (unsigned)n * 4u;
                    const f32x4 v = f32x4{ acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3] };
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), srdO, (int)off, 0, 0);
                }
        }
    }
    if (a.clk && lane == 0) {
        long long *c = a.clk + ((long long)blockIdx.x * 8 + wave) * 8;     // (4 of the 8 slots)
        c[0] = cPre; c[1] = cVm; c[2] = cBar; c[3] = cTail; c[4] = (long long)myCount * nk;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                             // vmcnt(0)
}


// ---------------------------------------------------------------------------------------------- 1x1 convolution on the split pipe
//
// out[m][n] = bias[n] + sum_c f(in[m][c]) * W[n][c],   m = pixel (B*H*W of them, NHWC), f = identity or the producer's
// deferred GroupNorm(+ReLU) (XL_CONV_NORM_IN: per-(image, channel) {scale, shift}), plus the GroupNorm partial sums of the
// OUTPUT (as the fp32 kernel's epilogue, csrc/xl_cnn.hip), so a 1x1 layer stays ONE launch.  The loop is the 256 x 256
// persistent loop above with a different activation path: the weights W are split once on the host (interleaved planes,
// LDS-DMA ring of 3 stages) but the activations arrive as fp32, so every thread
//     loads 8 channels of one row (2 x dwordx4, two K-steps ahead of the multiplies),
//     normalises, splits them into three bf16 terms (v_cvt_pk_bf16_f32; residuals are exact in fp32) and
//     writes 3 x 16 bytes into the activation stage of the NEXT K-step (2 stages, same rotated 96-byte rows),
// behind the fourth term group of the current step.  60-odd VALU instructions per wave and K-step under 48 MFMAs of 8
// passes each.  Measured (512 -> 512, 44 frames of 60 x 90, normalise-on-load + statistics): 0.67-0.69 ms stand-alone,
// 0.62 ms inside the network = 200 TFLOP/s fp32-equivalent; the fp32-MFMA kernel takes 1.0-1.15 ms.  With pieces removed
// (same launch): no conversion arithmetic 0.69 (the VALU work is free), no activation loads 0.58, no LDS writes 0.60,
// neither 0.54, nothing but the weight stream and the MFMAs 0.52; no statistics epilogue -0.02.  Giving the loads two
// K-steps instead of one (second register set) moved 0.69 to 0.68: what the activation path costs is not latency.
// LDS (157 KB): weights 3 x 24 KB | activations 2 x 24 KB | 8 KB | fp64 partials 8 KB | coefficient tables 2 x 8 KB |
// bias 4 KB.  The statistics epilogue stages its per-lane sums in activation stage 1 + the 8 KB behind it: C / 16 is even,
// so when a tile ends stage 1 has just been multiplied and stage 0 holds the first step of the next tile.
struct SplitConvArgs {
    const float *in; const uint16_t *u; const float *bias; float *out;
    const float *coef; float normLo;                 // NORM: [B][C][2] {scale, shift}; lower clamp (0 = ReLU, -inf = none)
    double *stats; int HW, G, nchunks, B;            // statistics of the output (stats == nullptr: none); 16 channels per group
    int M, C, N, ldIn, ldOut, nbm, nbn;
    int tpi;                                         // > 0: tiles start at image boundaries, tpi = ceil(HW / 256) per image (the
                                                     // grouping of the statistics is then independent of the batch); 0: dense
    int Z; long long zIn, zOut;                      // Z > 1: Z independent products in one launch (the frequencies of a Winograd
                                                     // layer: in / out advance by zIn / zOut floats, the weights by N rows; no bias)
    int tile0, ntiles;                               // the launch's range of the (z, m-tile, n-tile) order
    const float *res; int ldRes;                     // RES (with NORM): a residual added behind the normalisation, then ReLU - the
                                                     // producer's whole GroupNorm(+ReLU, +residual, +ReLU) epilogue applied on load
};

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_bf16(float x, float y)        // {bf16(x), bf16(y)} round to nearest even
{
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{ x, y }, bf16x2));
}
__device__ __forceinline__ float hi_f(unsigned w) { return __builtin_bit_cast(float, w & 0xffff0000u); }
__device__ __forceinline__ float lo_f(unsigned w) { return __builtin_bit_cast(float, w << 16); }
// x = t1 + t2 + t3 exactly, for two values at a time (one 32-bit word per term)
// (two values at a time as a float pair: v_pk_add_f32 for the residuals)
__device__ __forceinline__ void split_pair(f32x2 v, unsigned &w1, unsigned &w2, unsigned &w3)
{
    w1 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    const f32x2 r = v - f32x2{ lo_f(w1), hi_f(w1) };
    w2 = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
    const f32x2 r2 = r - f32x2{ lo_f(w2), hi_f(w2) };
    w3 = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2));
}

// NW = waves per workgroup: 8 -> tiles of 256 x 256 (2 x 4 waves of 128 x 64), the throughput form; 4 -> tiles of 128 x 128
// (2 x 2 waves of 64 x 64) for launches whose 256 x 256 tiles would leave most of the 256 CUs idle (a single frame: 44 tiles).
// Every output element accumulates its K-steps and term pairs in the same order in both forms: bitwise the same result.
// ZB = 1 / 2: the instantiations of the batched launches (Z > 1, the GEMMs of a Winograd layer): the same code under names of
// their own, so that a profiler's dispatch table tells them from the 1x1 layers - and, ZB = 2, the 512 -> 512 layers (the
// dominant kernel of the forward pass: compile-time K-step count) from the batched launches of every other shape.  The grid
// of a persistent kernel is the CU count whatever the problem, so the NAME is the only key a dispatch table has.
// BN = tile columns: 256 x 256 (NW 8, BN 256), 256 x 128 (NW 8, BN 128: 4 x 2 waves of 64 x 64 - launches that would run a
// last round of 256 x 256 tiles mostly empty) or 128 x 128 (NW 4, BN 128).
// RES (round 4): the residual form of NORM (XL_CONV_NORM_ADD) - launched as split_conv1x1_res_kernel, so that the names of the
// other instantiations (the keys of profiles/traffic.json and of the PMC tables) stay what they were.
template <bool NORM, bool ACC, int NW, int ZB, int BN, bool RES>               // ACC: out += result
__device__ __forceinline__ void split_conv1x1_body(const SplitConvArgs &a)
{
    static_assert(!RES || NORM, "the residual form is a normalise-on-load form");
    constexpr int NLD = RES ? 4 : 2;                                          // activation loads per thread and K-step
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
    typedef __attribute__((address_space(3))) void lds_void;
    constexpr int NTH = 64 * NW, BM = NTH / 2;                                // threads; tile rows (a thread = 8 channels of a row)
    constexpr int WN = BN / 64, WM = NW / WN, RI = BM / WM / 32;              // waves across columns / rows; 32-row blocks per wave
    static_assert(RI >= 1 && RI * WM * 32 == BM && WN * WM == NW, "tile shape");
    constexpr int NDMA = BN * kIUnit / 1024;                                  // DMA instructions per weight stage: waves 0 .. NDMA/3 - 1
    constexpr int PARTS = NTH * 8 / BN;                                       // statistics: threads per (group, slot) = 8 lane blocks x WM
    constexpr int kW = BN * kIUnit, kAS = BM * kIUnit;                         // one weight stage / one activation stage
    constexpr int kCvA = 3 * kW, kCvStage1 = kCvA + kAS;                       // activation stages; statistics staging: 64 NTH bytes
    constexpr int kCvPart = kCvA + 2 * kAS + 16 * NTH;                         // fp64 partials [NTH / 16][16][2]
    constexpr int kCvCoef = kCvPart + 16 * NTH;                                // coefficient tables of two tiles, 8 KB each (C <= 512)
    constexpr int kCvBias = kCvCoef + 16384;                                   // bias[N <= 1024]
    constexpr int NS = 8 * RI;                                                 // stores per wave and tile
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    // every wave issues three DMA instructions per K-step: the K-step is straight-line code for the scheduler and the counted
    // vmcnt waits are the same in all waves.  With 8 waves on 128 columns (12 instructions per stage) those of waves 4 .. 7 read
    // out of range and write their zeros into a scratch KB of LDS.
    constexpr bool ALLDMA = NDMA >= 3 * NW;
    constexpr int kCvScratch = kCvBias + 4096;
    const bool dmaWave = ALLDMA || wave * 3 < NDMA;
    const int dmaBase = __builtin_amdgcn_readfirstlane(dmaWave ? wave * 3 * 1024 : kCvScratch);
    const int dmaStage = __builtin_amdgcn_readfirstlane(dmaWave ? kW : 0), dmaQ = __builtin_amdgcn_readfirstlane(dmaWave ? 1024 : 0);

    const int total = a.ntiles;                                               // tiles tile0 .. tile0 + ntiles - 1 of the (z, m, n) order
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3, nloc = gridDim.x >> 3;
    const int q8 = total >> 3, r8 = total & 7;
    const int runStart = a.tile0 + (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8);
    const int runLen = q8 + (xcd < r8 ? 1 : 0);
    const int myCount = runLen > local ? (runLen - local + nloc - 1) / nloc : 0;
    if (myCount == 0) return;
    // (z, m-tile, n-tile) order: the workgroups of an XCD walk neighbouring tiles of one product, whose weights stay in its L2
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
    // rows of the tile at m0 (per-image tiles end with their image)
    auto tile_rows = [&](int m0) {
        int rows = a.M - m0;
        if (a.tpi) rows = (m0 / a.HW + 1) * a.HW - m0;
        return rows < BM ? rows : BM;
    };

    constexpr unsigned OOB = 0x80000000u;
    const long long rowU = (long long)a.C * 6;
    const int nk = ZB == 2 ? 32 : a.C / 16;

    // ---- stream two K-steps ahead of the multiplies: weights by LDS-DMA (3 instructions per wave and step), activations
    // into registers (row tid >> 1, channels 8 (tid & 1) .. + 7 of the step; two register sets, by the parity of the step)
    __amdgpu_buffer_rsrc_t srdU = __builtin_amdgcn_make_buffer_rsrc((void *)a.u, 0, (int)(a.N * rowU), 0x00020000);
    __amdgpu_buffer_rsrc_t srdIn = __builtin_amdgcn_make_buffer_rsrc((void *)a.in, 0, 0, 0x00020000);
    __amdgpu_buffer_rsrc_t srdRes = __builtin_amdgcn_make_buffer_rsrc((void *)a.in, 0, 0, 0x00020000);
    const int arow = tid >> 1, ahalf = tid & 1;
    unsigned gB[3], gA = OOB, gR = OOB;
    int dTile = 0, dK = 0;
    auto set_dma_tile = [&](int i) {
        if (i < myCount) {
            int m0, n0;
            tile_at(i, m0, n0);
            const int rows = tile_rows(m0);
            const int z = tile_z(i);
            srdIn = __builtin_amdgcn_make_buffer_rsrc((void *)(a.in + z * a.zIn + (long long)m0 * a.ldIn), 0, rows * a.ldIn * 4, 0x00020000);
            if (a.Z > 1) srdU = __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)a.u + z * (a.N * rowU)), 0, (int)(a.N * rowU), 0x00020000);
            gA = (unsigned)(arow * a.ldIn * 4 + ahalf * 32);              // rows past M fall outside the descriptor
            if constexpr (RES) {
                srdRes = __builtin_amdgcn_make_buffer_rsrc((void *)(a.res + (long long)m0 * a.ldRes), 0, rows * a.ldRes * 4, 0x00020000);
                gR = (unsigned)(arow * a.ldRes * 4 + ahalf * 32);
            }
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int sl = (wave * 3 + q) * 64 + lane;
                const int row = sl / 6, phys = sl - row * 6;
                int logical = phys - ((row >> 3) & 1);
                if (logical < 0) logical += 6;
                gB[q] = dmaWave ? (unsigned)((long long)(n0 + row) * rowU + logical * 16) : OOB;
            }
        } else {
            gA = OOB; gR = OOB;
#pragma unroll
            for (int q = 0; q < 3; ++q) gB[q] = OOB;
        }
    };
    auto dma_instr = [&](int q, int stage) {
        const int dst = ALLDMA ? stage * kW + (wave * 3 + q) * 1024 : dmaBase + stage * dmaStage + q * dmaQ;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srdU, (lds_void *)(dsm + dst), 16, (int)gB[q], dK * kIUnit, 0, 0);
    };
    u32x4 rA[2][2];                                                    // [K-step parity][half of my 8 channels]
    u32x4 rR[2][RES ? 2 : 1];                                          // ... of the residual
    auto load_a = [&](auto parTag) {
        constexpr int P = decltype(parTag)::value;
        rA[P][0] = __builtin_amdgcn_raw_buffer_load_b128(srdIn, (int)gA, dK * 64, 0);
        rA[P][1] = __builtin_amdgcn_raw_buffer_load_b128(srdIn, (int)(gA + 16u), dK * 64, 0);
        if constexpr (RES) {
            rR[P][0] = __builtin_amdgcn_raw_buffer_load_b128(srdRes, (int)gR, dK * 64, 0);
            rR[P][1] = __builtin_amdgcn_raw_buffer_load_b128(srdRes, (int)(gR + 16u), dK * 64, 0);
        }
    };
    auto advance_dma = [&]() {
        if (++dK == nk) { dK = 0; ++dTile; set_dma_tile(dTile); }
    };

    // ---- conversion, one K-step ahead of the multiplies
    unsigned wOff[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        int ph = 2 * p + ahalf + ((arow >> 3) & 1);
        if (ph >= 6) ph -= 6;
        wOff[p] = (unsigned)(kCvA + arow * kIUnit + ph * 16);
    }
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
        constexpr int stage = P;
        unsigned w[3][4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x4 x = __builtin_bit_cast(f32x4, rA[P][h]);
            if constexpr (NORM) {
                const f32x4 c0 = *reinterpret_cast<const f32x4 *>(dsm + cCoef + cK * 128 + h * 32);
                const f32x4 c1 = *reinterpret_cast<const f32x4 *>(dsm + cCoef + cK * 128 + h * 32 + 16);
                // (one rounding per element, as everywhere a GroupNorm is applied.  Scalar fmaf: v_pk_fma_f32 would halve the
                //  fma count but hipcc adds three register-pair moves per packed instruction - 64 VALU per K-step against 56)
                const f32x2 lo = f32x2{ fmaf(x[0], c0[0], c0[1]), fmaf(x[1], c0[2], c0[3]) };
                const f32x2 hi = f32x2{ fmaf(x[2], c1[0], c1[1]), fmaf(x[3], c1[2], c1[3]) };
                x = f32x4{ fmaxf(lo[0], a.normLo), fmaxf(lo[1], a.normLo), fmaxf(hi[0], a.normLo), fmaxf(hi[1], a.normLo) };
                if constexpr (RES) {                                   // gn_apply's epilogue: v += residual; v = max(v, 0)
                    const f32x4 r = __builtin_bit_cast(f32x4, rR[P][h]);
                    x = f32x4{ fmaxf(x[0] + r[0], 0.f), fmaxf(x[1] + r[1], 0.f), fmaxf(x[2] + r[2], 0.f), fmaxf(x[3] + r[3], 0.f) };
                }
            }
            split_pair(f32x2{ x[0], x[1] }, w[0][2 * h], w[1][2 * h], w[2][2 * h]);
            split_pair(f32x2{ x[2], x[3] }, w[0][2 * h + 1], w[1][2 * h + 1], w[2][2 * h + 1]);
        }
#pragma unroll
        for (int p = 0; p < 3; ++p)
            *reinterpret_cast<u32x4 *>(dsm + stage * kAS + wOff[p]) = u32x4{ w[p][0], w[p][1], w[p][2], w[p][3] };
    };
    auto advance_conv = [&]() {
        if (++cK == nk) { cK = 0; ++cTile; set_conv_tile(cTile); }
    };
    // coefficient table of tile i: the {scale, shift} pairs of its (at most two) images, 4 C floats, into table i & 1
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
            if (idx < a.C) *reinterpret_cast<u32x4 *>(dsm + kCvCoef + (i & 1) * 8192 + idx * 16) = t.v[e];
        }
    };

    // ---- fragments
    const int fr = lane & 31, kh = lane >> 5;
    unsigned slotOff[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        int ph = 2 * p + kh + ((fr >> 3) & 1);
        if (ph >= 6) ph -= 6;
        slotOff[p] = (unsigned)(ph * 16);
    }
    const unsigned frA = (unsigned)(kCvA + (wm * (32 * RI) + fr) * kIUnit), frB = (unsigned)((wn * 64 + fr) * kIUnit);
    bf16x8 fa[3][RI], fb[3][2];
    f32x16 acc[RI][2];
    auto ldA = [&](int stage, int p, int i) { return *reinterpret_cast<const bf16x8 *>(dsm + stage * kAS + frA + i * 32 * kIUnit + slotOff[p]); };
    auto ldB = [&](int stage, int p, int j) { return *reinterpret_cast<const bf16x8 *>(dsm + stage * kW + frB + j * 32 * kIUnit + slotOff[p]); };
    auto mma_term = [&](int pu, int pv) {
#pragma unroll
        for (int i = 0; i < RI; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[pu][j], fa[pv][i], acc[i][j], 0, 0, 0);
    };
    const int rhalf = kh * 4;
    auto init_acc = [&](int n0) {                                      // accumulators start at the bias
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
    for (int i = tid; i < a.nbn * BN; i += NTH) reinterpret_cast<float *>(dsm + kCvBias)[i] = (a.bias && i < a.N) ? a.bias[i] : 0.f;
    if constexpr (NORM) {
        const Tab t0 = load_table(0), t1 = load_table(1);
        store_table(0, t0);
        store_table(1, t1);
    }
    set_dma_tile(0);
    set_conv_tile(0);
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    load_a(P0{});
    dma_instr(0, 0); dma_instr(1, 0); dma_instr(2, 0);
    advance_dma();
    __builtin_amdgcn_s_waitcnt(0x0070);                               // everything landed
    __syncthreads();                                                  // tables and bias visible
    convert(P0{});
    advance_conv();
    load_a(P1{});                                                     // (the order of a steady-state step)
    dma_instr(0, 1); dma_instr(1, 1); dma_instr(2, 1);
    advance_dma();
    __builtin_amdgcn_s_waitcnt(0x0070 | (3 + NLD));                   // my writes of step 0; stage 0 of the ring landed before
    __builtin_amdgcn_s_barrier();
    int sc = 0, sd = 2;
    {
        int m0, n0;
        tile_at(0, m0, n0);
        init_acc(n0);
    }
    // one K-step; FIRST: the first step of a tile that follows another one (32 stores of its epilogue are in flight).  The
    // tile loop is written so that such a step is its own code: the compiler's vmcnt bookkeeping for rA (it assumes the
    // fewest outstanding operations over all paths into a block) then does not wait for those stores.
    auto step = [&](auto firstTag, auto parTag) __attribute__((always_inline)) {
        constexpr int sa = decltype(parTag)::value;                   // parity of the K-step
        const int next = sc == 2 ? 0 : sc + 1;
        load_a(parTag);                                               // step kk + 2: two steps until its conversion
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[2][j] = ldB(sc, 2, j);
#pragma unroll
        for (int i = 0; i < RI; ++i) fa[0][i] = ldA(sa, 0, i);
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[1][j] = ldB(sc, 1, j);
#pragma unroll
        for (int i = 0; i < RI; ++i) fa[1][i] = ldA(sa, 1, i);
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[0][j] = ldB(sc, 0, j);
        mma_term(2, 0); dma_instr(0, sd);
        // (the third plane of the activations is first used by term 3: read it here, into the registers the first term's
        //  weight fragments have just left - all 18 fragment reads up front cost 16 more live registers and spilled)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < RI; ++i) fa[2][i] = ldA(sa, 2, i);
        mma_term(1, 1); dma_instr(1, sd);
        __builtin_amdgcn_sched_barrier(0);
        mma_term(0, 2);
        __builtin_amdgcn_sched_barrier(0);
        // terms 4 and 5 with the conversion of step kk + 1 threaded through them (the fragments of the terms before are dead
        // by now: the registers the conversion needs are free): one MFMA (8 passes, 32 cycles of the pipe) covers the issue
        // of 6 VALU instructions of the same wave.  As a block of its own the conversion cost the wave 800-1400 ticks per
        // K-step in which it multiplied nothing (clock64 around the phases), and for most of that time the other wave of
        // the SIMD was converting too.  The LDS writes go out 4 MFMAs before the barrier.
        mma_term(1, 0);
        convert(std::integral_constant<int, sa ^ 1>{});           // (the compiler counts vmcnt for rA)
        mma_term(0, 1);
        {
            constexpr int nM = 4 * RI, groups = nM < 12 ? nM : 12, valu = (RES ? 88 : 72) / groups;    // MFMAs of the two terms; 12 x 6 / 8 x 9 VALU
#pragma unroll
            for (int g = 0; g < groups; ++g) {
                if constexpr (NORM) { if (g == 0 || g == groups / 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); }  // coefficient reads of 4 channels
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, valu, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x200, 3, 0);                      // the three LDS writes
            if constexpr (nM > groups) __builtin_amdgcn_sched_group_barrier(0x008, nM - groups, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        advance_conv();
        // the weights of step kk + 1 have landed: younger are 2 DMAs and 2 loads of step kk + 2 - and, in the first
        // step of a tile, the 32 stores of the tile before (vmcnt(36)); lgkmcnt(0): my activation writes are done
        if constexpr (decltype(firstTag)::value) __builtin_amdgcn_s_waitcnt(0x0070 | ((2 + NLD + NS) & 15) | (((2 + NLD + NS) >> 4) << 14));
        else __builtin_amdgcn_s_waitcnt(0x0070 | (2 + NLD));
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        mma_term(0, 0); dma_instr(2, sd);
        advance_dma();
        sc = next;
        sd = sd == 2 ? 0 : sd + 1;
        __builtin_amdgcn_sched_barrier(0);                            // (the vmcnt arithmetic above assumes this issue order)
    };
    auto epilogue = [&](int ti) __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
        // ---- tile ti (swapped operands): row = lane & 31, channels (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
        int m0, n0;
        tile_at(ti, m0, n0);
        if (a.stats != nullptr) {
            // GroupNorm partial sums of the output; a tile touches at most two images (HW >= 256): slot 0 = rows before
            // `split`, slot 1 = the rest.  Fixed order: per lane fp32 over its 4 rows x 8 channels of a group; fp64 over
            // the 128 lanes holding the group (16 parts of 8 lanes, then the parts); one writer per (image, tile, group).
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
        // (every wave issues exactly 32 stores per tile - the vmcnt arithmetic of the next step counts them: rows past the
        //  end of the tile fall outside the descriptor, N is a multiple of 256)
        const __amdgpu_buffer_rsrc_t srdO = __builtin_amdgcn_make_buffer_rsrc((void *)(a.out + tile_z(ti) * a.zOut + (long long)m0 * a.ldOut), 0,
                                                                              tile_rows(m0) * a.ldOut * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < RI; ++i) {
            const unsigned rowOff = (unsigned)((wm * (32 * RI) + i * 32 + fr) * a.ldOut * 4);
            f32x4 old[2][4];
            if constexpr (ACC) {                                       // (8 loads in flight per 32-row block; rows past the tile read 0)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        old[j][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                            srdO, (int)(rowOff + (unsigned)(n0 + wn * 64 + j * 32 + rhalf + 8 * q) * 4u), 0, 0));
            }
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = n0 + wn * 64 + j * 32 + rhalf + 8 * q;
                    const unsigned off = rowOff + (unsigned)n * 4u;
                    f32x4 v = f32x4{ acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3] };
                    if constexpr (ACC) v += old[j][q];
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), srdO, (int)off, 0, 0);
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

template <bool NORM, bool ACC = false, int NW = 8, int ZB = 0, int BN = (NW == 8 ? 256 : 128)>
__global__ __launch_bounds__(64 * NW)
void split_conv1x1_kernel(SplitConvArgs a)
{
    split_conv1x1_body<NORM, ACC, NW, ZB, BN, false>(a);
}

template <int NW = 8, int BN = (NW == 8 ? 256 : 128)>
__global__ __launch_bounds__(64 * NW)
void split_conv1x1_res_kernel(SplitConvArgs a)
{
    split_conv1x1_body<true, false, NW, 0, BN, true>(a);
}

}  // namespace

// XL_OP_CONV with XL_CONV_SPLIT_BF16: ksize 1, stride 1, nchunks2 = Z batched GEMMs, out fp32 [Z][T][Cout] (ld_out = Cout).
//   + XL_CONV_SPLIT_IL: in = V [Z][T][Cin/16][3][16] bf16, w = U [Z][Cout][Cin/16][3][16] bf16 (256 x 256 persistent kernel);
//   else              : in = plane 0 of V ([Z][T][Cin] bf16, planes Z*T*Cin elements apart), w = plane 0 of U
//                       ([Z][Cout][Cin] bf16, planes Z*Cout*Cin apart) (128 x 128 kernel, the first form).
// XL_OP_CONV with XL_CONV_SPLIT_BF16 | XL_CONV_SPLIT_IL and nchunks2 <= 1: a 1x1 stride-1 convolution, fp32 NHWC in / out
// (ld_in / ld_out), w = [Cout][Cin/16][3][16] bf16, bias, optionally XL_CONV_NORM_IN (aux2 = [B][Cin][2] coefficients) and the
// statistics epilogue (stats / groups / nchunks with 256-row tiles: nchunks >= ceil(HW / 256) + 1, 16 channels per group).
// tileFrom / tileCount: this launch covers the tiles [tileFrom, tileFrom + tileCount) of the form's own (z, m-tile, n-tile)
// numbering (tileCount < 0: all of them).  A 256 x 256 tile t is the 256 x 128 tiles 2t and 2t + 1.
template <int NW, int BN>
static int launch_split_conv1x1(const xl_op &op, SplitConvArgs a, bool norm, int Z, hipStream_t st, int tileFrom = 0, int tileCount = -1)
{
    constexpr int BM = 32 * NW;
    const long long M = a.M;
    const bool perImage = op.reserved_i < 0;         // tiles start at image boundaries (reserved_i = -256 / -128)
    a.tpi = perImage ? (a.HW + BM - 1) / BM : 0;
    a.nbm = perImage ? op.B * a.tpi : (int)((M + BM - 1) / BM);
    a.nbn = (op.Cout + BN - 1) / BN;
    const size_t lds = 3 * BN * kIUnit + 2 * BM * kIUnit + 2 * 16 * (64 * NW) + 16384 + 4096 + 1024;
    const bool accumulate = (op.flags & XL_CONV_ACCUMULATE) != 0;
    const bool resid = norm && a.res != nullptr;
    const bool dominant = Z > 1 && op.Cin == 512 && op.Cout == 512 && !accumulate && !norm;     // the 512 -> 512 Winograd layers
    const void *fn = accumulate ? reinterpret_cast<const void *>(split_conv1x1_kernel<false, true, NW, 0, BN>)
                   : resid ? reinterpret_cast<const void *>(split_conv1x1_res_kernel<NW, BN>)
                   : norm ? reinterpret_cast<const void *>(split_conv1x1_kernel<true, false, NW, 0, BN>)
                   : dominant ? reinterpret_cast<const void *>(split_conv1x1_kernel<false, false, NW, 2, BN>)
                   : Z > 1 ? reinterpret_cast<const void *>(split_conv1x1_kernel<false, false, NW, 1, BN>)
                           : reinterpret_cast<const void *>(split_conv1x1_kernel<false, false, NW, 0, BN>);
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
    if (accumulate) hipLaunchKernelGGL((split_conv1x1_kernel<false, true, NW, 0, BN>), dim3(grid), dim3(64 * NW), lds, st, a);
    else if (resid) hipLaunchKernelGGL((split_conv1x1_res_kernel<NW, BN>), dim3(grid), dim3(64 * NW), lds, st, a);
    else if (norm) hipLaunchKernelGGL((split_conv1x1_kernel<true, false, NW, 0, BN>), dim3(grid), dim3(64 * NW), lds, st, a);
    else if (dominant) hipLaunchKernelGGL((split_conv1x1_kernel<false, false, NW, 2, BN>), dim3(grid), dim3(64 * NW), lds, st, a);
    else if (Z > 1) hipLaunchKernelGGL((split_conv1x1_kernel<false, false, NW, 1, BN>), dim3(grid), dim3(64 * NW), lds, st, a);
    else hipLaunchKernelGGL((split_conv1x1_kernel<false, false, NW, 0, BN>), dim3(grid), dim3(64 * NW), lds, st, a);
    return XL_OK;
}

// reserved_i selects the tile form: +-256 (or 0 / 64, what the fp32 path passes): 256 x 256 tiles; 192: 256 rows x 128 columns;
// 384: 256 x 256 tiles for the full rounds of the 256 CUs and 256 x 128 tiles for a last partial round (same statistics rows);
// +-128: 128 x 128 tiles (the statistics epilogue then writes one entry per 128-row tile: nchunks >= ceil(HW / 128) + 1,
// GN_FINAL / GN_APPLY take 128); negative: tiles start at image boundaries.
static int xl_run_split_conv1x1(const xl_op &op, hipStream_t st)
{
    const long long M = (long long)op.B * op.Ho * op.Wo;
    const int HW = op.Ho * op.Wo;
    const bool norm = (op.flags & XL_CONV_NORM_IN) != 0;
    const bool small = op.reserved_i == 128 || op.reserved_i == -128;
    const int BM = small ? 128 : 256;
    const int Z = op.nchunks2 > 1 ? op.nchunks2 : 1;            // Z > 1: XL_CONV_SPLIT_ACT, the batched GEMMs of a Winograd layer
    if (Z > 1 && (op.bias || op.stats || norm || op.reserved_i < 0 || op.ld_in != op.Cin || op.ld_out != op.Cout)) return XL_ERR_ARG;
    if (op.ksize != 1 || op.stride != 1 || op.Cin % 32 != 0 || op.Cout % 256 != 0 || op.Cout > 1024 || op.ld_in < op.Cin ||
        op.ld_out < op.Cout || (op.ld_in & 3) || (op.ld_out & 3) || ((op.flags & XL_CONV_ACCUMULATE) && (norm || Z > 1 || op.stats)) || !op.in ||
        !op.w || !op.out || (((uintptr_t)op.in | (uintptr_t)op.out | (uintptr_t)op.w) & 15) || M >= 0x7fffffffLL ||
        (long long)op.Cout * op.Cin * 6 >= 0x7fffffffLL || 256LL * op.ld_in * 4 >= 0x7fffffffLL || 256LL * op.ld_out * 4 >= 0x7fffffffLL)
        return XL_ERR_ARG;
    if (norm && (!op.aux2 || op.Cin > 512 || HW < BM)) return XL_ERR_ARG;
    if (op.stats && (op.groups <= 0 || op.Cout != 16 * op.groups || HW < BM || op.nchunks < (HW + BM - 1) / BM + 1)) return XL_ERR_ARG;
    if (op.reserved_i < 0 && HW < BM) return XL_ERR_ARG;
    SplitConvArgs a;
    a.in = (const float *)op.in; a.u = (const uint16_t *)op.w; a.bias = (const float *)op.bias; a.out = (float *)op.out;
    a.coef = (const float *)op.aux2;
    a.normLo = (op.flags & XL_CONV_NORM_RELU) ? 0.f : -__builtin_inff();
    a.stats = (double *)op.stats; a.HW = HW; a.G = op.groups; a.nchunks = op.nchunks; a.B = op.B;
    a.M = (int)M; a.C = op.Cin; a.N = op.Cout; a.ldIn = op.ld_in; a.ldOut = op.ld_out;
    a.Z = Z; a.zIn = M * op.ld_in; a.zOut = M * op.ld_out;
    a.res = nullptr; a.ldRes = 0;
    if (op.flags & XL_CONV_NORM_ADD) {                // the producer's GroupNorm + ReLU + residual (aux, ld_aux) + ReLU applied on load
        if (!norm || !(op.flags & XL_CONV_NORM_RELU) || !op.aux || op.ld_aux < op.Cin || (op.ld_aux & 3) || ((uintptr_t)op.aux & 15) || Z > 1 ||
            256LL * op.ld_aux * 4 >= 0x7fffffffLL) return XL_ERR_ARG;
        a.res = (const float *)op.aux; a.ldRes = op.ld_aux;
    }
    if (op.flags & XL_CONV_M_TILE_MAJOR) {            // result [row][Z][Cout]: a row of product z starts Z*Cout floats after the one before
        if (Z < 2 || op.ld_out != op.Cout || 256LL * Z * op.Cout * 4 >= 0x7fffffffLL) return XL_ERR_ARG;
        a.zOut = op.Cout; a.ldOut = Z * op.Cout;
    }
    a.tpi = 0; a.nbm = 0; a.nbn = 0;
    if (small) return launch_split_conv1x1<4, 128>(op, a, norm, Z, st);
    if (op.reserved_i == 192) return launch_split_conv1x1<8, 128>(op, a, norm, Z, st);
    if (op.reserved_i == 384) {
        // full rounds of 256 x 256 tiles on the 256 CUs, then the rest as 256 x 128 tiles when those fit in ONE round (0.65 of
        // the time of a round of large tiles): two launches over disjoint tile ranges of the same output
        const int nbig = (int)((M + 255) / 256) * (op.Cout / 256) * Z, full = nbig / 256 * 256, rem = nbig - full;
        if (full > 0 && rem > 0 && 2 * rem <= 256) {
            const int rc = launch_split_conv1x1<8, 256>(op, a, norm, Z, st, 0, full);
            return rc != XL_OK ? rc : launch_split_conv1x1<8, 128>(op, a, norm, Z, st, 2 * full, 2 * rem);
        }
        if (full == 0 && 2 * rem <= 256) return launch_split_conv1x1<8, 128>(op, a, norm, Z, st);
    }
    return launch_split_conv1x1<8, 256>(op, a, norm, Z, st);
}

int xl_run_pair(const xl_op &op, hipStream_t st);       // csrc/xl_gemm_pair.hip

int xl_run_split_gemm(const xl_op &op, hipStream_t st)
{
    if (op.flags & XL_CONV_PAIR_F16) return xl_run_pair(op, st);
    if ((op.flags & XL_CONV_SPLIT_IL) && (op.nchunks2 <= 1 || (op.flags & XL_CONV_SPLIT_ACT))) return xl_run_split_conv1x1(op, st);
    const int T = op.B * op.Ho * op.Wo, Z = op.nchunks2;
    if (op.ksize != 1 || op.stride != 1 || Z < 1 || op.Cin % kBK != 0 || op.Cout % 4 != 0 || op.ld_in != op.Cin ||
        op.ld_out != op.Cout || op.bias || op.stats || (op.flags & XL_CONV_ACCUMULATE) || !op.in || !op.w || !op.out)
        return XL_ERR_ARG;
    if (op.flags & XL_CONV_SPLIT_IL) {
        // 32-bit offsets inside one GEMM's operands / result (each z has its own buffer descriptor)
        if (op.Cout % 256 != 0 || (long long)(T + 256) * op.Cout * 4 >= 0xffffffffLL || (long long)T * op.Cin * 6 >= 0x7fffffffLL || (long long)op.Cout * op.Cin * 6 >= 0x7fffffffLL ||
            (long long)T * op.Cout * 4 >= 0x7fffffffLL) return XL_ERR_ARG;
        SplitArgs2 a;
        a.v = (const uint16_t *)op.in; a.u = (const uint16_t *)op.w; a.out = (float *)op.out;
        a.T = T; a.C = op.Cin; a.N = op.Cout; a.Z = Z;
        a.nbm = (T + 255) / 256; a.nbn = (op.Cout + 255) / 256;
        a.clk = nullptr;
        const size_t lds = 3 * 2 * (size_t)kIOperand;                 // 144 KB: one workgroup per CU
        // (the 512-channel layers have an instantiation of their own: a compile-time K-step count, and a kernel name that
        //  tells them apart in a profiler's dispatch table - the grid of a persistent kernel is the CU count for any problem)
        static const bool fourWaves = getenv("XL_GEMM_PERSIST_WAVES") && atoi(getenv("XL_GEMM_PERSIST_WAVES")) == 4;
        auto kernel = fourWaves ? (op.Cin == 512 ? split_gemm_persist4_kernel<512> : split_gemm_persist4_kernel<0>)
                                : (op.Cin == 512 ? split_gemm_persist_kernel<512> : split_gemm_persist_kernel<0>);
        static XlLdsLimit configured[4];
        int cfgDev;
        const int slot = (op.Cin == 512 ? 1 : 0) + (fourWaves ? 2 : 0);
        if (configured[slot].needs(lds, &cfgDev)) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds) != hipSuccess) return XL_ERR_HIP;
            configured[slot].done(lds, cfgDev);
        }
        static const bool clkDbg = getenv("XL_SPLIT_CLK") != nullptr;
        const int nwg = a.nbm * a.nbn * Z;
        int grid = 256;
        if (grid > ((nwg + 7) & ~7)) grid = (nwg + 7) & ~7;
        if (clkDbg && hipMalloc(&a.clk, sizeof(long long) * 64 * grid) != hipSuccess) return XL_ERR_HIP;
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(fourWaves ? 256 : 512), lds, st, a);
        if (clkDbg) {
            std::vector<long long> h((size_t)64 * grid);
            if (hipStreamSynchronize(st) == hipSuccess && hipMemcpy(h.data(), a.clk, sizeof(long long) * 64 * grid, hipMemcpyDeviceToHost) == hipSuccess) {
                double v[2][4] = { { 0 } }; double steps = 0;
                for (int i = 0; i < 8 * grid; ++i) { const int r = (i & 7) < 4 ? 0 : 1; for (int c = 0; c < 4; ++c) v[r][c] += h[8 * (size_t)i + c]; steps += h[8 * (size_t)i + 4]; }
                steps /= 2;                                         // per wave group
                for (int r = 0; r < 2; ++r)
                    fprintf(stderr, "[split clk] waves %d-%d, ticks per K-step: terms 0-4 + reads %.0f, wait for DMAs %.0f, barrier %.0f, term 5 + prefetch %.0f\n",
                            4 * r, 4 * r + 3, v[r][0] / steps, v[r][1] / steps, v[r][2] / steps, v[r][3] / steps);
            }
            (void)hipFree(a.clk);
        }
        return XL_OK;
    }
    SplitArgs a;
    a.v = (const uint16_t *)op.in; a.u = (const uint16_t *)op.w; a.out = (float *)op.out;
    a.T = T; a.C = op.Cin; a.N = op.Cout; a.Z = Z;
    a.vPlane = (long long)Z * T * op.Cin; a.uPlane = (long long)Z * op.Cout * op.Cin;
    const long long vBytes = a.vPlane * 2, uBytes = a.uPlane * 2, outBytes = (long long)Z * T * op.Cout * 4;
    if (vBytes >= 0x7fffffffLL || uBytes >= 0x7fffffffLL || outBytes >= 0x7fffffffLL) return XL_ERR_ARG;
    a.vBytes = (unsigned)vBytes; a.uBytes = (unsigned)uBytes; a.outBytes = (unsigned)outBytes;
    a.nbm = (T + kBM - 1) / kBM; a.nbn = (op.Cout + kBN - 1) / kBN;
    hipLaunchKernelGGL(split_gemm_kernel, dim3(a.nbm * a.nbn * Z), dim3(256), 0, st, a);
    return XL_OK;
}
