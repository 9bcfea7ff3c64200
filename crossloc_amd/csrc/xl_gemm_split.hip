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
// DMA and LDS read removed the same MFMA + barrier stream takes 0.92 ms: under this load the chip clocks near 1.5 GHz
// (power), so the kernel sits at about 87 % of what the matrix pipe delivers at that clock.
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
    const int nk = a.C / 16;

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
            const int m = m0 + wm * 128 + i * 32 + (lane & 31);
            const unsigned rowOff = m < a.T ? (unsigned)((long long)m * a.N * 4) : OOB;
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
    if (a.clk && lane == 0) {
        long long *c = a.clk + ((long long)blockIdx.x * 8 + wave) * 8;
        c[0] = cPre; c[1] = cVm; c[2] = cBar; c[3] = cTail; c[4] = (long long)myCount * nk;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                             // vmcnt(0)
}


}  // namespace

// XL_OP_CONV with XL_CONV_SPLIT_BF16: ksize 1, stride 1, nchunks2 = Z batched GEMMs, out fp32 [Z][T][Cout] (ld_out = Cout).
//   + XL_CONV_SPLIT_IL: in = V [Z][T][Cin/16][3][16] bf16, w = U [Z][Cout][Cin/16][3][16] bf16 (256 x 256 persistent kernel);
//   else              : in = plane 0 of V ([Z][T][Cin] bf16, planes Z*T*Cin elements apart), w = plane 0 of U
//                       ([Z][Cout][Cin] bf16, planes Z*Cout*Cin apart) (128 x 128 kernel, the first form).
int xl_run_split_gemm(const xl_op &op, hipStream_t st)
{
    const int T = op.B * op.Ho * op.Wo, Z = op.nchunks2;
    if (op.ksize != 1 || op.stride != 1 || Z < 1 || op.Cin % kBK != 0 || op.Cout % 4 != 0 || op.ld_in != op.Cin ||
        op.ld_out != op.Cout || op.bias || op.stats || (op.flags & XL_CONV_ACCUMULATE) || !op.in || !op.w || !op.out)
        return XL_ERR_ARG;
    if (op.flags & XL_CONV_SPLIT_IL) {
        // 32-bit offsets inside one GEMM's operands / result (each z has its own buffer descriptor)
        if ((long long)T * op.Cin * 6 >= 0x7fffffffLL || (long long)op.Cout * op.Cin * 6 >= 0x7fffffffLL ||
            (long long)T * op.Cout * 4 >= 0x7fffffffLL) return XL_ERR_ARG;
        SplitArgs2 a;
        a.v = (const uint16_t *)op.in; a.u = (const uint16_t *)op.w; a.out = (float *)op.out;
        a.T = T; a.C = op.Cin; a.N = op.Cout; a.Z = Z;
        a.nbm = (T + 255) / 256; a.nbn = (op.Cout + 255) / 256;
        a.clk = nullptr;
        const size_t lds = 3 * 2 * (size_t)kIOperand;                 // 144 KB: one workgroup per CU
        static XlLdsLimit configured;
        int cfgDev;
        if (configured.needs(lds, &cfgDev)) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(split_gemm_persist_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds) != hipSuccess) return XL_ERR_HIP;
            configured.done(lds, cfgDev);
        }
        static const bool clkDbg = getenv("XL_SPLIT_CLK") != nullptr;
        const int nwg = a.nbm * a.nbn * Z;
        int grid = 256;
        if (grid > ((nwg + 7) & ~7)) grid = (nwg + 7) & ~7;
        if (clkDbg && hipMalloc(&a.clk, sizeof(long long) * 64 * grid) != hipSuccess) return XL_ERR_HIP;
        hipLaunchKernelGGL(split_gemm_persist_kernel, dim3(grid), dim3(512), lds, st, a);
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
