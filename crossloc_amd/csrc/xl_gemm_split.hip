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


// ---------------------------------------------------------------------------------------------- LDS-DMA form
//
// Same product, operands by DMA: tiles go global -> LDS directly (buffer_load ... lds), a ring of NSTAGE stages of one
// K-step of DK channels each, ONE barrier per K-step.  DK = 32: LDS rows of 64 B, 16-byte slots XOR-swizzled by
// (row >> 1) & 3 (applied to the source offset of the lane that fills a slot and again to the fragment reads: conflict-
// free ds_read_b128), 48 KB per stage, 2 chunks of 24 MFMAs per wave and step.  DK = 16: rows of 32 B (halves swapped on
// rows with bit 3 set), 24 KB per stage - but 32-byte global requests: twice the L2 requests of DK = 32 (measured:
// TCC_REQ 3.7e8 vs 1.9e8 per launch, 1.69 vs 1.27 ms), kept for reference only.
constexpr int kWaitVm0 = 0x0F70;                        // s_waitcnt vmcnt(0)

template <int NSTAGE, int DK>
__global__ __launch_bounds__(256)
void split_gemm_dma_kernel(SplitArgs a)
{
    constexpr int kDRow = DK * 2;                       // bytes per LDS row
    constexpr int kSlots = kDRow / 16;                  // 16-byte slots per row: 2 or 4
    constexpr int kRowsPerInstr = 64 / kSlots;          // rows one wave instruction (1 KB) fills
    constexpr int kDPlane = 128 * kDRow;                // one plane of one operand
    constexpr int kDStage = 6 * kDPlane;                // A planes, then B planes
    constexpr int kInstr = 128 / kRowsPerInstr / 4;     // row groups per wave, plane and operand
    constexpr int kPerStep = 6 * kInstr;                // DMA instructions per wave and K-step
    constexpr int kChunks = DK / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
    typedef __attribute__((address_space(3))) void lds_void;
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
    auto swz = [](int row) { return kSlots == 4 ? ((row >> 1) & 3) : ((row >> 3) & 1); };
    // DMA: wave w fills row groups w*kInstr .. of every plane of both operands; lane l fills physical slot l % kSlots of
    // row l / kSlots of the group with the logical slot (l % kSlots) ^ swz(row)
    unsigned gA[kInstr], gB[kInstr];
#pragma unroll
    for (int g = 0; g < kInstr; ++g) {
        const int drow = (wave * kInstr + g) * kRowsPerInstr + lane / kSlots;
        const int dslot = (lane % kSlots) ^ swz(drow);
        gA[g] = (m0 + drow < a.T) ? (unsigned)((((long long)z * a.T + m0 + drow) * a.C + dslot * 8) * 2) : OOB;
        gB[g] = (n0 + drow < a.N) ? (unsigned)((((long long)z * a.N + n0 + drow) * a.C + dslot * 8) * 2) : OOB;
    }
    auto issue_dma = [&](int kk, int stage) {
        const int kb = kk * kDRow;                                  // byte offset of the K-step inside a row (scalar)
#pragma unroll
        for (int g = 0; g < kInstr; ++g) {
            unsigned char *base = dsm + stage * kDStage + (wave * kInstr + g) * 1024;
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(srdV[p], (lds_void *)(base + p * kDPlane), 16, (int)gA[g], kb, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(srdU[p], (lds_void *)(base + (3 + p) * kDPlane), 16, (int)gB[g], kb, 0, 0);
            }
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment addressing: lane -> row lane & 31 of a 32-row block, k-half lane >> 5 of 16-channel chunk c
    const int fr = lane & 31, kh = lane >> 5;
    unsigned fOffA[2][kChunks], fOffB[2][kChunks];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int c = 0; c < kChunks; ++c) {
            const int ra = wm * 64 + i * 32 + fr, rb = wn * 64 + i * 32 + fr;
            fOffA[i][c] = (unsigned)(ra * kDRow + (((2 * c + kh) ^ swz(ra)) * 16));
            fOffB[i][c] = (unsigned)(3 * kDPlane + rb * kDRow + (((2 * c + kh) ^ swz(rb)) * 16));
        }

    const int nk = a.C / DK;
    // steps 0 .. NSTAGE-2 in flight before the loop (steps past nk fetch out-of-range / unused data nobody reads)
#pragma unroll
    for (int sIdx = 0; sIdx < NSTAGE - 1; ++sIdx) issue_dma(sIdx, sIdx);
    int stage = 0;                                                    // kk % NSTAGE
    for (int kk = 0; kk < nk; ++kk) {
        // my DMAs of step kk have landed (the younger steps may still be in flight) ...
        __builtin_amdgcn_s_waitcnt(0x0F70 | ((NSTAGE - 2) * kPerStep));        // vmcnt((NSTAGE-2) * kPerStep) <= 15
        // ... and after the barrier everybody's; every wave has also finished the MFMAs (hence the reads) of step kk-1
        __syncthreads();
        const int prev = stage == 0 ? NSTAGE - 1 : stage - 1;
        issue_dma(kk + NSTAGE - 1, prev);                              // into the stage read one step ago
        const unsigned char *sb = dsm + stage * kDStage;
#pragma unroll
        for (int c = 0; c < kChunks; ++c) {
            bf16x8 fa[3][2], fb[3][2];
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    fa[p][i] = *reinterpret_cast<const bf16x8 *>(sb + p * kDPlane + fOffA[i][c]);
                    fb[p][i] = *reinterpret_cast<const bf16x8 *>(sb + p * kDPlane + fOffB[i][c]);
                }
            constexpr int PU[6] = { 2, 1, 0, 1, 0, 0 }, PV[6] = { 0, 1, 2, 0, 1, 0 };       // u3v1 u2v2 u1v3 u2v1 u1v2 u1v1
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[PU[t]][j], fa[PV[t]][i], acc[i][j], 0, 0, 0);
        }
        stage = stage == NSTAGE - 1 ? 0 : stage + 1;
    }
    __builtin_amdgcn_s_waitcnt(kWaitVm0);                              // drain the trailing DMAs

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

// Software-pipelined form of the DK = 32 ring (one workgroup per CU, one wave per SIMD: nothing else hides a stall, so
// every LDS read and every DMA issue sits in the shadow of MFMAs).  Per K-step, with fragment sets F0 / F1 of the two
// 16-channel chunks:
//   [24 MFMAs on F0]  beside them: the 12 reads of F1 (chunk 1 of this step's stage);
//   s_waitcnt vmcnt + barrier: the next step's stage has landed for everybody, and every wave has issued all its reads
//                              of this step's stage;
//   [24 MFMAs on F1]  beside them: the 12 DMAs of step kk+2 and the 12 reads of F0 of step kk+1.
// NSTAGE = 3: the DMAs of step kk+2 go into the stage read a step ago and are issued BEFORE the barrier (chunk 0 phase),
// so each has two K-steps to land.
template <int NSTAGE>
__global__ __launch_bounds__(256)
void split_gemm_pipe_kernel(SplitArgs a)
{
    constexpr int DK = 32, kDRow = 64, kDPlane = 128 * kDRow, kDStage = 6 * kDPlane;
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
    typedef __attribute__((address_space(3))) void lds_void;
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
    // DMA: wave w fills rows 32w .. 32w+31 (two 16-row instructions) of every plane of both operands
    unsigned gA[2], gB[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int drow = (wave * 2 + g) * 16 + (lane >> 2);
        const int dslot = (lane & 3) ^ ((drow >> 1) & 3);
        gA[g] = (m0 + drow < a.T) ? (unsigned)((((long long)z * a.T + m0 + drow) * a.C + dslot * 8) * 2) : OOB;
        gB[g] = (n0 + drow < a.N) ? (unsigned)((((long long)z * a.N + n0 + drow) * a.C + dslot * 8) * 2) : OOB;
    }
    auto issue_dma = [&](int kk, int stage) {
        const int kb = kk * kDRow;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            unsigned char *base = dsm + stage * kDStage + (wave * 2 + g) * 1024;
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(srdV[p], (lds_void *)(base + p * kDPlane), 16, (int)gA[g], kb, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(srdU[p], (lds_void *)(base + (3 + p) * kDPlane), 16, (int)gB[g], kb, 0, 0);
            }
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int fr = lane & 31, kh = lane >> 5;
    unsigned fOffA[2][2], fOffB[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int ra = wm * 64 + i * 32 + fr, rb = wn * 64 + i * 32 + fr;
            fOffA[i][c] = (unsigned)(ra * kDRow + (((2 * c + kh) ^ ((ra >> 1) & 3)) * 16));
            fOffB[i][c] = (unsigned)(3 * kDPlane + rb * kDRow + (((2 * c + kh) ^ ((rb >> 1) & 3)) * 16));
        }
    bf16x8 fa[2][3][2], fb[2][3][2];                                   // [set][plane][32-row block]
    auto read_frags = [&](int set, int stage, int c) {
        const unsigned char *sb = dsm + stage * kDStage;
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fa[set][p][i] = *reinterpret_cast<const bf16x8 *>(sb + p * kDPlane + fOffA[i][c]);
                fb[set][p][i] = *reinterpret_cast<const bf16x8 *>(sb + p * kDPlane + fOffB[i][c]);
            }
    };
    auto multiply = [&](int set) {
        constexpr int PU[6] = { 2, 1, 0, 1, 0, 0 }, PV[6] = { 0, 1, 2, 0, 1, 0 };           // u3v1 u2v2 u1v3 u2v1 u1v2 u1v1
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[set][PU[t]][j], fa[set][PV[t]][i], acc[i][j], 0, 0, 0);
    };

    const int nk = a.C / DK;
    issue_dma(0, 0);
    issue_dma(1, 1);
    __builtin_amdgcn_s_waitcnt(kWaitVm0);
    __syncthreads();
    read_frags(0, 0, 0);
    int stage = 0;
    for (int kk = 0; kk < nk; ++kk) {
        const int next = stage == NSTAGE - 1 ? 0 : stage + 1;
        const int tgt = NSTAGE == 2 ? stage : (stage == 0 ? NSTAGE - 1 : stage - 1);      // stage of step kk+2
        // ---- chunk 0: MFMAs on F0, the reads of F1 (and, with three stages, the DMAs of step kk+2) beside them
        read_frags(1, stage, 1);
        if (NSTAGE == 3) issue_dma(kk + 2, tgt);
        multiply(0);
        if (NSTAGE == 3) {
#pragma unroll
            for (int r = 0; r < 12; ++r) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 12; ++r) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // step kk+1 has landed (three stages: step kk+2, issued above, may still be in flight)
        __builtin_amdgcn_s_waitcnt(NSTAGE == 3 ? (0x0F70 | 12) : 0x0F70);
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        // ---- chunk 1: MFMAs on F1; beside them the reads of F0 of the next step (and, with two stages, the DMAs of kk+2)
        if (NSTAGE == 2) issue_dma(kk + 2, tgt);
        read_frags(0, next, 0);
        multiply(1);
        if (NSTAGE == 2) {
#pragma unroll
            for (int r = 0; r < 12; ++r) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 12; ++r) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        stage = next;
    }
    __builtin_amdgcn_s_waitcnt(kWaitVm0);

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

template <int NSTAGE>
int launch_split_pipe(const SplitArgs &a, hipStream_t st)
{
    const size_t lds = (size_t)NSTAGE * 6 * 128 * 64;
    static bool configured = false;
    if (!configured) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(split_gemm_pipe_kernel<NSTAGE>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return XL_ERR_HIP;
        configured = true;
    }
    hipLaunchKernelGGL((split_gemm_pipe_kernel<NSTAGE>), dim3(a.nbm * a.nbn * a.Z), dim3(256), lds, st, a);
    return XL_OK;
}

// ---------------------------------------------------------------------------------------------- 256 x 256 form
//
// What bounds the forms above is operand delivery, not the matrix pipe: a 128 x 128 tile moves 48 KB of split operands
// out of L2 per 1.05 MFLOP (fp32-equivalent), in 64-byte requests (one per row, plane and 32-channel step): 10 GB and
// 1.9e8 L2 requests per 512-channel layer launch (TCC_REQ), 8 TB/s at the measured 1.27 ms while the MFMA pipe is 58 %
// busy.  This form halves both: 256 x 256 tiles (8 waves of 128 x 64, one workgroup per CU) and an operand layout with
// the three planes of a 16-channel chunk next to each other,
//     V[z][t][c / 16][plane][c % 16]   (96 bytes per row and K-step, 1.5 L2 requests on average instead of 3)
// so that a K-step of 16 channels - exactly one k-depth of v_mfma_f32_32x32x16_bf16 - is one contiguous piece per row.
// Stage = (256 + 256) rows x 96 B = 48 KB, ring of 3 stages (144 KB), ONE barrier per K-step, the DMAs of step kk+2
// issued at the top of step kk.  LDS rows are 6 slots of 16 bytes, rotated by one slot on rows with bit 3 set (rows r and
// r + 8 would otherwise hit the same banks: 96 r mod 256); the rotation is applied to the source offset of the lane that
// fills a slot and to the fragment reads.
// The activation operand (first touched here: HBM latency) and the weight operand (L2 hits) have separate rings, 4 and
// 2 stages deep, and separate issuing waves (0-3 / 4-7): vmcnt counts per wave and in order, so only a wave that issues
// nothing but activation DMAs can wait for step kk while its steps kk+1 and kk+2 are still in flight.
constexpr int kIUnit = 96;                              // bytes per row and K-step: 3 planes x 16 bf16
constexpr int kIOperand = 256 * kIUnit;                 // one operand of one stage: 24 KB
constexpr int kIRingA = 4, kIRingB = 2;
constexpr int kILds = (kIRingA + kIRingB) * kIOperand;  // 144 KB

struct SplitArgs2 {
    const uint16_t *v, *u;      // interleaved layout, [Z][T][C/16][3][16] and [Z][N][C/16][3][16]
    float *out;                 // [Z][T][N] fp32
    int T, C, N, Z, nbm, nbn;
    long long *clk;             // diagnostics (XL_SPLIT_CLK=1): per-wave {loop ticks, ticks parked at waitcnt + barrier}
    int ablate;                 // diagnostics (XL_SPLIT_ABLATE bit mask, results are garbage): 1 no activation DMAs, 2 no
                                // weight DMAs, 4 no fragment reads, 8 no MFMAs
};

__global__ __launch_bounds__(512)
void split_gemm_256_kernel(SplitArgs2 a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
    typedef __attribute__((address_space(3))) void lds_void;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;                          // 2 x 4 waves of 128 x 64
    long long tEntry = 0;
    if (a.clk) tEntry = clock64();

    int tile = xcd_remap(blockIdx.x, a.nbm * a.nbn * a.Z);
    const int z = tile / (a.nbm * a.nbn);
    tile -= z * (a.nbm * a.nbn);
    const int mt = tile / a.nbn, nt = tile - mt * a.nbn;
    const int m0 = mt * 256, n0 = nt * 256;

    constexpr unsigned OOB = 0x80000000u;
    const long long rowB = (long long)a.C * 6;                        // bytes per operand row
    const __amdgpu_buffer_rsrc_t srdV = __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)a.v + (long long)z * a.T * rowB), 0, (int)(a.T * rowB), 0x00020000);
    const __amdgpu_buffer_rsrc_t srdU = __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)a.u + (long long)z * a.N * rowB), 0, (int)(a.N * rowB), 0x00020000);
    // DMA: an operand stage is 1536 slots of 16 bytes = 24 instructions of 1 KB; waves 0-3 issue the activation operand
    // (6 instructions each), waves 4-7 the weights
    const bool isA = wave < 4;
    unsigned gOff[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const int sl = ((wave & 3) * 6 + q) * 64 + lane;
        const int row = sl / 6, phys = sl - row * 6;
        int logical = phys - ((row >> 3) & 1);
        if (logical < 0) logical += 6;
        const int lim = isA ? a.T : a.N, r0 = isA ? m0 : n0;
        gOff[q] = (r0 + row < lim) ? (unsigned)((long long)(r0 + row) * rowB + logical * 16) : OOB;
    }
    unsigned char *ringA = dsm, *ringB = dsm + kIRingA * kIOperand;
    auto issue_dma = [&](int kk, int stage) {                       // wave-uniform role: no divergence
        const int kb = kk * kIUnit;
        unsigned char *base = (isA ? ringA : ringB) + stage * kIOperand + (wave & 3) * 6 * 1024;
        if (isA) {
#pragma unroll
            for (int q = 0; q < 6; ++q)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(srdV, (lds_void *)(base + q * 1024), 16, (int)gOff[q], kb, 0, 0);
        } else {
#pragma unroll
            for (int q = 0; q < 6; ++q)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(srdU, (lds_void *)(base + q * 1024), 16, (int)gOff[q], kb, 0, 0);
        }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment addressing: lane -> row lane & 31 of a 32-row block, k-half lane >> 5; slot 2 * plane + k-half, rotated.
    // Bit 3 of a row is bit 3 of lane & 31 for every block of both operands, so the slot offsets are per lane and plane.
    const int fr = lane & 31, kh = lane >> 5;
    unsigned slotOff[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        int ph = 2 * p + kh + ((fr >> 3) & 1);
        if (ph >= 6) ph -= 6;
        slotOff[p] = (unsigned)(ph * 16);
    }
    const unsigned frA = (unsigned)((wm * 128 + fr) * kIUnit), frB = (unsigned)((wn * 64 + fr) * kIUnit);
    bf16x8 fa[3][4], fb[3][2];
    bf16x8 faN[4], fbN[2];                                             // first term's operands of the NEXT step
    auto ldA = [&](const unsigned char *sb, int p, int i) { return *reinterpret_cast<const bf16x8 *>(sb + frA + i * 32 * kIUnit + slotOff[p]); };
    auto ldB = [&](const unsigned char *sb, int p, int j) { return *reinterpret_cast<const bf16x8 *>(sb + frB + j * 32 * kIUnit + slotOff[p]); };
    auto mma_term = [&](int pu, int pv) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[pu][j], fa[pv][i], acc[i][j], 0, 0, 0);
    };

    // Per K-step (terms smallest first: u3v1 u2v2 u1v3 u2v1 u1v2 u1v1):
    //   top     DMAs of step kk+2 (into the stage read at step kk-1); the reads of everything but the first term's
    //           operands, which were prefetched at the end of the previous step; 5 terms (40 MFMAs);
    //   then    s_waitcnt + barrier: step kk+1 has landed for everybody and every wave has issued its reads of step kk;
    //   tail    prefetch of the first term's operands of step kk+1 beside the 6th term (8 MFMAs).
    // The barrier therefore never has the matrix pipe waiting for LDS data behind it.
    const int nk = a.C / 16;
    // in flight ahead of the step being multiplied: 3 activation steps, 1 weight step
    if (isA) { issue_dma(0, 0); issue_dma(1, 1); issue_dma(2, 2); }
    else issue_dma(0, 0);
    if (isA) __builtin_amdgcn_s_waitcnt(0x0F70 | 12);
    else __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) fbN[j] = ldB(ringB, 2, j);
#pragma unroll
    for (int i = 0; i < 4; ++i) faN[i] = ldA(ringA, 0, i);
    int sa = 0, sbI = 0;                                               // kk % 4, kk % 2
    long long tLoop = 0, tPark = 0, tRead = 0;
    if (a.clk) tLoop = clock64();
    for (int kk = 0; kk < nk; ++kk) {
        const unsigned char *pa = ringA + sa * kIOperand, *pb = ringB + sbI * kIOperand;
        // activations of step kk+3 into the stage read at step kk-1; weights of step kk+1 into the other weight stage
        // (both free: every wave is past the barrier of step kk-1, after which only the prefetched operands of step kk
        // were read from other stages)
        if (isA) issue_dma(kk + 3, (sa + 3) & 3);
        else issue_dma(kk + 1, sbI ^ 1);
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[2][j] = fbN[j];
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[0][i] = faN[i];
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[1][j] = ldB(pb, 1, j);
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[1][i] = ldA(pa, 1, i);
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[0][j] = ldB(pb, 0, j);
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[2][i] = ldA(pa, 2, i);
        mma_term(2, 0);
        mma_term(1, 1);
        mma_term(0, 2);
        mma_term(1, 0);
        mma_term(0, 1);
        __builtin_amdgcn_sched_barrier(0);
        long long t0 = 0;
        if (a.clk) t0 = clock64();
        // step kk+1 has landed: activation waves still have steps kk+2 and kk+3 in flight, weight waves nothing
        if (isA) __builtin_amdgcn_s_waitcnt(0x0F70 | 12);
        else __builtin_amdgcn_s_waitcnt(0x0F70);
        if (a.clk) { const long long t1 = clock64(); tRead += t1 - t0; t0 = t1; }
        __syncthreads();
        if (a.clk) tPark += clock64() - t0;
        __builtin_amdgcn_sched_barrier(0);
        sa = (sa + 1) & 3; sbI ^= 1;
        const unsigned char *na = ringA + sa * kIOperand, *nb = ringB + sbI * kIOperand;
#pragma unroll
        for (int j = 0; j < 2; ++j) fbN[j] = ldB(nb, 2, j);
#pragma unroll
        for (int i = 0; i < 4; ++i) faN[i] = ldA(na, 0, i);
        mma_term(0, 0);
    }
    __builtin_amdgcn_s_waitcnt(kWaitVm0);
    if (a.clk && lane == 0) {
        long long *c = a.clk + ((long long)blockIdx.x * 8 + wave) * 4;
        c[0] = clock64() - tLoop; c[1] = tRead; c[2] = tPark; c[3] = tLoop - tEntry;
    }

    const __amdgpu_buffer_rsrc_t srdO = __builtin_amdgcn_make_buffer_rsrc((void *)(a.out + (long long)z * a.T * a.N), 0, (int)((long long)a.T * a.N * 4), 0x00020000);
    const int rhalf = (lane >> 5) * 4;
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
    if (a.clk && lane == 0) {
        __builtin_amdgcn_s_waitcnt(kWaitVm0);
        a.clk[((long long)blockIdx.x * 8 + wave) * 4 + 3] |= (clock64() - tEntry) << 24;   // total lifetime above the prologue ticks
    }
}

// ---------------------------------------------------------------------------------------------- persistent 256 x 256 form
//
// XL_SPLIT_CLK on the kernel above: a tile lives 178k shader ticks of which 98k are MFMA time - prologue 13k, epilogue
// (until the stores are acknowledged) 19k, dispatch of the next workgroup 19k, and a K-step takes 3970 ticks instead of
// 3072 because all eight waves issue their 6 DMA instructions (~100 ticks each, in order before their MFMAs) at the same
// moment.  This form removes both: ONE workgroup per CU walks its tiles, the operand stream (a ring of 3 stages of
// 48 KB, two K-steps ahead of the multiplies) runs across tile boundaries so there is a prologue per workgroup only, the
// epilogue is 32 store instructions per wave whose completion nobody waits for, and the DMA instructions of a step are
// issued one after each of the six term groups.
// vmcnt bookkeeping (in order, per wave): at the barrier of step s the operands of step s+1 must have landed; younger
// than those are the DMAs of step s+2 issued so far - and, in the first step of a tile, the 32 stores of the tile before.
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
    const __amdgpu_buffer_rsrc_t srdO = __builtin_amdgcn_make_buffer_rsrc((void *)a.out, 0, 0x7ffffff0, 0x00020000);
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
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + wm * 128 + i * 32 + (lane & 31);
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
    if (a.clk && lane == 0) {
        long long *c = a.clk + ((long long)blockIdx.x * 8 + wave) * 8;
        c[0] = cPre; c[1] = cVm; c[2] = cBar; c[3] = cTail; c[4] = (long long)myCount * nk;
    }
    __builtin_amdgcn_s_waitcnt(kWaitVm0);
}

template <int NSTAGE, int DK>
int launch_split_dma(const SplitArgs &a, hipStream_t st)
{
    const size_t lds = (size_t)NSTAGE * 6 * 128 * DK * 2;
    static bool configured = false;
    if (!configured) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(split_gemm_dma_kernel<NSTAGE, DK>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return XL_ERR_HIP;
        configured = true;
    }
    hipLaunchKernelGGL((split_gemm_dma_kernel<NSTAGE, DK>), dim3(a.nbm * a.nbn * a.Z), dim3(256), lds, st, a);
    return XL_OK;
}

}  // namespace

// XL_OP_CONV with XL_CONV_SPLIT_BF16: ksize 1, stride 1, nchunks2 = Z batched GEMMs; in = plane 0 of V ([Z][T][Cin]
// bf16, planes Z*T*Cin elements apart), w = plane 0 of U ([Z][Cout][Cin] bf16, planes Z*Cout*Cin apart), out fp32
// [Z][T][Cout] with ld_out = Cout.
int xl_run_split_gemm(const xl_op &op, hipStream_t st)
{
    if (op.flags & XL_CONV_SPLIT_IL) {
        // interleaved-plane operands, 256 x 256 tiles
        const int T = op.B * op.Ho * op.Wo, Z = op.nchunks2;
        if (op.ksize != 1 || op.stride != 1 || Z < 1 || op.Cin % 16 != 0 || op.Cout % 4 != 0 || op.ld_in != op.Cin ||
            op.ld_out != op.Cout || op.bias || op.stats || (op.flags & XL_CONV_ACCUMULATE) || !op.in || !op.w || !op.out)
            return XL_ERR_ARG;
        if ((long long)T * op.Cin * 6 >= 0x7fffffffLL || (long long)op.Cout * op.Cin * 6 >= 0x7fffffffLL ||
            (long long)T * op.Cout * 4 >= 0x7fffffffLL) return XL_ERR_ARG;
        SplitArgs2 a;
        a.v = (const uint16_t *)op.in; a.u = (const uint16_t *)op.w; a.out = (float *)op.out;
        a.T = T; a.C = op.Cin; a.N = op.Cout; a.Z = Z;
        a.nbm = (T + 255) / 256; a.nbn = (op.Cout + 255) / 256;
        a.ablate = getenv("XL_SPLIT_ABLATE") ? atoi(getenv("XL_SPLIT_ABLATE")) : 0;
        const size_t lds = (size_t)kILds;
        static bool configured = false;
        if (!configured) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(split_gemm_256_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds) != hipSuccess) return XL_ERR_HIP;
            configured = true;
        }
        static const bool clkDbg = getenv("XL_SPLIT_CLK") != nullptr;
        static const bool persist = getenv("XL_SPLIT_NO_PERSIST") == nullptr;
        a.clk = nullptr;
        const int nwg = a.nbm * a.nbn * Z;
        if (persist && (long long)Z * T * op.Cout * 4 < 0x7ffffff0LL) {
            static bool cfgP = false;
            if (!cfgP) {
                if (hipFuncSetAttribute(reinterpret_cast<const void *>(split_gemm_persist_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        3 * 2 * kIOperand) != hipSuccess) return XL_ERR_HIP;
                cfgP = true;
            }
            int grid = 256;                                       // one workgroup per CU (144 KB of LDS each)
            if (grid > ((nwg + 7) & ~7)) grid = (nwg + 7) & ~7;
            if (clkDbg && hipMalloc(&a.clk, sizeof(long long) * 64 * grid) != hipSuccess) return XL_ERR_HIP;
            hipLaunchKernelGGL(split_gemm_persist_kernel, dim3(grid), dim3(512), 3 * 2 * kIOperand, st, a);
            if (clkDbg) {
                std::vector<long long> h((size_t)64 * grid);
                if (hipStreamSynchronize(st) == hipSuccess && hipMemcpy(h.data(), a.clk, sizeof(long long) * 64 * grid, hipMemcpyDeviceToHost) == hipSuccess) {
                    double v[2][4] = { { 0 } }; double steps = 0;
                    for (int i = 0; i < 8 * grid; ++i) { const int r = (i & 7) < 4 ? 0 : 1; for (int c = 0; c < 4; ++c) v[r][c] += h[8 * (size_t)i + c]; steps += h[8 * (size_t)i + 4]; }
                    steps /= 2;                                     // per wave group
                    for (int r = 0; r < 2; ++r)
                        fprintf(stderr, "[split clk] waves %d-%d, ticks per K-step: terms 0-4 + reads %.0f, wait for DMAs %.0f, barrier %.0f, term 5 + prefetch %.0f\n",
                                4 * r, 4 * r + 3, v[r][0] / steps, v[r][1] / steps, v[r][2] / steps, v[r][3] / steps);
                }
                (void)hipFree(a.clk);
            }
            return XL_OK;
        }
        if (clkDbg && hipMalloc(&a.clk, sizeof(long long) * 32 * nwg) != hipSuccess) return XL_ERR_HIP;
        hipLaunchKernelGGL(split_gemm_256_kernel, dim3(nwg), dim3(512), lds, st, a);
        if (clkDbg) {
            long long *h = (long long *)malloc(sizeof(long long) * 32 * nwg);
            if (hipStreamSynchronize(st) == hipSuccess && hipMemcpy(h, a.clk, sizeof(long long) * 32 * nwg, hipMemcpyDeviceToHost) == hipSuccess) {
                double tot[2] = { 0, 0 }, rd[2] = { 0, 0 }, pk[2] = { 0, 0 }, pro[2] = { 0, 0 }, life[2] = { 0, 0 }; long long n[2] = { 0, 0 };
                for (long long i = 0; i < 8LL * nwg; ++i) {
                    const int r = ((i & 7) < 4) ? 1 : 0;
                    tot[r] += h[4 * i]; rd[r] += h[4 * i + 1]; pk[r] += h[4 * i + 2];
                    pro[r] += (double)(h[4 * i + 3] & 0xffffff); life[r] += (double)(h[4 * i + 3] >> 24); ++n[r];
                }
                for (int r = 0; r < 2; ++r)
                    fprintf(stderr, "[split clk] %s waves: lifetime %.0f ticks = prologue %.0f + loop %.0f + epilogue %.0f; in the loop: waiting for own DMAs %.0f (%.1f%%), at the barrier %.0f (%.1f%%), %d K-steps\n",
                            r ? "activation" : "weight", life[r] / n[r], pro[r] / n[r], tot[r] / n[r], (life[r] - pro[r] - tot[r]) / n[r],
                            rd[r] / n[r], 100.0 * rd[r] / tot[r], pk[r] / n[r], 100.0 * pk[r] / tot[r], a.C / 16);
            }
            free(h);
            (void)hipFree(a.clk);
        }
        return XL_OK;
    }
    const int T = op.B * op.Ho * op.Wo, Z = op.nchunks2;
    if (op.ksize != 1 || op.stride != 1 || Z < 1 || op.Cin % kBK != 0 || op.Cout % 4 != 0 || op.ld_in != op.Cin ||
        op.ld_out != op.Cout || op.bias || op.stats || (op.flags & XL_CONV_ACCUMULATE) || !op.in || !op.w || !op.out)
        return XL_ERR_ARG;
    SplitArgs a;
    a.v = (const uint16_t *)op.in; a.u = (const uint16_t *)op.w; a.out = (float *)op.out;
    a.T = T; a.C = op.Cin; a.N = op.Cout; a.Z = Z;
    a.vPlane = (long long)Z * T * op.Cin; a.uPlane = (long long)Z * op.Cout * op.Cin;
    const long long vBytes = a.vPlane * 2, uBytes = a.uPlane * 2, outBytes = (long long)Z * T * op.Cout * 4;
    if (vBytes >= 0x7fffffffLL || uBytes >= 0x7fffffffLL || outBytes >= 0x7fffffffLL) return XL_ERR_ARG;
    a.vBytes = (unsigned)vBytes; a.uBytes = (unsigned)uBytes; a.outBytes = (unsigned)outBytes;
    a.nbm = (T + kBM - 1) / kBM; a.nbn = (op.Cout + kBN - 1) / kBN;
    // XL_SPLIT_FORM: 0 = register-staged single-buffer loop; 216 / 316 = LDS-DMA ring of 2 / 3 stages of 16 channels,
    // 232 / 332 = of 32 channels
    static const int form = getenv("XL_SPLIT_FORM") ? atoi(getenv("XL_SPLIT_FORM")) : 0;
    if (form == 0 || op.Cin % 32 != 0) {
        hipLaunchKernelGGL(split_gemm_kernel, dim3(a.nbm * a.nbn * Z), dim3(256), 0, st, a);
        return XL_OK;
    }
    if (form == 216) return launch_split_dma<2, 16>(a, st);
    if (form == 316) return launch_split_dma<3, 16>(a, st);
    if (form == 232) return launch_split_dma<2, 32>(a, st);
    if (form == 332) return launch_split_dma<3, 32>(a, st);
    if (form == 2) return launch_split_pipe<2>(a, st);
    if (form == 3) return launch_split_pipe<3>(a, st);
    return XL_ERR_ARG;
    return XL_OK;
}
