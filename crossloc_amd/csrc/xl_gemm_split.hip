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

}  // namespace

// XL_OP_CONV with XL_CONV_SPLIT_BF16: ksize 1, stride 1, nchunks2 = Z batched GEMMs; in = plane 0 of V ([Z][T][Cin]
// bf16, planes Z*T*Cin elements apart), w = plane 0 of U ([Z][Cout][Cin] bf16, planes Z*Cout*Cin apart), out fp32
// [Z][T][Cout] with ld_out = Cout.
int xl_run_split_gemm(const xl_op &op, hipStream_t st)
{
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
    hipLaunchKernelGGL(split_gemm_kernel, dim3(a.nbm * a.nbn * Z), dim3(256), 0, st, a);
    return XL_OK;
}
