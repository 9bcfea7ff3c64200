// crossloc_hip: weight re-pack kernels - what a plan runs once per weight version (after load_state_dict, and after
// every optimizer step of a training loop): the Winograd transform U = G g G^T (float64 inside, rounded once) and the exact
// three-term bf16 split of the operands of the split-pipe kernels.  One launch per layer, no library calls.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/crossloc_cnn.h"
#include "../../include/crossloc_dsac.h"   // status codes
#include "xl_common.h"

namespace {

__device__ __forceinline__ unsigned bf16_rn(float x)
{
    unsigned u = __builtin_bit_cast(unsigned, x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ void split3(float a, unsigned &h1, unsigned &h2, unsigned &h3)
{
    h1 = bf16_rn(a);
    const float r1 = a - __builtin_bit_cast(float, h1 << 16);
    h2 = bf16_rn(r1);
    h3 = bf16_rn(r1 - __builtin_bit_cast(float, h2 << 16));
}

// G of F(m x m, 3x3): [(m+2)][3], the matrices of crossloc_amd/networks.py::_Plan._WINO_G
template <int M> struct WinoG;
template <> struct WinoG<2> { static constexpr double g[4][3] = { { 1.0, 0.0, 0.0 }, { 0.5, 0.5, 0.5 }, { 0.5, -0.5, 0.5 }, { 0.0, 0.0, 1.0 } }; };
template <> struct WinoG<4> { static constexpr double g[6][3] = { { 1.0 / 4, 0.0, 0.0 }, { -1.0 / 6, -1.0 / 6, -1.0 / 6 }, { -1.0 / 6, 1.0 / 6, -1.0 / 6 },
                                                                   { 1.0 / 24, 1.0 / 12, 1.0 / 6 }, { 1.0 / 24, -1.0 / 12, 1.0 / 6 }, { 0.0, 0.0, 1.0 } }; };
template <> struct WinoG<6> { static constexpr double g[8][3] = { { 1.0, 0.0, 0.0 }, { -2.0 / 9, -2.0 / 9, -2.0 / 9 }, { -2.0 / 9, 2.0 / 9, -2.0 / 9 },
                                                                   { 1.0 / 90, 1.0 / 45, 2.0 / 45 }, { 1.0 / 90, -1.0 / 45, 2.0 / 45 },
                                                                   { 1.0 / 45, 1.0 / 90, 1.0 / 180 }, { 1.0 / 45, -1.0 / 90, 1.0 / 180 }, { 0.0, 0.0, 1.0 } }; };

// one thread per (row, k) pair of the operand: 9 weights in, (M+2)^2 transformed values out (coalesced over k)
template <int M>
__global__ __launch_bounds__(256)
void wino_weight_kernel(const float *__restrict__ w, void *__restrict__ dst, int Cout, int Cin, int dgrad, int form)
{
    constexpr int N = M + 2;
    const int rows = dgrad ? Cin : Cout, K = dgrad ? Cout : Cin;
    const long long total = (long long)rows * K;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int r = (int)(i / K), k = (int)(i - (long long)r * K);
        const int o = dgrad ? k : r, c = dgrad ? r : k;
        const float *src = w + ((long long)o * Cin + c) * 9;
        double g[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) g[a][b] = (double)(dgrad ? src[(2 - a) * 3 + (2 - b)] : src[a * 3 + b]);
        double t[N][3];                                  // G g
#pragma unroll
        for (int x = 0; x < N; ++x)
#pragma unroll
            for (int b = 0; b < 3; ++b)
                t[x][b] = WinoG<M>::g[x][0] * g[0][b] + WinoG<M>::g[x][1] * g[1][b] + WinoG<M>::g[x][2] * g[2][b];
        const long long plane = (long long)N * N * rows * K;
#pragma unroll
        for (int x = 0; x < N; ++x)
#pragma unroll
            for (int y = 0; y < N; ++y) {
                const float u = (float)(t[x][0] * WinoG<M>::g[y][0] + t[x][1] * WinoG<M>::g[y][1] + t[x][2] * WinoG<M>::g[y][2]);
                const long long e = ((long long)(x * N + y) * rows + r) * K + k;
                if (form == 0) reinterpret_cast<float *>(dst)[e] = u;
                else {
                    unsigned h1, h2, h3;
                    split3(u, h1, h2, h3);
                    uint16_t *d = reinterpret_cast<uint16_t *>(dst);
                    if (form == 1) { d[e] = (uint16_t)h1; d[plane + e] = (uint16_t)h2; d[2 * plane + e] = (uint16_t)h3; }
                    else {
                        const long long b0 = (((long long)(x * N + y) * rows + r) * (K >> 4) + (k >> 4)) * 48 + (k & 15);
                        d[b0] = (uint16_t)h1; d[b0 + 16] = (uint16_t)h2; d[b0 + 32] = (uint16_t)h3;
                    }
                }
            }
    }
}

__global__ __launch_bounds__(256)
void split_weight_kernel(const float *__restrict__ src, uint16_t *__restrict__ dst, int rows, int K, int taps)
{
    const long long total = (long long)rows * K;
    const int Cin = taps > 0 ? K / taps : K;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int r = (int)(i / K), k = (int)(i - (long long)r * K);
        float v;
        if (taps == 1) v = src[i];
        else if (taps == 0) v = src[(long long)k * rows + r];                      // transposed source [K][rows]
        else { const int tap = k / Cin, c = k - tap * Cin; v = src[((long long)r * Cin + c) * taps + tap]; }
        unsigned h1, h2, h3;
        split3(v, h1, h2, h3);
        const long long b0 = ((long long)r * (K >> 4) + (k >> 4)) * 48 + (k & 15);
        dst[b0] = (uint16_t)h1; dst[b0 + 16] = (uint16_t)h2; dst[b0 + 32] = (uint16_t)h3;
    }
}


// ---------------------------------------------------------------------------------------------- fp16 pair / triple operands (round 5)
// XL_CONV_PAIR_F16 (csrc/xl_gemm_pair.hip): a weight w, scaled by the power of two 2^e of its matrix (max|w| 2^e in [2^14, 2^15)),
// is stored as {hi = fp16(x), lo = fp16(x - hi)}; layout [Z][rows][K/16][2][16] fp16, then 2 Z floats:
// the maxima of the matrices as float bits (pass 1, atomicMax on the bits of |w| - non-negative floats order like their bits)
// and the inverse scales 2^-e (pass 2).
__device__ __forceinline__ float pair_scale_of_max(unsigned maxBits)
{
    if (maxBits == 0u || (maxBits >> 23) == 0u) return 1.f;                       // all zero (or subnormal): no scaling
    const int E = (int)(maxBits >> 23) - 127;                                    // floor(log2(max))
    int e = 14 - E;
    if (e > 100) e = 100;
    if (e < -100) e = -100;
    return __builtin_bit_cast(float, (unsigned)(e + 127) << 23);
}
__device__ __forceinline__ void pair_hi_lo(float x, uint16_t &hi, uint16_t &lo)
{
    const _Float16 h = (_Float16)x;
    const _Float16 l = (_Float16)(x - (float)h);
    hi = __builtin_bit_cast(uint16_t, h); lo = __builtin_bit_cast(uint16_t, l);
}

template <int M>
__device__ __forceinline__ void wino_u_of(const float *__restrict__ w, int Cin, int o, int c, int dgrad, float (&u)[(M + 2) * (M + 2)])
{
    constexpr int N = M + 2;
    const float *src = w + ((long long)o * Cin + c) * 9;
    double g[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) g[a][b] = (double)(dgrad ? src[(2 - a) * 3 + (2 - b)] : src[a * 3 + b]);
    double t[N][3];
#pragma unroll
    for (int x = 0; x < N; ++x)
#pragma unroll
        for (int b = 0; b < 3; ++b)
            t[x][b] = WinoG<M>::g[x][0] * g[0][b] + WinoG<M>::g[x][1] * g[1][b] + WinoG<M>::g[x][2] * g[2][b];
#pragma unroll
    for (int x = 0; x < N; ++x)
#pragma unroll
        for (int y = 0; y < N; ++y)
            u[x * N + y] = (float)(t[x][0] * WinoG<M>::g[y][0] + t[x][1] * WinoG<M>::g[y][1] + t[x][2] * WinoG<M>::g[y][2]);
}

// PASS 0: the maxima per frequency; PASS 1: scale, split, store (and the inverse scales).  PASS 0 runs on few workgroups (a thread
// keeps the running maxima of its elements in registers, the workgroup combines them through an LDS table - no atomics on
// shared addresses - and issues one global atomicMax per frequency): a training loop re-packs every layer after every step.
// One launch covers a LIST of matrices (blockIdx.y = entry of `items`, a device table - xl_cnn_repack_pairs: a training loop
// re-packs every layer after every optimizer step, and 24 layers x (memset + two launches) of 25 us each were 1.4 ms of a
// 35 ms step) or the single matrix `one` (items == nullptr).
struct PairItem { const float *src; uint16_t *dst; int rows, K, kind, pad; };     // = xl_pair_item (include/crossloc_cnn.h)

template <int M, int PASS>
__global__ __launch_bounds__(256)
void wino_weight_pair_kernel(const PairItem *__restrict__ items, PairItem one)
{
    constexpr int N = M + 2, Z = N * N;
    extern __shared__ unsigned sTab[];                                // PASS 0: [256 threads][Z]
    const PairItem it = items ? items[blockIdx.y] : one;              // rows = Cout, K = Cin, kind = dgrad
    const float *__restrict__ w = it.src;
    uint16_t *__restrict__ dst = it.dst;
    const int Cout = it.rows, Cin = it.K, dgrad = it.kind;
    const int rows = dgrad ? Cin : Cout, K = dgrad ? Cout : Cin;
    const long long total = (long long)rows * K;
    unsigned *__restrict__ maxBits = reinterpret_cast<unsigned *>(dst + (long long)Z * total * 2);
    float *__restrict__ invScale = reinterpret_cast<float *>(maxBits + Z);
    unsigned mx[Z];
    if (PASS == 0) {
#pragma unroll
        for (int z = 0; z < Z; ++z) mx[z] = 0u;
    }
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int r = (int)(i / K), k = (int)(i - (long long)r * K);
        float u[Z];
        wino_u_of<M>(w, Cin, dgrad ? k : r, dgrad ? r : k, dgrad, u);
        if (PASS == 0) {
#pragma unroll
            for (int z = 0; z < Z; ++z) { const unsigned b = __builtin_bit_cast(unsigned, fabsf(u[z])); mx[z] = b > mx[z] ? b : mx[z]; }
        } else {
#pragma unroll
            for (int z = 0; z < Z; ++z) {
                const float sc = pair_scale_of_max(maxBits[z]);
                uint16_t hi, lo;
                pair_hi_lo(u[z] * sc, hi, lo);
                const long long b0 = (((long long)z * rows + r) * (K >> 4) + (k >> 4)) * 32 + (k & 15);
                dst[b0] = hi; dst[b0 + 16] = lo;
                if (i == 0) invScale[z] = 1.f / sc;
            }
        }
    }
    if (PASS == 0) {
#pragma unroll
        for (int z = 0; z < Z; ++z) sTab[threadIdx.x * Z + z] = mx[z];
        __syncthreads();
        for (int z = threadIdx.x; z < Z; z += 256) {
            unsigned m = 0u;
            for (int t = 0; t < 256; ++t) { const unsigned b = sTab[((t + z) & 255) * Z + z]; m = b > m ? b : m; }   // (staggered: no bank conflicts)
            if (m) atomicMax(&maxBits[z], m);
        }
    }
}

// the maxima slots of a list of matrices (Z words behind each matrix' pairs) to zero: one launch instead of a memset per matrix
__global__ void pair_zero_max_kernel(const PairItem *__restrict__ items, int Z)
{
    const PairItem it = items[blockIdx.x];
    unsigned *maxBits = reinterpret_cast<unsigned *>(it.dst + (long long)Z * it.rows * it.K * 2);
    for (int z = threadIdx.x; z < Z; z += blockDim.x) maxBits[z] = 0u;
}

template <int PASS>
__global__ __launch_bounds__(256)
void pair_weight_kernel(const PairItem *__restrict__ items, PairItem one)
{
    const PairItem it = items ? items[blockIdx.y] : one;              // kind = taps
    const float *__restrict__ src = it.src;
    uint16_t *__restrict__ dst = it.dst;
    const int rows = it.rows, K = it.K, taps = it.kind;
    const long long total = (long long)rows * K;
    unsigned *__restrict__ maxBits = reinterpret_cast<unsigned *>(dst + total * 2);
    float *__restrict__ invScale = reinterpret_cast<float *>(maxBits + 1);
    const int Cin = taps > 0 ? K / taps : K;
    unsigned m = 0u;
    const float sc = PASS ? pair_scale_of_max(maxBits[0]) : 1.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        if (PASS == 0) {                                               // (the maximum does not care about the order: a linear read)
            const unsigned b = __builtin_bit_cast(unsigned, fabsf(src[i]));
            m = b > m ? b : m;
            continue;
        }
        const int r = (int)(i / K), k = (int)(i - (long long)r * K);
        float v;
        if (taps == 1) v = src[i];
        else if (taps == 0) v = src[(long long)k * rows + r];
        else { const int tap = k / Cin, c = k - tap * Cin; v = src[((long long)r * Cin + c) * taps + tap]; }
        uint16_t hi, lo;
        pair_hi_lo(v * sc, hi, lo);
        const long long b0 = ((long long)r * (K >> 4) + (k >> 4)) * 32 + (k & 15);
        dst[b0] = hi; dst[b0 + 16] = lo;
        if (i == 0) invScale[0] = 1.f / sc;
    }
    if (PASS == 0) xl_wave_max_commit(__builtin_bit_cast(float, m), maxBits);      // (one atomic per wave at most)
}

// activation pairs [rows][K/8][2][8] fp16 {hi, (x - hi) * 2^11} of x = src * scale[0] (the layout the Winograd input transform writes)
__global__ __launch_bounds__(256)
void pair_activation_kernel(const float *__restrict__ src, uint16_t *__restrict__ dst, long long rows, int K, const float *__restrict__ scale)
{
    const long long total = rows * K;
    const float sc = scale[0];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / K;
        const int k = (int)(i - r * K);
        const float x = src[i] * sc;
        const _Float16 h = (_Float16)x;
        const _Float16 l = (_Float16)((x - (float)h) * 2048.f);
        const long long b0 = (r * (K >> 3) + (k >> 3)) * 16 + (k & 7);
        dst[b0] = __builtin_bit_cast(uint16_t, h); dst[b0 + 8] = __builtin_bit_cast(uint16_t, l);
    }
}

// one workgroup: bound = sum_i (sqrtN_i max|gamma_i| + max|beta_i|); s = the largest power of two with s * bound <= 2^14
// (kernel arguments hold 128 layers: longer lists run as a chain of launches that carry the sum in out[4..5] as a double)
struct PairGnList { const float *gamma[128]; const float *beta[128]; int C[128]; float sqrtN[128]; int n; };
__global__ __launch_bounds__(256)
void pair_scales_kernel(PairGnList L, float *__restrict__ out, int first, int last)
{
    __shared__ float sG[256], sB[256];
    __shared__ double sBound;
    if (threadIdx.x == 0) sBound = first ? 0.0 : *reinterpret_cast<double *>(out + 4);
    for (int i = 0; i < L.n; ++i) {
        float g = 0.f, b = 0.f;
        for (int c = threadIdx.x; c < L.C[i]; c += 256) { g = fmaxf(g, fabsf(L.gamma[i][c])); b = fmaxf(b, fabsf(L.beta[i][c])); }
        sG[threadIdx.x] = g; sB[threadIdx.x] = b;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) { sG[threadIdx.x] = fmaxf(sG[threadIdx.x], sG[threadIdx.x + s]); sB[threadIdx.x] = fmaxf(sB[threadIdx.x], sB[threadIdx.x + s]); }
            __syncthreads();
        }
        if (threadIdx.x == 0) sBound += (double)L.sqrtN[i] * (double)sG[0] + (double)sB[0];
        __syncthreads();
    }
    if (threadIdx.x == 0) *reinterpret_cast<double *>(out + 4) = sBound;
    if (threadIdx.x == 0 && last) {
        double bound = sBound;
        if (!(bound > 1e-30)) bound = 1.0;
        if (!(bound < 1e30)) bound = 1e30;                  // (a non-finite parameter: the results are garbage whatever the scale)
        int e;
        const double m = frexp(bound, &e);                   // bound = m 2^e, m in [0.5, 1): 2^(14 - e) bound <= 2^14
        (void)m;
        const float s = ldexpf(1.f, 14 - e), sv = ldexpf(1.f, 14 - e - 8);
        out[0] = s; out[1] = 1.f / s; out[2] = sv; out[3] = 1.f / sv;
    }
}

template <int M>
int launch_wino_pairs(const PairItem *items, const PairItem &one, int n, long long maxTotal, hipStream_t st)
{
    constexpr int Z = (M + 2) * (M + 2);
    long long blocks = (maxTotal + 255) / 256, blocks0 = blocks;
    if (blocks > 4096) blocks = 4096;
    if (blocks0 > 256) blocks0 = 256;                             // PASS 0: one workgroup per CU, a few elements per thread
    const size_t lds0 = sizeof(unsigned) * 256 * Z;
    // (64 KB of dynamic LDS for m = 6: set on every call - idempotent, and cheaper than tracking the attribute per device here)
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(wino_weight_pair_kernel<M, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds0) != hipSuccess)
        return XL_ERR_HIP;
    hipLaunchKernelGGL((wino_weight_pair_kernel<M, 0>), dim3((unsigned)blocks0, n), dim3(256), lds0, st, items, one);
    hipLaunchKernelGGL((wino_weight_pair_kernel<M, 1>), dim3((unsigned)blocks, n), dim3(256), 0, st, items, one);
    return hipGetLastError() == hipSuccess ? XL_OK : XL_ERR_HIP;
}
int launch_plain_pairs(const PairItem *items, const PairItem &one, int n, long long maxTotal, hipStream_t st)
{
    long long blocks = (maxTotal + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    // (the maxima pass on few workgroups: thousands of waves arriving at once would all find the slot still empty and serialise
    //  their atomics - 46 us for a 512 x 512 matrix against 5 for the split pass)
    hipLaunchKernelGGL(pair_weight_kernel<0>, dim3((unsigned)(blocks < 32 ? blocks : 32), n), dim3(256), 0, st, items, one);
    hipLaunchKernelGGL(pair_weight_kernel<1>, dim3((unsigned)blocks, n), dim3(256), 0, st, items, one);
    return hipGetLastError() == hipSuccess ? XL_OK : XL_ERR_HIP;
}
}  // namespace

extern "C" {

int xl_cnn_pack_wino_weight(const float *w, void *dst, int Cout, int Cin, int m, int dgrad, int form, void *stream)
{
    if (!w || !dst || Cout < 1 || Cin < 1 || (m != 2 && m != 4 && m != 6) || form < 0 || form > 2) return XL_ERR_ARG;
    const int K = dgrad ? Cout : Cin;
    if (form == 2 && K % 16 != 0) return XL_ERR_ARG;
    const long long total = (long long)Cout * Cin;
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipStream_t st = (hipStream_t)stream;
    if (m == 2) hipLaunchKernelGGL(wino_weight_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, st, w, dst, Cout, Cin, dgrad ? 1 : 0, form);
    else if (m == 4) hipLaunchKernelGGL(wino_weight_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, st, w, dst, Cout, Cin, dgrad ? 1 : 0, form);
    else hipLaunchKernelGGL(wino_weight_kernel<6>, dim3((unsigned)blocks), dim3(256), 0, st, w, dst, Cout, Cin, dgrad ? 1 : 0, form);
    return hipGetLastError() == hipSuccess ? XL_OK : XL_ERR_HIP;
}

int xl_cnn_split_weight(const float *src, void *dst, int rows, int K, int taps, void *stream)
{
    if (!src || !dst || rows < 1 || K < 16 || K % 16 != 0 || (taps != 0 && taps != 1 && taps != 9) || (taps > 0 && K % taps != 0)) return XL_ERR_ARG;
    const long long total = (long long)rows * K;
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(split_weight_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, (uint16_t *)dst, rows, K, taps);
    return hipGetLastError() == hipSuccess ? XL_OK : XL_ERR_HIP;
}

int xl_cnn_pack_wino_weight_pair(const float *w, void *dst, int Cout, int Cin, int m, int dgrad, void *stream)
{
    if (!w || !dst || Cout < 1 || Cin < 1 || (m != 4 && m != 6)) return XL_ERR_ARG;
    const int K = dgrad ? Cout : Cin, Z = (m + 2) * (m + 2);
    if (K % 16 != 0) return XL_ERR_ARG;
    const long long total = (long long)Cout * Cin;
    hipStream_t st = (hipStream_t)stream;
    uint16_t *d = (uint16_t *)dst;
    if (hipMemsetAsync(d + (long long)Z * total * 2, 0, sizeof(unsigned) * Z, st) != hipSuccess) return XL_ERR_HIP;
    const PairItem one = { w, d, Cout, Cin, dgrad ? 1 : 0, 0 };
    return m == 4 ? launch_wino_pairs<4>(nullptr, one, 1, total, st) : launch_wino_pairs<6>(nullptr, one, 1, total, st);
}

int xl_cnn_pair_weight(const float *src, void *dst, int rows, int K, int taps, void *stream)
{
    if (!src || !dst || rows < 1 || K < 16 || K % 16 != 0 || (taps != 0 && taps != 1 && taps != 9) || (taps > 0 && K % taps != 0)) return XL_ERR_ARG;
    const long long total = (long long)rows * K;
    hipStream_t st = (hipStream_t)stream;
    uint16_t *d = (uint16_t *)dst;
    if (hipMemsetAsync(d + total * 2, 0, sizeof(unsigned), st) != hipSuccess) return XL_ERR_HIP;
    const PairItem one = { src, d, rows, K, taps, 0 };
    return launch_plain_pairs(nullptr, one, 1, total, st);
}

int xl_cnn_item_size(int which)
{
    return which == 0 ? (int)sizeof(xl_pair_item) : which == 1 ? (int)sizeof(xl_gnb_params_item) : -1;
}

int xl_cnn_repack_pairs(const xl_pair_item *items_dev, int n, int m, long long max_elements, void *stream)
{
    static_assert(sizeof(xl_pair_item) == sizeof(PairItem), "xl_pair_item and the kernels' PairItem are one layout");
    if (n == 0) return XL_OK;
    if (!items_dev || n < 0 || n > 65535 || (m != 0 && m != 4 && m != 6) || max_elements < 1) return XL_ERR_ARG;
    const PairItem *items = reinterpret_cast<const PairItem *>(items_dev);
    hipStream_t st = (hipStream_t)stream;
    const PairItem none = { nullptr, nullptr, 0, 0, 0, 0 };
    hipLaunchKernelGGL(pair_zero_max_kernel, dim3(n), dim3(64), 0, st, items, m ? (m + 2) * (m + 2) : 1);
    if (m == 6) return launch_wino_pairs<6>(items, none, n, max_elements, st);
    if (m == 4) return launch_wino_pairs<4>(items, none, n, max_elements, st);
    return launch_plain_pairs(items, none, n, max_elements, st);
}

int xl_cnn_pair_activation(const float *src, void *dst, long long rows, int K, const float *scale, void *stream)
{
    if (!src || !dst || !scale || rows < 1 || K < 16 || K % 16 != 0) return XL_ERR_ARG;
    long long blocks = (rows * K + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(pair_activation_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, (uint16_t *)dst, rows, K, scale);
    return hipGetLastError() == hipSuccess ? XL_OK : XL_ERR_HIP;
}

int xl_cnn_pair_scales(const float *const *gamma, const float *const *beta, const int *C, const float *sqrtN, int n, float *out, void *stream)
{
    if (!out || n < 0 || (((uintptr_t)out) & 7) || (n > 0 && (!gamma || !beta || !C || !sqrtN))) return XL_ERR_ARG;
    for (int i = 0; i < n; ++i)
        if (!gamma[i] || !beta[i] || C[i] < 1 || !(sqrtN[i] >= 0.f)) return XL_ERR_ARG;
    for (int i0 = 0; i0 == 0 || i0 < n; i0 += 128) {
        PairGnList L;
        L.n = n - i0 < 128 ? n - i0 : 128;
        for (int i = 0; i < L.n; ++i) { L.gamma[i] = gamma[i0 + i]; L.beta[i] = beta[i0 + i]; L.C[i] = C[i0 + i]; L.sqrtN[i] = sqrtN[i0 + i]; }
        hipLaunchKernelGGL(pair_scales_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, L, out, i0 == 0 ? 1 : 0, i0 + 128 >= n ? 1 : 0);
    }
    return hipGetLastError() == hipSuccess ? XL_OK : XL_ERR_HIP;
}

}  // extern "C"
