// crossloc_hip: weight re-pack kernels - what a plan runs once per weight version (after load_state_dict, and after
// every optimizer step of a training loop): the Winograd transform U = G g G^T (float64 inside, rounded once) and the exact
// three-term bf16 split of the operands of the split-pipe kernels.  One launch per layer, no library calls.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/crossloc_cnn.h"
#include "../../include/crossloc_dsac.h"   // status codes

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

}  // extern "C"
