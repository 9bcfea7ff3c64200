// Sustained rate of the 16-bit matrix pipe under an MFMA-only load with RANDOM operands, fp16 against bf16: how much of the gap
// between round 5's fp16-pair GEMM (1010 TFLOP/s) and round 4's six-pass bf16 GEMM (1335 TFLOP/s) is the multipliers' power.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma16_power_probe tools/mfma16_power_probe.hip && /tmp/mfma16_power_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned rnd(unsigned &s) { s = s * 1664525u + 1013904223u; return s; }

// MODE 0: fp16 x fp16, MODE 1: bf16 x bf16, MODE 2: one fp16 pass + two bf16 passes per K-step (the mixed scheme)
template <int MODE>
__global__ __launch_bounds__(512) void loop(float *out, long long *clk, int iters, int zeroData)
{
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    u32x4 A[6], B[6];                                                  // six operand sets, cycled: consecutive MFMAs see different data
    for (int k = 0; k < 6; ++k)
        for (int j = 0; j < 4; ++j) {
            // two 16-bit values per word with sign and a random mantissa, exponent kept moderate (finite, no overflow of the sums)
            const unsigned ra = rnd(s), rb = rnd(s);
            const unsigned ea = (MODE == 1) ? 0x3f803f80u : 0x3c003c00u;   // 1.0 in bf16 / fp16
            const unsigned ma = (MODE == 1) ? 0x807f807fu : 0x83ff83ffu;   // sign + mantissa bits
            A[k][j] = zeroData ? 0u : (ea | (ra & ma));
            B[k][j] = zeroData ? 0u : (ea | (rb & ma));
        }
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int k = (u * 8 + i) % 6;
                const bool half = MODE == 0 || (MODE == 2 && u == 0);
                if (half) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A[k]), __builtin_bit_cast(f16x8, B[(k + 1) % 6]), acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[k]), __builtin_bit_cast(bf16x8, B[(k + 1) % 6]), acc[i], 0, 0, 0);
            }
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    float t = 0.f;
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 16; ++r) t += acc[i][r];
    out[blockIdx.x * 512 + threadIdx.x] = t;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

template <int MODE>
void run(const char *name, int iters, int zeroData)
{
    const int blocks = 256;
    float *out; long long *clk;
    hipMalloc(&out, sizeof(float) * blocks * 512);
    hipMalloc(&clk, sizeof(long long) * 2 * blocks);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(loop<MODE>, dim3(blocks), dim3(512), 0, 0, out, clk, iters, zeroData);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int w = 0; w < 10; ++w) hipLaunchKernelGGL(loop<MODE>, dim3(blocks), dim3(512), 0, 0, out, clk, iters, zeroData);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ms /= 10.f;
    std::vector<long long> h(2 * blocks);
    hipMemcpy(h.data(), clk, sizeof(long long) * 2 * blocks, hipMemcpyDeviceToHost);
    double cs = 0, ws = 0;
    for (int i = 0; i < blocks; ++i) { cs += h[2 * i]; ws += h[2 * i + 1]; }
    const double flop = 2.0 * 32 * 32 * 16 * 24.0 * iters * 8.0 * blocks;      // 8 waves per block, 24 MFMAs per trip
    printf("%-34s %s: %.3f ms per launch  %7.1f TFLOP/s = %.3f of 2500 | shader clock %.0f MHz\n", name, zeroData ? "zeros " : "random", ms,
           flop / ms / 1e9, flop / ms / 1e9 / 2500.0, cs / ws * 100.0);
    hipFree(out); hipFree(clk);
}

int main()
{
    const int iters = 6000;                                           // ~1.5 ms per launch at 1000 TFLOP/s: the GEMM's launch length
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("fp16 x fp16 (three passes)", iters, 0);
        run<1>("bf16 x bf16 (three passes)", iters, 0);
        run<2>("1 fp16 + 2 bf16 passes", iters, 0);
    }
    run<0>("fp16 x fp16", iters, 1);
    run<1>("bf16 x bf16", iters, 1);
    return 0;
}
