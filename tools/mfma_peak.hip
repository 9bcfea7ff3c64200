// Sustained fp32 MFMA rate and shader clock of the device under an MFMA-only load: the practical ceiling the
// implicit-GEMM convolution is compared with in DESIGN.md (the 157.3 TFLOP/s datasheet peak assumes 2.4 GHz).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_peak tools/mfma_peak.hip && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float *out, long long *clk, int iters)
{
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

template <int NACC>
void run(int blocks, int iters)
{
    float *out; long long *clk;
    hipMalloc(&out, sizeof(float) * blocks * 256);
    hipMalloc(&clk, sizeof(long long) * 2 * blocks);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(256), 0, 0, out, clk, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(256), 0, 0, out, clk, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(2 * blocks);
    hipMemcpy(h.data(), clk, sizeof(long long) * 2 * blocks, hipMemcpyDeviceToHost);
    double cs = 0, ws = 0;
    for (int i = 0; i < blocks; ++i) { cs += h[2 * i]; ws += h[2 * i + 1]; }
    const double flop = 2.0 * 32 * 32 * 2 * 16.0 * NACC * iters * 4.0 * blocks;      // 4 waves per block
    const double mfmaPerWave = 16.0 * NACC * iters;
    printf("blocks %d acc %d: %.3f ms  %.1f TFLOP/s | clock64/wall(100MHz) = %.1f MHz-equivalent ticks ratio %.3f | clock64 ticks per MFMA %.2f\n",
           blocks, NACC, ms, flop / ms / 1e9, cs / ws * 100.0, cs / ws, cs / blocks / mfmaPerWave);
    hipFree(out); hipFree(clk);
}

int main()
{
    run<4>(256, 20000);
    run<4>(512, 20000);
    run<4>(1024, 10000);
    run<8>(256, 10000);
    run<1>(256, 40000);
    return 0;
}
