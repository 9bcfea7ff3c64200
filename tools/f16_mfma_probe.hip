// Probe (round 5): does v_mfma_f32_32x32x16_f16 on gfx950 keep SUBNORMAL fp16 inputs, and how exact is a two-term fp16 product?
//   hipcc --offload-arch=gfx950 -O2 -o f16_mfma_probe tools/f16_mfma_probe.hip && ./f16_mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void probe(float aval, float bval, float *out)
{
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)aval; b[i] = (_Float16)bval; }
    f32x16 acc = { 0 };
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = acc[0];
}

int main()
{
    float *d;
    hipMalloc(&d, 4);
    const float cases[][2] = { { 1.0f, 1.0f }, { 9.5367431640625e-07f /* 2^-20 */, 1024.f }, { 5.9604644775390625e-08f /* 2^-24 */, 4096.f },
                               { 9.5367431640625e-07f, 9.5367431640625e-07f }, { 65504.f, 65504.f } };
    for (auto &c : cases) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, c[0], c[1], d);
        float h;
        hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("a = %.9g  b = %.9g  K = 16:  mfma = %.9g   expected = %.9g\n", c[0], c[1], h, 16.0 * (double)c[0] * (double)c[1]);
    }
    return 0;
}
