// xl_optim.hip — fused multi-tensor Adam step (reference: torch.optim.Adam created in
// /root/reference/utils/learning.py:390, stepped at train_single_task.py:299).  One launch updates every
// parameter tensor: a device-side table of (param, grad, exp_avg, exp_avg_sq, n) chunks, one workgroup per chunk.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/crossloc_optim.h"
#include "../../include/crossloc_dsac.h"

namespace {

__global__ __launch_bounds__(256)
void adam_kernel(const xl_adam_chunk *__restrict__ chunks, float lr, float beta1, float beta2, float eps,
                 float weight_decay, float bc1, float sqrt_bc2)
{
    const xl_adam_chunk c = chunks[blockIdx.x];
    const float step_size = lr / bc1;
    for (int i = threadIdx.x; i < c.n; i += 256) {
        float g = c.grad[i];
        const float p = c.param[i];
        if (weight_decay != 0.f) g += weight_decay * p;
        const float m = beta1 * c.exp_avg[i] + (1.f - beta1) * g;            // exp_avg.lerp_(grad, 1 - beta1)
        const float v = beta2 * c.exp_avg_sq[i] + (1.f - beta2) * g * g;     // exp_avg_sq.mul_(b2).addcmul_(g, g, 1-b2)
        c.exp_avg[i] = m;
        c.exp_avg_sq[i] = v;
        const float denom = sqrtf(v) / sqrt_bc2 + eps;
        c.param[i] = p - step_size * (m / denom);
    }
}

}  // namespace

extern "C" int xl_adam_step(const xl_adam_chunk *chunks_dev, int n_chunks, float lr, float beta1, float beta2, float eps,
                            float weight_decay, float bias_correction1, float bias_correction2, void *stream)
{
    if (!chunks_dev || n_chunks <= 0 || bias_correction1 <= 0.f || bias_correction2 <= 0.f) return XL_ERR_ARG;
    hipLaunchKernelGGL(adam_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, chunks_dev, lr, beta1, beta2, eps,
                       weight_decay, bias_correction1, sqrtf(bias_correction2));
    return hipGetLastError() == hipSuccess ? XL_OK : XL_ERR_HIP;
}
