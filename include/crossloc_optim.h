/*
 * crossloc_optim.h — fused Adam step (libcrossloc_hip.so), SURVEY.md §8f row f1.
 * Replaces the per-tensor `optimizer.step()` of torch.optim.Adam (/root/reference/utils/learning.py:390,
 * train_single_task.py:299; default betas (0.9, 0.999), eps 1e-8, no amsgrad) with one launch over a device
 * table of chunks.  Update rule (identical to torch's single-tensor path):
 *   m = b1*m + (1-b1)*g;  v = b2*v + (1-b2)*g*g;  p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
 * with bc1 = 1 - b1^t, bc2 = 1 - b2^t supplied by the host.
 */
#ifndef CROSSLOC_OPTIM_H
#define CROSSLOC_OPTIM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct xl_adam_chunk {
    float *param;
    const float *grad;
    float *exp_avg;
    float *exp_avg_sq;
    int32_t n;          /* elements in this chunk (<= 65536 keeps the workgroups balanced) */
    int32_t pad;
} xl_adam_chunk;

int xl_adam_step(const xl_adam_chunk *chunks_dev, int n_chunks, float lr, float beta1, float beta2, float eps,
                 float weight_decay, float bias_correction1, float bias_correction2, void *stream);

#ifdef __cplusplus
}
#endif
#endif
