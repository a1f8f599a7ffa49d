/* rlipv2_optim.h -- C ABI of the fused optimiser step of the bf16 train step (gfx950).
 *
 * The reference's step is  clip_grad_norm_(0.1) + AdamW  over all trainable parameters (engine.py:170-172,
 * main.py:523-541: three parameter groups by name with their own learning rate, weight decay 1e-4).  With
 * bf16 parameters and float32 master weights that is, as PyTorch ops: bf16->f32 gradient copies, a
 * multi-tensor norm, a multi-tensor scale, the fused AdamW and f32->bf16 parameter copies -- ~90 launches
 * and 52 bytes of HBM traffic per parameter.  Here it is two launches and 30 bytes per parameter:
 *
 *   adamw_grad_sqnorm_bf16 : sum of squares of all bf16 gradients (float32 accumulation) -> one float
 *   adamw_step_bf16        : g *= min(1, max_norm / (sqrt(sum) + 1e-6));  AdamW on the float32 master weight
 *                            and moments;  bf16 parameter = round-to-nearest-even(master)
 *
 * AdamW as torch.optim.AdamW (decoupled weight decay, bias-corrected, no amsgrad):
 *   p *= 1 - lr*wd;  m = b1*m + (1-b1)*g;  v = b2*v + (1-b2)*g*g;
 *   p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
 *
 * The tensors are described by a device-resident table; work is split into fixed-size chunks listed in a
 * second device-resident table (both built once by the caller; rebuilt only if a pointer changes).
 */
#ifndef RLIPV2_OPTIM_H
#define RLIPV2_OPTIM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ADAMW_CHUNK 16384 /* elements per workgroup */
#define ADAMW_MAX_GROUPS 8

typedef struct adamw_tensor {
    const void *grad; /* bf16 [numel] */
    float *master;    /* float32 [numel] */
    float *exp_avg;
    float *exp_avg_sq;
    void *param; /* bf16 [numel] */
    int64_t numel;
    int64_t group; /* index into the groups array of adamw_step_bf16 */
} adamw_tensor;

typedef struct adamw_chunk {
    int32_t tensor; /* index into the tensor table */
    int32_t index;  /* chunk number inside that tensor: elements [index*ADAMW_CHUNK, ...) */
} adamw_chunk;

/* Per-group scalars, all derived on the host in double precision and rounded once (1 - 0.999f evaluated in
 * float32 is off by 5e-5, which would show in every update). */
typedef struct adamw_group {
    float beta1, beta2;
    float one_minus_beta1, one_minus_beta2;
    float eps;
    float decay;                 /* 1 - lr * weight_decay */
    float step_size;             /* lr / (1 - beta1^t) */
    float bias_correction2_sqrt; /* sqrt(1 - beta2^t) */
} adamw_group;

int adamw_abi_sizes(int *tensor_bytes, int *chunk_bytes, int *group_bytes, int *chunk_elements);

/* *sqnorm (device float) = sum over all listed chunks of g^2; zeroed inside (async memset). */
int adamw_grad_sqnorm_bf16(const adamw_tensor *tensors, const adamw_chunk *chunks, int n_chunks, float *sqnorm,
                           void *stream);

/* max_norm <= 0 disables clipping (sqnorm may then be NULL). */
int adamw_step_bf16(const adamw_tensor *tensors, const adamw_chunk *chunks, int n_chunks, const float *sqnorm,
                    float max_norm, const adamw_group *groups, int n_groups, void *stream);

/* As adamw_step_bf16 with every gradient multiplied by grad_scale (> 0) first -- in the norm as in the update.  The
 * data-parallel step leaves the all-reduced gradients as SUMS and passes grad_scale = 1 / world_size here instead of
 * making one more pass over the 425 MB gradient buffer (DistributedDataParallel averages, reference main.py:515-517). */
int adamw_step_scaled_bf16(const adamw_tensor *tensors, const adamw_chunk *chunks, int n_chunks, const float *sqnorm,
                           float max_norm, float grad_scale, const adamw_group *groups, int n_groups, void *stream);

#ifdef __cplusplus
}
#endif
#endif
