/*
 * rlipv2_msda_cpu.h -- C ABI of the CPU twins of the MSDA operator (librlipv2_msda_cpu.so, built with g++ -fopenmp from
 * rlipv2_amd/csrc/msda_cpu.cpp; no HIP, no GPU).
 *
 * What it replaces: the CPU side of the reference's extension module "MultiScaleDeformableAttention".  The reference
 * declares `ms_deform_attn_cpu_forward / _backward` (models/ops/src/cpu/ms_deform_attn_cpu.h:14-31) but their bodies raise
 * (models/ops/src/cpu/ms_deform_attn_cpu.cpp:24,40), and the dispatcher raises "Not implemented on the CPU" for CPU
 * tensors (models/ops/src/ms_deform_attn.h:54); on the CPU the reference's models run on the pure-PyTorch formulation
 * `ms_deform_attn_core_pytorch` (models/ops/functions/ms_deform_attn_func.py:45-65: per-level F.grid_sample, bilinear,
 * zero padding, align_corners=False) -- BASELINE config 1.  SURVEY.md section 8b asks the drop-in to accept CPU tensors
 * through CPU twins of the two entry points; these are they.  Same operands as msda_forward / msda_backward
 * (include/rlipv2_msda.h) minus the stream; all pointers are HOST pointers, contiguous row-major:
 *      value [N, S, M, D], spatial_shapes int64 [L, 2] (H, W), level_start int64 [L],
 *      sampling_loc [N, Lq, M, L, P, 2] (x, y), attn_weight [N, Lq, M, L, P], out / grad_out [N, Lq, M*D]
 * dtype: every tensor float32 (MSDA_CPU_F32) or float64 (MSDA_CPU_F64) -- the reference's two types.  (bfloat16 CPU
 * tensors are widened by the Python binding.)  Sampling rule and gradients: ms_deform_im2col_cuda.cuh:33-159, 282-291.
 * Outputs are caller-allocated and fully written (grad_value is zero-filled inside).  Synchronous; threads = OpenMP's.
 * Results do not depend on the number of threads (every output element has one writer and a fixed summation order).
 */
#ifndef RLIPV2_MSDA_CPU_H
#define RLIPV2_MSDA_CPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { MSDA_CPU_F32 = 0, MSDA_CPU_F64 = 1 };
enum { MSDA_CPU_ERR_DTYPE = -1, MSDA_CPU_ERR_NULL = -2, MSDA_CPU_ERR_DIMS = -3, MSDA_CPU_ERR_LEVELS = -4 };

/* replaces ms_deform_attn_cpu_forward (models/ops/src/cpu/ms_deform_attn_cpu.h:14-20) */
int msda_forward_cpu(int dtype, const void *value, const int64_t *spatial_shapes, const int64_t *level_start,
                     const void *sampling_loc, const void *attn_weight, int N, int S, int M, int D, int L, int Lq, int P,
                     void *out);

/* replaces ms_deform_attn_cpu_backward (models/ops/src/cpu/ms_deform_attn_cpu.h:22-31) */
int msda_backward_cpu(int dtype, const void *value, const int64_t *spatial_shapes, const int64_t *level_start,
                      const void *sampling_loc, const void *attn_weight, const void *grad_out, int N, int S, int M, int D, int L,
                      int Lq, int P, void *grad_value, void *grad_sampling_loc, void *grad_attn_weight);

const char *msda_cpu_strerror(int status);
int msda_cpu_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif
