/* rlipv2_norm.h -- C ABI of the fused residual-add + LayerNorm kernels (gfx950, bf16 data, f32 statistics).
 *
 * The post-norm encoder layer of RLIPv2-ParSeDA computes  src = norm1(src + attn(src))  and
 * src = norm2(src + ffn(src))  on [N*S, 256] = [88 892, 256] tokens (reference:
 * DeformableTransformerEncoderLayer.forward / forward_ffn, models/dab_deformable/deformable_transformer.py:
 * 1261-1300; twin models/deformable_transformer.py:719-758).  As separate add + LayerNorm kernels that is
 * 5 passes over the 45 MB tensor forward and PyTorch's LayerNorm kernels run at ~1/8 of the HBM rate at
 * this width (94 us forward / 188 us backward measured); here each direction is ONE pass:
 *
 *   forward : y = LN(a + b) * gamma + beta,  mean / rstd per row saved          (reads a, b; writes y)
 *   backward: dx = dLN/d(a+b) (the gradient of BOTH a and b), dgamma, dbeta       (reads dy, a, b; writes dx)
 *
 * a, b, y, dy, dx: [rows, C] bf16 contiguous, 16-byte aligned; gamma, beta, dgamma, dbeta: [C] bf16;
 * mean, rstd: [rows] float32.  C must be 256.  b may be NULL (plain LayerNorm).  The sum a + b and all
 * statistics are float32.  Nothing allocates or synchronises; work is enqueued on `stream`.
 * Return value: 0 or an msda_status code (rlipv2_msda.h).
 */
#ifndef RLIPV2_NORM_H
#define RLIPV2_NORM_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

int add_layernorm_supported(long rows, int C);

int add_layernorm_forward_bf16(const void *a, const void *b, const void *gamma, const void *beta, long rows, int C,
                               float eps, void *y, float *mean, float *rstd, void *stream);

/* workspace: add_layernorm_workspace_bytes(rows, C) bytes of device memory (per-workgroup partial sums of
 * dgamma / dbeta). */
size_t add_layernorm_workspace_bytes(long rows, int C);

int add_layernorm_backward_bf16(const void *dy, const void *a, const void *b, const void *gamma, const float *mean,
                                const float *rstd, long rows, int C, void *dx, void *dgamma, void *dbeta,
                                void *workspace, size_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif
