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

/* ---- the widths of the Swin backbones (csrc/layernorm_wide.hip; round 5, not yet run on hardware) ------------------------
 * The pre-norm residual block of the Swin Transformer (reference models/swin/swin_transformer.py:304-403):
 *     x = x + drop_path(branch);  y = norm(x)        -- as add + layer_norm: 5 passes over the tensor forward
 * here ONE pass per direction, for C in {96, 128, 192, 384, 512, 768, 1024, 1536} (256: the kernels above):
 *   forward : s = a + b (rounded to bf16, written to `sum` when given), y = LN(s) * gamma + beta, mean / rstd saved
 *             (b NULL: plain LayerNorm of a; sum NULL with b: post-norm form, the sum stays float32)
 *   backward: dx = dLN/ds (dy) + dsum   (dsum NULL: no gradient reaches the sum on the residual path)
 *             x = the tensor the statistics were taken from (the saved `sum`, or `a` when b was NULL)
 * No gamma / beta gradients: the reference freezes every norm of its Swin backbones (models/swin/backbone.py:66-69); a
 * caller that trains them keeps PyTorch's op.  Same conventions as above (bf16 data, 16-byte aligned, float32 statistics). */
int layernorm_wide_supported(long rows, int C);

int layernorm_wide_forward_bf16(const void *a, const void *b, const void *gamma, const void *beta, long rows, int C, float eps,
                                void *y, void *sum, float *mean, float *rstd, void *stream);

int layernorm_wide_backward_bf16(const void *dy, const void *dsum, const void *x, const void *gamma, const float *mean,
                                 const float *rstd, long rows, int C, void *dx, void *stream);

#ifdef __cplusplus
}
#endif
#endif
