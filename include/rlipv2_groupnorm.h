/* rlipv2_groupnorm.h -- C ABI of GroupNorm(32, 256) over the token-major feature pyramid (gfx950, bf16 data, f32 statistics).
 *
 * RLIPv2-ParSeDA projects every backbone level to 256 channels and normalises it with `nn.GroupNorm(32, hidden_dim)`
 * (input_proj, reference models/hoi.py:1936-1957; twin models/deformable_detr.py), then flattens the levels to
 * [N, sum(H*W), 256] for the encoder (models/dab_deformable/deformable_transformer.py:520-547).  On channels-last data
 * PyTorch's GroupNorm converts to NCHW and back (four copies of the level-0 map per step) and the flatten is a `cat`;
 * here all levels are normalised in ONE launch pair per direction, straight from the token-major projection outputs
 * into their slices of the flattened tensor:
 *
 *   forward : out[n, start_l + t, c] = (x_l[n, t, c] - mean[l, n, g]) * rstd[l, n, g] * gamma_l[c] + beta_l[c]
 *   backward: dx_l, dgamma_l, dbeta_l from dy = d out  (the usual GroupNorm gradient, statistics over H*W x 8 channels)
 *
 * x_l, dx_l: [N, hw_l, 256] bf16 contiguous; out, dy: [N, S, 256] bf16 contiguous with S = sum(hw_l) and level l at rows
 * [start_l, start_l + hw_l); gamma_l, beta_l, dgamma_l, dbeta_l: [256] bf16; mean, rstd: [levels, N, 32] float32.
 * levels <= 4.  All pointers 16-byte aligned.  `x`, `gamma`, ... are HOST arrays of `levels` device pointers.
 * Deterministic (fixed summation order, no atomics).  Nothing allocates or synchronises; work is enqueued on `stream`.
 * Return value: 0 or an msda_status code (rlipv2_msda.h).
 */
#ifndef RLIPV2_GROUPNORM_H
#define RLIPV2_GROUPNORM_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

int groupnorm_tokens_supported(int channels, int groups, int levels);

/* device workspace for either direction */
size_t groupnorm_tokens_workspace_bytes(int N, const int *hw, int levels);

int groupnorm_tokens_forward_bf16(const void *const *x, const int *hw, int levels, int N, const void *const *gamma,
                                  const void *const *beta, float eps, void *out, float *mean, float *rstd,
                                  void *workspace, size_t workspace_bytes, void *stream);

int groupnorm_tokens_backward_bf16(const void *dy, const void *const *x, const int *hw, int levels, int N,
                                   const void *const *gamma, const float *mean, const float *rstd, void *const *dx,
                                   void *const *dgamma, void *const *dbeta, void *workspace, size_t workspace_bytes,
                                   void *stream);

#ifdef __cplusplus
}
#endif
#endif
