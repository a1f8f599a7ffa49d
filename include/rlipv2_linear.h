/* rlipv2_linear.h -- C ABI of the token-major Linear weight-gradient kernel (gfx950, bf16 MFMA).
 *
 * The encoder of RLIPv2-ParSeDA applies nn.Linear layers to the flattened multi-scale feature map,
 * [N*S, C] with N*S = 88 892 rows at batch 4 x 800x1333 (reference: MSDeformAttn.value_proj /
 * sampling_offsets / attention_weights / output_proj, models/ops/modules/ms_deform_attn.py:59-62,
 * and DeformableTransformerEncoderLayer.linear1 / linear2, models/dab_deformable/
 * deformable_transformer.py:571-576).  In the backward pass each of them needs
 *
 *      dW[M, K] = dY[T, M]^T . X[T, K]        db[M] = column sums of dY
 *
 * -- a GEMM whose reduction dimension is the 88 892 tokens and whose output is a small square.  The
 * library GEMM handles that shape at 4-9 % of the HBM rate (measured, tools/gemm_shapes.py), and the
 * bias gradient re-reads dY once more; this kernel streams dY and X once, split over the tokens.
 *
 * Plain pointers and sizes; nothing allocates, nothing synchronises; work is enqueued on `stream`.
 */
#ifndef RLIPV2_LINEAR_H
#define RLIPV2_LINEAR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bytes of scratch device memory linear_wgrad_bf16 needs for this problem (0 if the shape is not
 * supported: M and K must be multiples of 128, T >= 1). */
size_t linear_wgrad_workspace_bytes(int T, int M, int K);

/* 1 if (T, M, K) runs on the MFMA kernel. */
int linear_wgrad_supported(int T, int M, int K);

/* dW = dY^T X and db = sum_t dY[t, :].
 *   dy [T, M] bf16 row-major, x [T, K] bf16 row-major (16-byte aligned, contiguous),
 *   dw [M, K] and db [M]: bf16 when out_f32 == 0, float32 otherwise; db may be NULL.
 *   workspace: linear_wgrad_workspace_bytes(T, M, K) bytes of device memory.
 * Accumulation is float32 throughout.  Returns 0, or an msda_status code (rlipv2_msda.h). */
int linear_wgrad_bf16(const void *dy, const void *x, int T, int M, int K, void *dw, void *db, int out_f32,
                      void *workspace, size_t workspace_bytes, void *stream);

/* 1 if (T, N, K) runs on the expand kernel: K == 256, N a multiple of 64, T >= 1, T * N * 2 < 2^32. */
int linear_expand_supported(int T, int N, int K);

/* c[t, n] = epilogue(sum_k a[t, k] * b[n, k]) for the Linears that widen the 256-channel token tensors:
 *   a [T, K] bf16 row-major, b [N, K] bf16 row-major (nn.Linear's weight layout), c [T, N] bf16; all 16-byte aligned;
 *   bias [N] bf16 or NULL: added before the activation (the `F.linear(x, w, b)` of `DeformableTransformerEncoderLayer.
 *   forward_ffn`, dab_deformable/deformable_transformer.py:1285-1289);
 *   relu != 0: max(., 0) (the `activation` of the same line);
 *   mask [T, N] bf16 or NULL: c is zeroed where mask <= 0 -- with mask = the saved ReLU output and b = W2^T this
 *   is `threshold_backward(dy @ W2, h, 0)`, the input gradient of `linear2(relu(.))` reaching `linear1`, in one pass.
 * float32 accumulation, one bf16 rounding at the end.  Returns 0, or an msda_status code (rlipv2_msda.h). */
int linear_expand_bf16(const void *a, const void *b, const void *bias, const void *mask, int T, int N, int K, int relu,
                       void *c, void *stream);

#ifdef __cplusplus
}
#endif
#endif
