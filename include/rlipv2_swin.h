/* rlipv2_swin.h -- C ABI of the fused window attention of the Swin backbones (gfx950, bf16 data, float32 softmax;
 * csrc/window_attention.hip; round 5, not yet run on hardware).
 *
 * Replaces the attention core of WindowAttention.forward (reference models/swin/swin_transformer.py:262-301):
 *     attn = (q * scale) @ k^T + relative_position_bias[index]  (+ mask of the shifted window)   :275-290
 *     attn = softmax(attn); x = attn @ v                                                         :291-297
 * for one call over all windows and heads of a block.  attn_drop must be 0 (it is in every RLIPv2 Swin preset).
 *
 *   qkv      [windows, N, 3, heads, 32] bf16 -- the packed projection exactly as `self.qkv(x)` leaves it (:272-273)
 *   bias_t   [heads, 64, 64] float32: bias_t[h][i][j] = relative_position_bias[h][i][j] for query i, key j < N, padded to
 *            64 x 64: -30000 in the columns j >= N (padded keys never receive weight), 0 elsewhere
 *   mask_t   [K, 64, 64] float32, zero-padded like bias_t: the K distinct shift masks of the stage (:517-533);
 *            mask_id [windows_per_image] int32: index into mask_t, -1 = the window has no mask.  Both NULL: no masks
 *   out      [windows, N, heads * 32] bf16 -- the layout `(attn @ v).transpose(1, 2).reshape(B_, N, C)` produces (:297)
 *   d_out    like out;  d_qkv like qkv (every element written)
 * N = window_size^2 <= 64 (window_size <= 8), head dimension 32 (every preset).  Nothing N x N is saved between forward
 * and backward: the backward recomputes the probabilities from q and k.  No allocation, no synchronisation; work is
 * enqueued on `stream`.  Return value: 0 or an msda_status code (rlipv2_msda.h).
 */
#ifndef RLIPV2_SWIN_H
#define RLIPV2_SWIN_H

#ifdef __cplusplus
extern "C" {
#endif

int window_attention_supported(int windows, int heads, int tokens, int head_dim);

int window_attention_forward_bf16(const void *qkv, const float *bias_t, const float *mask_t, const int *mask_id, int windows,
                                  int windows_per_image, int heads, int tokens, float scale, void *out, void *stream);

int window_attention_backward_bf16(const void *qkv, const void *d_out, const float *bias_t, const float *mask_t,
                                   const int *mask_id, int windows, int windows_per_image, int heads, int tokens, float scale,
                                   void *d_qkv, void *stream);

/* The same two kernels with pad / cyclic shift / window partition / window reverse folded into their addressing
 * (models/swin/swin_transformer.py:362-396: F.pad, torch.roll, window_partition before the attention, window_reverse, roll back and
 * the crop after it -- five copies of the token map per block and direction as PyTorch ops):
 *   qkv, out, d_out, d_qkv stay in IMAGE order: [B, rows_per_image, 3 * heads * 32] / [B, rows_per_image, heads * 32]
 *   rowmap   [windows_per_image * N] int32: the row of token t of window w inside its image, or -(slot + 1) for a token that the
 *            reference creates by zero-padding the normalised map; rows_per_image + pads_per_image == windows_per_image * N
 *   pad_row  [3 * heads * 32] bf16: q / k / v of a padding token = the projection of a zero input = the qkv bias (NULL: zeros)
 *   d_pad    [B * pads_per_image, 3 * heads * 32] bf16: the gradient rows of the padding tokens (their sum over all rows is the
 *            padding's contribution to the bias gradient); may be NULL when there is no padding or no pad_row
 * Outputs of padding tokens are not written (the reference crops them), their d_out is zero. */
int window_attention_rows_forward_bf16(const void *qkv, const void *pad_row, const int *rowmap, int rows_per_image, int pads_per_image,
                                       const float *bias_t, const float *mask_t, const int *mask_id, int windows,
                                       int windows_per_image, int heads, int tokens, float scale, void *out, void *stream);

int window_attention_rows_backward_bf16(const void *qkv, const void *pad_row, const int *rowmap, int rows_per_image,
                                        int pads_per_image, const void *d_out, const float *bias_t, const float *mask_t,
                                        const int *mask_id, int windows, int windows_per_image, int heads, int tokens, float scale,
                                        void *d_qkv, void *d_pad, void *stream);

#ifdef __cplusplus
}
#endif
#endif
