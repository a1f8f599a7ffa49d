/* rlipv2_alif.h -- C ABI of the fused bi-directional attention core of the ALIF language-image fusion (gfx950, bf16).
 *
 * Replaces, inside RLIPv2_BiMultiHeadAttention.forward (reference models/fuse_helper.py:365-466), the chain
 *   attn_weights = q k^T  ->  [clamps off]  ->  attn_weights_l = softmax_i(attn_weights^T - rowmax)  (:395-429)
 *   ->  attn_weights_v = softmax_j(attn_weights)  ->  dropout on both  ->  attn_probs_v @ value_l, attn_probs_l @ value_v
 * (~10 PyTorch launches on a 27-MFLOP problem per (image, head)) by ONE launch: a workgroup per (image, head), all
 * three products on v_mfma_f32_32x32x16_bf16, both softmaxes from one float32 copy of the logits in LDS.
 *
 *   q          [B, Tv, H*256] bf16   query projection of (vision + pos), ALREADY multiplied by head_dim^-0.5
 *   k          [B, Tl, H*256] bf16   key projection of the language tokens
 *   values_l_t [B, H*256, 64] bf16   value projection of the language tokens, TRANSPOSED (channel-major; columns
 *                                     >= Tl may hold anything finite)
 *   values_v_t [B, H*256, Tvp] bf16  value projection of the vision tokens, transposed, Tvp = alif_attention_padded_tv(Tv)
 *   keep_v / keep_l                   uint8 [B, H, Tv, Tl] / [B, H, Tl, Tv] dropout keep masks (both or neither);
 *                                     kept probabilities are multiplied by keep_scale = 1 / (1 - p)
 *   out_v [B, Tv, H*256], out_l [B, Tl, H*256] bf16
 *   probs_v [B, H, Tv, Tl], probs_l [B, H, Tl, Tv] bf16: the probabilities BEFORE dropout (for the backward pass)
 *
 * Supported: head_dim 256, Tl <= 64, Tv <= 288 (the `fusion_last_vis` configuration: Tv = 273 at 800x1333).  Nothing
 * allocates or synchronises; work is enqueued on `stream`.  Return value: 0 or an msda_status code (rlipv2_msda.h). */
#ifndef RLIPV2_ALIF_H
#define RLIPV2_ALIF_H

#ifdef __cplusplus
extern "C" {
#endif

int alif_attention_supported(int B, int H, int Tv, int Tl, int head_dim);
int alif_attention_padded_tv(int Tv);
int alif_attention_forward_bf16(const void *q, const void *k, const void *values_l_t, const void *values_v_t,
                                const void *keep_v, const void *keep_l, float keep_scale, int B, int H, int Tv, int Tl,
                                void *out_v, void *out_l, void *probs_v, void *probs_l, void *stream);

/* Backward of the two softmaxes + dropouts: gradient of the shared logits d_logits [B, H, Tv, Tl] (bf16) from the saved
 * probabilities and the gradients of the dropped probabilities d_probs_v [B, H, Tv, Tl] / d_probs_l [B, H, Tl, Tv]
 * (bf16); with keep masks also writes the dropped probabilities P * keep * keep_scale (the A operands of the
 * value-projection gradients).  One launch instead of ~12 elementwise / reduction launches. */
int alif_attention_softmax_backward_bf16(const void *probs_v, const void *probs_l, const void *d_probs_v,
                                         const void *d_probs_l, const void *keep_v, const void *keep_l, float keep_scale,
                                         int B, int H, int Tv, int Tl, void *d_logits, void *dropped_v, void *dropped_l,
                                         void *stream);

#ifdef __cplusplus
}
#endif
#endif
