/* rlipv2_decoder.h -- C ABI of the gradient-free glue between the layers of the DAB deformable decoders (gfx950).
 *
 * Between two decoder layers the reference (DABDeformableTransformerDecoderHOI.forward,
 * models/dab_deformable/deformable_transformer.py:1470-1552) refines the anchor boxes and rebuilds the layer's
 * reference points and sine position features from them -- all of it detached from autograd (:1525, :1541) and all of
 * it [N, nq, 4]-sized: ~25 launch-bound PyTorch kernels per layer.  Two launches here:
 *
 * dab_refine_boxes: out = sigmoid(delta + inverse_sigmoid(ref)), inverse_sigmoid(x) = log(max(clamp(x, 0, 1), eps) /
 *   max(1 - clamp(x, 0, 1), eps)) (util/misc.py inverse_sigmoid); delta [rows, 4] bf16 (delta_bf16 = 1) or float32,
 *   ref / out [rows, 4] float32.
 *
 * dab_reference_embed: boxes -> ref_in [N, nq, L, 4] float32 = box * (rx, ry, rx, ry) of every level's valid ratios, and
 *   the sine embedding of ref_in[:, :, 0, :] (gen_sineembed_for_position, :36-68): 128 sin / cos features per coordinate
 *   (sin on even, cos on odd frequencies, angle = coordinate * 2 pi / dim_t[j]), ordered (y, x, w, h) -> [N, nq, 512],
 *   float32 (embed_bf16 = 0) or bf16.  parse = 1: nq = 2 n, boxes = [sub_ref | obj_ref] (ParSe pair decoder);
 *   parse = 0: nq = n, box = 0.5 * (sub_ref + obj_ref) (verb decoder).  sub_ref / obj_ref [N, n, 4], valid_ratios
 *   [N, L, 2], dim_t [128] float32.  L <= 8.
 *
 * Float32 arithmetic in the reference's operation order.  Nothing allocates or synchronises; work is enqueued on `stream`.
 * Return value: 0 or an msda_status code (rlipv2_msda.h).
 */
#ifndef RLIPV2_DECODER_H
#define RLIPV2_DECODER_H

#ifdef __cplusplus
extern "C" {
#endif

int dab_refine_boxes(const void *delta, int delta_bf16, const float *ref, float *out, long rows, float eps, void *stream);

int dab_reference_embed(const float *sub_ref, const float *obj_ref, const float *valid_ratios, const float *dim_t,
                        int N, int n, int L, int parse, float *ref_in, void *embed, int embed_bf16, void *stream);

#ifdef __cplusplus
}
#endif
#endif
