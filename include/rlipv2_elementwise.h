/* rlipv2_elementwise.h -- C ABI of the element-wise fusions of the R50 trunk (gfx950, bf16).
 *
 * add_relu_bf16: y = relu(a + b) over n bfloat16 elements (n a multiple of 8, pointers 16-byte aligned; a, b, y any
 * memory format as long as all three share it) -- the bottleneck tail `relu(bn3(conv3(x)) + identity)` of the ResNet-50
 * trunk the reference takes from torchvision (models/DDETR_backbone.py:98-133), one pass instead of `add` + `relu`.
 * The sum is rounded to bfloat16 before the clamp, so the result is bit-identical to the two-op form.  Nothing allocates
 * or synchronises; work is enqueued on `stream`.  Return value: 0 or an msda_status code (rlipv2_msda.h). */
#ifndef RLIPV2_ELEMENTWISE_H
#define RLIPV2_ELEMENTWISE_H

#ifdef __cplusplus
extern "C" {
#endif

int add_relu_bf16(const void *a, const void *b, void *y, long n, void *stream);

/* affine_relu_bf16: y = relu(x * scale[c] + bias[c]) for a CHANNELS-LAST tensor of n elements with C channels (C a
 * multiple of 8; scale / bias bfloat16 [C], 16-byte aligned) -- frozen BatchNorm + ReLU behind a convolution of the trunk
 * (FrozenBatchNorm2d, models/DDETR_backbone.py:27-59, followed by relu) as one pass; product and sum rounded separately,
 * bit-identical to `addcmul` then `relu`.  affine_relu_backward_bf16: dx = dy * (y > 0) * scale[c]. */
int affine_relu_bf16(const void *x, const void *scale, const void *bias, void *y, long n, int C, void *stream);
int affine_relu_backward_bf16(const void *dy, const void *y, const void *scale, void *dx, long n, int C, void *stream);

#ifdef __cplusplus
}
#endif
#endif
