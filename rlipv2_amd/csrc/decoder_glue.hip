// decoder_glue.hip -- the gradient-free box / reference-point glue between DAB decoder layers (include/rlipv2_decoder.h).
// Tiny tensors ([N, nq, 4]): the point is ONE launch instead of ~12 each; arithmetic is float32 in the reference's order.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rlipv2_decoder.h"
#include "../../include/rlipv2_msda.h"

namespace {

__device__ __forceinline__ uint16_t rne(float f)
{
    const uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

template <bool BF16>
__global__ __launch_bounds__(256) void refine_kernel(const void *__restrict__ delta, const float *__restrict__ ref,
                                                     float *__restrict__ out, long n, float eps)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float d = BF16 ? __uint_as_float((uint32_t) static_cast<const uint16_t *>(delta)[i] << 16)
                         : static_cast<const float *>(delta)[i];
    const float x = fminf(fmaxf(ref[i], 0.f), 1.f);
    const float inv = logf(fmaxf(x, eps) / fmaxf(1.f - x, eps));
    const float z = d + inv;
    out[i] = 1.f / (1.f + expf(-z));
}

template <bool BF16>
__global__ __launch_bounds__(256) void reference_kernel(const float *__restrict__ sub_ref, const float *__restrict__ obj_ref,
                                                        const float *__restrict__ ratios, const float *__restrict__ dim_t,
                                                        int N, int n, int L, int parse, float *__restrict__ ref_in,
                                                        void *__restrict__ embed)
{
    const int nq = parse ? 2 * n : n;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long row = idx >> 9;                       // 512 features per (image, query)
    if (row >= (long)N * nq) return;
    const int c = (int)(idx & 511);
    const int b = (int)(row / nq), q = (int)(row % nq);
    auto box = [&](int k) -> float {
        if (parse) return q < n ? sub_ref[((long)b * n + q) * 4 + k] : obj_ref[((long)b * n + q - n) * 4 + k];
        return 0.5f * (sub_ref[((long)b * n + q) * 4 + k] + obj_ref[((long)b * n + q) * 4 + k]);
    };
    if (c < 4 * L) {                                 // ref_in[b, q, l, k] = box[k] * (rx, ry, rx, ry)[l]
        const int l = c >> 2, k = c & 3;
        ref_in[(row * L + l) * 4 + k] = box(k) * ratios[((long)b * L + l) * 2 + (k & 1)];
    }
    // sine features of the level-0 reference, coordinate order (y, x, w, h)
    const int kk = c >> 7, j = c & 127;
    const int k = kk == 0 ? 1 : kk == 1 ? 0 : kk;
    const float pos = box(k) * ratios[(long)b * L * 2 + (k & 1)];
    const float ang = (pos * 6.283185307179586f) / dim_t[j];
    const float v = (j & 1) ? cosf(ang) : sinf(ang);
    if (BF16) static_cast<uint16_t *>(embed)[idx] = rne(v);
    else static_cast<float *>(embed)[idx] = v;
}

}  // namespace

extern "C" int dab_refine_boxes(const void *delta, int delta_bf16, const float *ref, float *out, long rows, float eps,
                                void *stream_)
{
    if (rows < 0) return MSDA_ERR_BAD_SHAPE;
    if (rows == 0) return MSDA_OK;
    if (!delta || !ref || !out) return MSDA_ERR_NULL_POINTER;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const long n = rows * 4;
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (delta_bf16) hipLaunchKernelGGL(refine_kernel<true>, grid, block, 0, stream, delta, ref, out, n, eps);
    else hipLaunchKernelGGL(refine_kernel<false>, grid, block, 0, stream, delta, ref, out, n, eps);
    return hipGetLastError() == hipSuccess ? MSDA_OK : MSDA_ERR_LAUNCH;
}

extern "C" int dab_reference_embed(const float *sub_ref, const float *obj_ref, const float *valid_ratios,
                                   const float *dim_t, int N, int n, int L, int parse, float *ref_in, void *embed,
                                   int embed_bf16, void *stream_)
{
    if (N < 0 || n < 0 || L < 1 || L > 8) return MSDA_ERR_BAD_SHAPE;
    if (N == 0 || n == 0) return MSDA_OK;
    if (!sub_ref || !obj_ref || !valid_ratios || !dim_t || !ref_in || !embed) return MSDA_ERR_NULL_POINTER;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const long total = (long)N * (parse ? 2 * n : n) * 512;
    const dim3 grid((unsigned)((total + 255) / 256)), block(256);
    if (embed_bf16)
        hipLaunchKernelGGL(reference_kernel<true>, grid, block, 0, stream, sub_ref, obj_ref, valid_ratios, dim_t, N, n, L,
                           parse, ref_in, embed);
    else
        hipLaunchKernelGGL(reference_kernel<false>, grid, block, 0, stream, sub_ref, obj_ref, valid_ratios, dim_t, N, n, L,
                           parse, ref_in, embed);
    return hipGetLastError() == hipSuccess ? MSDA_OK : MSDA_ERR_LAUNCH;
}
