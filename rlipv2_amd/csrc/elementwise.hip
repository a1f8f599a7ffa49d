// elementwise.hip -- the few element-wise fusions of the R50 trunk that no library epilogue covers (gfx950, bf16).
//
//   add_relu:  y = relu(a + b)   -- the tail of every ResNet bottleneck, `relu(bn3(conv3(.)) + identity)` (reference trunk:
//              torchvision's Bottleneck.forward behind models/DDETR_backbone.py:98-133).  As `add` + `relu` it is five
//              passes over the block's output tensor (107 MB per block in layer1 at batch 4); here three.  The backward
//              needs only y: dy * (y > 0), one `threshold_backward`, shared by both addends.
// 16 bytes per lane, grid-stride; nothing allocates or synchronises.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rlipv2_elementwise.h"
#include "../../include/rlipv2_msda.h"

namespace {

__device__ __forceinline__ float lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ uint32_t rne(float f)
{
    const uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
// relu(a + b) of two packed bf16 pairs; the sum is rounded to bf16 BEFORE the clamp, as `add` then `relu` would
__device__ __forceinline__ uint32_t add_relu2(uint32_t a, uint32_t b)
{
    const uint32_t l = rne(lo(a) + lo(b)), h = rne(hi(a) + hi(b));
    return ((l & 0x8000u) ? 0u : l) | (((h & 0x8000u) ? 0u : h) << 16);
}

__global__ __launch_bounds__(256) void add_relu_kernel(const uint4 *__restrict__ a, const uint4 *__restrict__ b,
                                                       uint4 *__restrict__ y, long n16)
{
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) {
        const uint4 x = a[i], z = b[i];
        y[i] = make_uint4(add_relu2(x.x, z.x), add_relu2(x.y, z.y), add_relu2(x.z, z.z), add_relu2(x.w, z.w));
    }
}

// y = relu(x * scale[c] + bias[c]) for a channels-last tensor (channel = element index mod C, C a multiple of 8): frozen
// BatchNorm + ReLU behind the trunk's 3x3 / 7x7 convolutions (`relu(bn(conv(x)))`, two passes as addcmul + relu).
// Product and sum are rounded separately in float32 (no fused multiply-add), the result to bf16 before the clamp: the
// same bits as the two-op form.
__global__ __launch_bounds__(256) void affine_relu_kernel(const uint4 *__restrict__ x, const uint4 *__restrict__ scale,
                                                          const uint4 *__restrict__ bias, uint4 *__restrict__ y, long n16,
                                                          int c16)
{
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) {
        const uint4 v = x[i], sc = scale[i % c16], bs = bias[i % c16];
        auto f = [](uint32_t xv, uint32_t sv, uint32_t bv) {
            const uint32_t l = rne(__fadd_rn(__fmul_rn(lo(xv), lo(sv)), lo(bv)));
            const uint32_t h = rne(__fadd_rn(__fmul_rn(hi(xv), hi(sv)), hi(bv)));
            return ((l & 0x8000u) ? 0u : l) | (((h & 0x8000u) ? 0u : h) << 16);
        };
        y[i] = make_uint4(f(v.x, sc.x, bs.x), f(v.y, sc.y, bs.y), f(v.z, sc.z, bs.z), f(v.w, sc.w, bs.w));
    }
}

// dx = dy * (y > 0) * scale[c]   (threshold_backward followed by the addcmul backward, one pass)
__global__ __launch_bounds__(256) void affine_relu_backward_kernel(const uint4 *__restrict__ dy, const uint4 *__restrict__ y,
                                                                   const uint4 *__restrict__ scale, uint4 *__restrict__ dx,
                                                                   long n16, int c16)
{
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) {
        const uint4 g = dy[i], o = y[i], sc = scale[i % c16];
        auto f = [](uint32_t gv, uint32_t ov, uint32_t sv) {
            const uint32_t l = (ov & 0xffffu) && !(ov & 0x8000u) ? rne(lo(gv) * lo(sv)) : 0u;
            const uint32_t h = (ov >> 16) && !(ov & 0x80000000u) ? rne(hi(gv) * hi(sv)) : 0u;
            return l | (h << 16);
        };
        dx[i] = make_uint4(f(g.x, o.x, sc.x), f(g.y, o.y, sc.y), f(g.z, o.z, sc.z), f(g.w, o.w, sc.w));
    }
}

unsigned grid_for(long n16) { const long b = (n16 + 255) / 256; return (unsigned)(b < 256 * 16 ? b : 256 * 16); }

}  // namespace

extern "C" int affine_relu_bf16(const void *x, const void *scale, const void *bias, void *y, long n, int C, void *stream)
{
    if (n < 0 || C <= 0 || (C & 7) || n % C) return MSDA_ERR_BAD_SHAPE;
    if (n == 0) return MSDA_OK;
    if (!x || !scale || !bias || !y) return MSDA_ERR_NULL_POINTER;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(scale) | reinterpret_cast<uintptr_t>(bias) |
         reinterpret_cast<uintptr_t>(y)) & 15u)
        return MSDA_ERR_ALIGNMENT;
    (void)hipGetLastError();
    hipLaunchKernelGGL(affine_relu_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (const uint4 *)x,
                       (const uint4 *)scale, (const uint4 *)bias, (uint4 *)y, n / 8, C / 8);
    return hipGetLastError() == hipSuccess ? MSDA_OK : MSDA_ERR_LAUNCH;
}

extern "C" int affine_relu_backward_bf16(const void *dy, const void *y, const void *scale, void *dx, long n, int C,
                                         void *stream)
{
    if (n < 0 || C <= 0 || (C & 7) || n % C) return MSDA_ERR_BAD_SHAPE;
    if (n == 0) return MSDA_OK;
    if (!dy || !y || !scale || !dx) return MSDA_ERR_NULL_POINTER;
    if ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(scale) |
         reinterpret_cast<uintptr_t>(dx)) & 15u)
        return MSDA_ERR_ALIGNMENT;
    (void)hipGetLastError();
    hipLaunchKernelGGL(affine_relu_backward_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream,
                       (const uint4 *)dy, (const uint4 *)y, (const uint4 *)scale, (uint4 *)dx, n / 8, C / 8);
    return hipGetLastError() == hipSuccess ? MSDA_OK : MSDA_ERR_LAUNCH;
}

extern "C" int add_relu_bf16(const void *a, const void *b, void *y, long n, void *stream)
{
    if (n < 0 || (n & 7)) return MSDA_ERR_BAD_SHAPE;
    if (n == 0) return MSDA_OK;
    if (!a || !b || !y) return MSDA_ERR_NULL_POINTER;
    if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(y)) & 15u)
        return MSDA_ERR_ALIGNMENT;
    const long n16 = n / 8;
    const long blocks = (n16 + 255) / 256;
    (void)hipGetLastError();
    hipLaunchKernelGGL(add_relu_kernel, dim3((unsigned)(blocks < 256 * 16 ? blocks : 256 * 16)), dim3(256), 0,
                       (hipStream_t)stream, (const uint4 *)a, (const uint4 *)b, (uint4 *)y, n16);
    return hipGetLastError() == hipSuccess ? MSDA_OK : MSDA_ERR_LAUNCH;
}
