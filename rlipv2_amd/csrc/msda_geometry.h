// msda_geometry.h -- the MSDeformAttn module's "sampling geometry" (reference
// models/ops/modules/ms_deform_attn.py:101-112: view + softmax over the L*P logits + offsets / (W, H)
// [2-d reference points] or offsets / P * wh * 0.5 [4-d reference boxes] + reference point) as device
// functions for ONE quad lane = one level of one (row, head), L = P = 4.
//
// Shared by the stand-alone kernels of msda_prep.hip and by the fused MSDA kernels of msda_quad.hip (which
// run it as their prologue / epilogue), so both routes compute sampling_loc / attn_weight and their
// gradients with the same instructions in the same order: bit-identical results.
#pragma once

#include "msda_device.h"

namespace msda {
namespace geom {

constexpr int kL = 4, kP = 4, kLP = 16;

// n consecutive elements of the projection row as floats (n = 4 or 8)
template <typename QT, int NV> __device__ __forceinline__ void load_n(const QT *p, float (&v)[NV]);
template <> __device__ __forceinline__ void load_n<float, 4>(const float *p, float (&v)[4])
{
    const float4 a = *reinterpret_cast<const float4 *>(p);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
}
template <> __device__ __forceinline__ void load_n<float, 8>(const float *p, float (&v)[8])
{
    const float4 a = reinterpret_cast<const float4 *>(p)[0], b = reinterpret_cast<const float4 *>(p)[1];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <> __device__ __forceinline__ void load_n<bf16_t, 4>(const bf16_t *p, float (&v)[4])
{
    const uint2 a = *reinterpret_cast<const uint2 *>(p);
    v[0] = bf16_lo(a.x); v[1] = bf16_hi(a.x); v[2] = bf16_lo(a.y); v[3] = bf16_hi(a.y);
}
template <> __device__ __forceinline__ void load_n<bf16_t, 8>(const bf16_t *p, float (&v)[8])
{
    const uint4 a = *reinterpret_cast<const uint4 *>(p);
    v[0] = bf16_lo(a.x); v[1] = bf16_hi(a.x); v[2] = bf16_lo(a.y); v[3] = bf16_hi(a.y);
    v[4] = bf16_lo(a.z); v[5] = bf16_hi(a.z); v[6] = bf16_lo(a.w); v[7] = bf16_hi(a.w);
}
template <typename QT, int NV> __device__ __forceinline__ void store_n(QT *p, const float (&v)[NV]);
template <> __device__ __forceinline__ void store_n<float, 4>(float *p, const float (&v)[4])
{
    *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <> __device__ __forceinline__ void store_n<float, 8>(float *p, const float (&v)[8])
{
    reinterpret_cast<float4 *>(p)[0] = make_float4(v[0], v[1], v[2], v[3]);
    reinterpret_cast<float4 *>(p)[1] = make_float4(v[4], v[5], v[6], v[7]);
}
template <> __device__ __forceinline__ void store_n<bf16_t, 4>(bf16_t *p, const float (&v)[4])
{
    *reinterpret_cast<uint2 *>(p) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
}
template <> __device__ __forceinline__ void store_n<bf16_t, 8>(bf16_t *p, const float (&v)[8])
{
    *reinterpret_cast<uint4 *>(p) = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]),
                                               pack_bf16x2(v[6], v[7]));
}

__device__ __forceinline__ float quad_max(float v)
{
    v = fmaxf(v, dpp_quad<MSDA_QUAD_PERM(1, 0, 3, 2)>(v));
    return fmaxf(v, dpp_quad<MSDA_QUAD_PERM(2, 3, 0, 1)>(v));
}

// per-level (sx, sy) multiplying an offset: 1/(W, H) or wh * 0.5 / P
template <int REFDIM>
__device__ __forceinline__ void level_scale(const float *ref_row, const int64_t *shapes, int l, float &sx, float &sy)
{
    if (REFDIM == 2) {
        sx = 1.f / (float)shapes[2 * l + 1];
        sy = 1.f / (float)shapes[2 * l];
    } else {
        sx = ref_row[l * 4 + 2] * (0.5f / kP);
        sy = ref_row[l * 4 + 3] * (0.5f / kP);
    }
}

// Forward, quad lane l of (row, head m): the level's 4 attention weights `w` and 4 sampling locations
// `o` = (x0, y0, ..., x3, y3).  All four lanes of the quad must call it together (quad shuffles).
template <typename QT, int REFDIM>
__device__ __forceinline__ void forward(const QT *__restrict__ row, const float *__restrict__ ref_row,
                                        const int64_t *__restrict__ shapes, int m, int M, int l, float (&o)[8],
                                        float (&w)[4])
{
    float off[8];
    load_n<QT, 8>(row + m * 32 + l * 8, off);
    load_n<QT, 4>(row + M * 32 + m * 16 + l * 4, w);
    // softmax over the 16 samples of the head
    const float mx = quad_max(fmaxf(fmaxf(w[0], w[1]), fmaxf(w[2], w[3])));
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { w[i] = __expf(w[i] - mx); sum += w[i]; }
    const float inv = 1.f / quad_sum(sum);
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] *= inv;
    float sx, sy;
    level_scale<REFDIM>(ref_row, shapes, l, sx, sy);
    const float rx = ref_row[l * REFDIM], ry = ref_row[l * REFDIM + 1];
#pragma unroll
    for (int pnt = 0; pnt < 4; ++pnt) {
        o[2 * pnt] = fmaf(off[2 * pnt], sx, rx);
        o[2 * pnt + 1] = fmaf(off[2 * pnt + 1], sy, ry);
    }
}

// bfloat16 rows through the hardware's packed conversion (v_cvt_pk_bf16_f32: round to nearest even, the same bits as
// float_to_bf16_bits for every finite value at a fifth of the instructions).  Only the kernels written after round 3 ask for
// it (HW_CVT): the device code of the hardware-validated ones stays what it was.
template <int NV> __device__ __forceinline__ void store_n_hw(bf16_t *p, const float (&v)[NV])
{
    typedef __attribute__((ext_vector_type(2))) float f32x2_;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_;
    uint32_t w[NV / 2];
#pragma unroll
    for (int i = 0; i < NV / 2; ++i) {
        const f32x2_ f = {v[2 * i], v[2 * i + 1]};
        w[i] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2_));
    }
    if (NV == 4) *reinterpret_cast<uint2 *>(p) = make_uint2(w[0], w[1]);
    else *reinterpret_cast<uint4 *>(p) = make_uint4(w[0], w[1], w[NV == 8 ? 2 : 0], w[NV == 8 ? 3 : 1]);
}

// Backward, quad lane l of (row, head m): a = the level's attention weights, ga = their gradients (overwritten by
// the gradients of the logits), gl = gradients of the locations; writes the level's 8 + 4 entries of the
// projection row's gradient.  Returns the location scale in (sx, sy) for the reference-point gradient.
template <typename QT, int REFDIM, bool HW_CVT = false>
__device__ __forceinline__ void backward(QT *__restrict__ grow, const float *__restrict__ ref_row,
                                         const int64_t *__restrict__ shapes, int m, int M, int l, const float (&a)[4],
                                         float (&ga)[4], const float (&gl)[8])
{
    // softmax backward: g_logit = aw * (g_aw - sum_j aw_j g_aw_j)
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) dot = fmaf(a[i], ga[i], dot);
    dot = quad_sum(dot);
#pragma unroll
    for (int i = 0; i < 4; ++i) ga[i] = a[i] * (ga[i] - dot);
    if constexpr (HW_CVT) store_n_hw<4>(grow + M * 32 + m * 16 + l * 4, ga);
    else store_n<QT, 4>(grow + M * 32 + m * 16 + l * 4, ga);
    // offsets: g_off = g_loc * scale
    float o[8], sx, sy;
    level_scale<REFDIM>(ref_row, shapes, l, sx, sy);
#pragma unroll
    for (int pnt = 0; pnt < 4; ++pnt) {
        o[2 * pnt] = gl[2 * pnt] * sx;
        o[2 * pnt + 1] = gl[2 * pnt + 1] * sy;
    }
    if constexpr (HW_CVT) store_n_hw<8>(grow + m * 32 + l * 8, o);
    else store_n<QT, 8>(grow + m * 32 + l * 8, o);
}

}  // namespace geom
}  // namespace msda
