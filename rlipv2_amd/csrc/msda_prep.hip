// msda_prep.hip -- fused "sampling geometry" of the MSDeformAttn module for gfx950 (L = 4, P = 4).
//
// The reference's module turns the query projection into the op's operands with a chain of
// PyTorch ops (models/ops/modules/ms_deform_attn.py:101-112): view, softmax over the L*P logits,
// offsets / (W, H) [2-d reference points] or offsets / P * wh * 0.5 [4-d reference boxes], add the
// reference point.  At the encoder shape each of those is a pass over a 45-91 MB tensor, forward and
// backward.  Here one kernel reads the projection row once and writes sampling_loc and
// attn_weight (float32, the layouts the op expects); its backward reads their gradients once and
// writes the gradient of the projection row (and, when asked, of the reference points).
//
//   qproj  [R, M*L*P*3]   f32 or bf16: M*L*P*2 offsets (m, l, p, xy) then M*L*P logits (m, l, p)
//   ref    [R, L, 2|4]    f32, normalised (x, y[, w, h])
//   loc    [R, M, L, P, 2], aw [R, M, L, P]   f32            (R = N * Lq)
// thread = (row, head, level): see the kernels.
#include "msda_device.h"
#include "msda_internal.h"

namespace msda {

namespace {

constexpr int kL = 4, kP = 4, kLP = 16;
constexpr int kBlock = 256;

template <typename QT> __device__ __forceinline__ void load16(const QT *p, float (&v)[16]);
template <> __device__ __forceinline__ void load16<float>(const float *p, float (&v)[16])
{
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float4 a = reinterpret_cast<const float4 *>(p)[i];
        v[4 * i] = a.x; v[4 * i + 1] = a.y; v[4 * i + 2] = a.z; v[4 * i + 3] = a.w;
    }
}
template <> __device__ __forceinline__ void load16<bf16_t>(const bf16_t *p, float (&v)[16])
{
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const uint4 a = reinterpret_cast<const uint4 *>(p)[i];
        v[8 * i] = bf16_lo(a.x); v[8 * i + 1] = bf16_hi(a.x); v[8 * i + 2] = bf16_lo(a.y); v[8 * i + 3] = bf16_hi(a.y);
        v[8 * i + 4] = bf16_lo(a.z); v[8 * i + 5] = bf16_hi(a.z); v[8 * i + 6] = bf16_lo(a.w); v[8 * i + 7] = bf16_hi(a.w);
    }
}
template <typename QT> __device__ __forceinline__ void store16(QT *p, const float (&v)[16]);
template <> __device__ __forceinline__ void store16<float>(float *p, const float (&v)[16])
{
#pragma unroll
    for (int i = 0; i < 4; ++i)
        reinterpret_cast<float4 *>(p)[i] = make_float4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
}
template <> __device__ __forceinline__ void store16<bf16_t>(bf16_t *p, const float (&v)[16])
{
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        uint4 a;
        a.x = pack_bf16x2(v[8 * i], v[8 * i + 1]); a.y = pack_bf16x2(v[8 * i + 2], v[8 * i + 3]);
        a.z = pack_bf16x2(v[8 * i + 4], v[8 * i + 5]); a.w = pack_bf16x2(v[8 * i + 6], v[8 * i + 7]);
        reinterpret_cast<uint4 *>(p)[i] = a;
    }
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
    v = fmaxf(v, __shfl_xor(v, 1, 64));
    return fmaxf(v, __shfl_xor(v, 2, 64));
}
__device__ __forceinline__ float quad_sum(float v)
{
    v += __shfl_xor(v, 1, 64);
    return v + __shfl_xor(v, 2, 64);
}

// thread = (row, head, level): the four lanes of a (row, head) sit in one DPP quad, the softmax over the 16 samples is
// 4 values per lane + two quad shuffles.  Every load / store instruction of a wave then covers one contiguous
// kilobyte (16 bytes per lane) or two interleaved ones (the 32-byte location / offset pieces) -- with a thread per
// (row, head) and 64-128 bytes per thread every store instruction touched 64 different lines (54 / 61 us per encoder
// call against 25 us of HBM time).
template <typename QT, int REFDIM>
__global__ __launch_bounds__(kBlock) void prep_forward_kernel(const QT *__restrict__ qproj, const float *__restrict__ ref,
                                                              const int64_t *__restrict__ shapes, int R, int M,
                                                              float *__restrict__ loc, float *__restrict__ aw)
{
    const long t = (long)blockIdx.x * kBlock + threadIdx.x;
    const long pair = t >> 2;                   // (row, head)
    const int l = (int)(t & 3);                 // level
    if (pair >= (long)R * M) return;            // (whole quads leave together)
    const long r = pair / M;
    const int m = (int)(pair % M);
    const QT *row = qproj + r * (M * kLP * 3);
    const float *ref_row = ref + r * (kL * REFDIM);
    float off[8], lg[4];
    load_n<QT, 8>(row + m * 32 + l * 8, off);
    load_n<QT, 4>(row + M * 32 + m * 16 + l * 4, lg);
    // softmax over the 16 samples of the head
    const float mx = quad_max(fmaxf(fmaxf(lg[0], lg[1]), fmaxf(lg[2], lg[3])));
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { lg[i] = __expf(lg[i] - mx); sum += lg[i]; }
    const float inv = 1.f / quad_sum(sum);
#pragma unroll
    for (int i = 0; i < 4; ++i) lg[i] *= inv;
    store_n<float, 4>(aw + pair * 16 + l * 4, lg);
    float sx, sy;
    level_scale<REFDIM>(ref_row, shapes, l, sx, sy);
    const float rx = ref_row[l * REFDIM], ry = ref_row[l * REFDIM + 1];
    float o[8];
#pragma unroll
    for (int pnt = 0; pnt < 4; ++pnt) {
        o[2 * pnt] = fmaf(off[2 * pnt], sx, rx);
        o[2 * pnt + 1] = fmaf(off[2 * pnt + 1], sy, ry);
    }
    store_n<float, 8>(loc + pair * 32 + l * 8, o);
}

template <typename QT, int REFDIM>
__global__ __launch_bounds__(kBlock) void prep_backward_kernel(const QT *__restrict__ qproj, const float *__restrict__ ref,
                                                               const int64_t *__restrict__ shapes,
                                                               const float *__restrict__ aw,
                                                               const float *__restrict__ g_loc,
                                                               const float *__restrict__ g_aw, int R, int M,
                                                               QT *__restrict__ g_qproj, float *__restrict__ g_ref)
{
    const long t = (long)blockIdx.x * kBlock + threadIdx.x;
    const long pair = t >> 2;
    const int l = (int)(t & 3);
    if (pair >= (long)R * M) return;
    const long r = pair / M;
    const int m = (int)(pair % M);
    const float *ref_row = ref + r * (kL * REFDIM);
    QT *grow = g_qproj + r * (M * kLP * 3);
    // softmax backward: g_logit = aw * (g_aw - sum_j aw_j g_aw_j)
    float a[4], ga[4];
    load_n<float, 4>(aw + pair * 16 + l * 4, a);
    load_n<float, 4>(g_aw + pair * 16 + l * 4, ga);
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) dot = fmaf(a[i], ga[i], dot);
    dot = quad_sum(dot);
#pragma unroll
    for (int i = 0; i < 4; ++i) ga[i] = a[i] * (ga[i] - dot);
    store_n<QT, 4>(grow + M * 32 + m * 16 + l * 4, ga);
    // offsets: g_off = g_loc * scale; reference points: g_ref_xy = sum g_loc, g_ref_wh = sum g_loc * off * 0.5 / P
    float gl[8], o[8], sx, sy;
    load_n<float, 8>(g_loc + pair * 32 + l * 8, gl);
    level_scale<REFDIM>(ref_row, shapes, l, sx, sy);
#pragma unroll
    for (int pnt = 0; pnt < 4; ++pnt) {
        o[2 * pnt] = gl[2 * pnt] * sx;
        o[2 * pnt + 1] = gl[2 * pnt + 1] * sy;
    }
    store_n<QT, 8>(grow + m * 32 + l * 8, o);
    if (g_ref) {
        float gr[4] = {0.f, 0.f, 0.f, 0.f};
        float off[8];
        if (REFDIM == 4) load_n<QT, 8>(qproj + r * (M * kLP * 3) + m * 32 + l * 8, off);
#pragma unroll
        for (int pnt = 0; pnt < 4; ++pnt) {
            gr[0] += gl[2 * pnt];
            gr[1] += gl[2 * pnt + 1];
            if (REFDIM == 4) {
                gr[2] += gl[2 * pnt] * off[2 * pnt] * (0.5f / kP);
                gr[3] += gl[2 * pnt + 1] * off[2 * pnt + 1] * (0.5f / kP);
            }
        }
#pragma unroll
        for (int k = 0; k < REFDIM; ++k) atomic_add(g_ref + r * (kL * REFDIM) + l * REFDIM + k, gr[k]);
    }
}

}  // namespace

}  // namespace msda

using namespace msda;

extern "C" {

int msda_prepare_forward(int qdtype, const void *qproj, const float *ref, int refdim, const int64_t *shapes, int R,
                         int M, int L, int P, float *loc, float *aw, void *stream)
{
    if (qdtype != MSDA_F32 && qdtype != MSDA_BF16) return MSDA_ERR_BAD_DTYPE;
    if (L != kL || P != kP || (refdim != 2 && refdim != 4) || R < 0 || M < 0) return MSDA_ERR_BAD_SHAPE;
    if ((long)R * M == 0) return MSDA_OK;
    if (!qproj || !ref || !shapes || !loc || !aw) return MSDA_ERR_NULL_POINTER;
    const int grid = (int)(((long)R * M * 4 + kBlock - 1) / kBlock);
    hipStream_t s = (hipStream_t)stream;
    (void)hipGetLastError();
#define MSDA_PREP_FWD(QT, RD)                                                                                   \
    hipLaunchKernelGGL((prep_forward_kernel<QT, RD>), dim3(grid), dim3(kBlock), 0, s, (const QT *)qproj, ref, \
                       shapes, R, M, loc, aw)
    if (qdtype == MSDA_F32) { if (refdim == 2) MSDA_PREP_FWD(float, 2); else MSDA_PREP_FWD(float, 4); }
    else { if (refdim == 2) MSDA_PREP_FWD(bf16_t, 2); else MSDA_PREP_FWD(bf16_t, 4); }
#undef MSDA_PREP_FWD
    return hipGetLastError() == hipSuccess ? MSDA_OK : MSDA_ERR_LAUNCH;
}

int msda_prepare_backward(int qdtype, const void *qproj, const float *ref, int refdim, const int64_t *shapes,
                          const float *aw, const float *g_loc, const float *g_aw, int R, int M, int L, int P,
                          void *g_qproj, float *g_ref, void *stream)
{
    if (qdtype != MSDA_F32 && qdtype != MSDA_BF16) return MSDA_ERR_BAD_DTYPE;
    if (L != kL || P != kP || (refdim != 2 && refdim != 4) || R < 0 || M < 0) return MSDA_ERR_BAD_SHAPE;
    if ((long)R * M == 0) return MSDA_OK;
    if (!qproj || !ref || !shapes || !aw || !g_loc || !g_aw || !g_qproj) return MSDA_ERR_NULL_POINTER;
    const int grid = (int)(((long)R * M * 4 + kBlock - 1) / kBlock);
    hipStream_t s = (hipStream_t)stream;
    (void)hipGetLastError();
    if (g_ref && hipMemsetAsync(g_ref, 0, (size_t)R * kL * refdim * sizeof(float), s) != hipSuccess) return MSDA_ERR_LAUNCH;
#define MSDA_PREP_BWD(QT, RD)                                                                                    \
    hipLaunchKernelGGL((prep_backward_kernel<QT, RD>), dim3(grid), dim3(kBlock), 0, s, (const QT *)qproj, ref, \
                       shapes, aw, g_loc, g_aw, R, M, (QT *)g_qproj, g_ref)
    if (qdtype == MSDA_F32) { if (refdim == 2) MSDA_PREP_BWD(float, 2); else MSDA_PREP_BWD(float, 4); }
    else { if (refdim == 2) MSDA_PREP_BWD(bf16_t, 2); else MSDA_PREP_BWD(bf16_t, 4); }
#undef MSDA_PREP_BWD
    return hipGetLastError() == hipSuccess ? MSDA_OK : MSDA_ERR_LAUNCH;
}

}  // extern "C"
