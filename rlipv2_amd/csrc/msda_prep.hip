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
#include "msda_geometry.h"
#include "msda_internal.h"

namespace msda {

namespace {

constexpr int kL = 4, kP = 4, kLP = 16;
constexpr int kBlock = 256;

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
    float o[8], lg[4];
    geom::forward<QT, REFDIM>(qproj + r * (M * kLP * 3), ref + r * (kL * REFDIM), shapes, m, M, l, o, lg);
    geom::store_n<float, 4>(aw + pair * 16 + l * 4, lg);
    geom::store_n<float, 8>(loc + pair * 32 + l * 8, o);
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
    float a[4], ga[4], gl[8];
    geom::load_n<float, 4>(aw + pair * 16 + l * 4, a);
    geom::load_n<float, 4>(g_aw + pair * 16 + l * 4, ga);
    geom::load_n<float, 8>(g_loc + pair * 32 + l * 8, gl);
    geom::backward<QT, REFDIM>(grow, ref_row, shapes, m, M, l, a, ga, gl);
    // reference points: g_ref_xy = sum g_loc, g_ref_wh = sum g_loc * off * 0.5 / P
    if (g_ref) {
        float gr[4] = {0.f, 0.f, 0.f, 0.f};
        float off[8];
        if (REFDIM == 4) geom::load_n<QT, 8>(qproj + r * (M * kLP * 3) + m * 32 + l * 8, off);
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
