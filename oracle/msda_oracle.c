/*
 * oracle/msda_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference's multi-scale deformable attention
 * (forward + backward), used only as the checker in tests/, in
 * __graft_entry__.smoke() and as bench.py's `cpu_baseline` leg.  The product
 * path (rlipv2_amd/) never imports, links or calls anything in this directory.
 *
 * What it follows (paths relative to the reference checkout):
 *   - math of the op:            models/ops/functions/ms_deform_attn_func.py:45-65
 *                                 (per-level bilinear grid_sample, zeros padding,
 *                                 align_corners=False, weighted sum over L*P samples)
 *   - pixel convention / bounds: models/ops/src/cuda/ms_deform_im2col_cuda.cuh:282-291
 *                                 (h_im = loc_y*H - 0.5, w_im = loc_x*W - 0.5, a sample
 *                                 contributes iff h_im>-1 && w_im>-1 && h_im<H && w_im<W)
 *   - bilinear fetch, 4 guarded corners, weights hh*hw, hh*lw, lh*hw, lh*lw:
 *                                 ms_deform_im2col_cuda.cuh:33-84
 *   - backward formulas (grad_value scatter, grad_attn_weight, grad_sampling_loc):
 *                                 ms_deform_im2col_cuda.cuh:87-159
 *   - output zero-initialised, layout [N, Lq, M*D]:
 *                                 models/ops/src/cuda/ms_deform_attn_cuda.cu:54,77,121-123
 *
 * The reference's native code for this path is CUDA-only (its CPU entry points
 * raise, models/ops/src/cpu/ms_deform_attn_cpu.cpp:24,40) so it cannot be built
 * here; this restatement is pinned instead against vectors produced by importing
 * the reference's own pure-PyTorch `ms_deform_attn_core_pytorch` + autograd in
 * the build container (tests/golden/make_msda_golden.py -> tests/golden/ npz files).
 *
 * Operation order is kept as the reference states it (no FMA contraction:
 * build with -ffp-contract=off) so that float results are reproducible.
 *
 * Layouts (all contiguous, row-major):
 *   value   [N, S, M, D]          S = sum_l H_l*W_l, level-major then row-major (h, w)
 *   shapes  int64 [L, 2] = (H, W) starts int64 [L]
 *   loc     [N, Lq, M, L, P, 2]   (x, y) normalised to the level extent
 *   aw      [N, Lq, M, L, P]
 *   out / grad_out [N, Lq, M*D]
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define MSDA_ORACLE_DEFINE(T, SUFFIX)                                                          \
                                                                                               \
/* ms_deform_im2col_cuda.cuh:33-84 (one channel vector of D entries instead of one channel) */ \
static void bilinear_fwd_##SUFFIX(const T *lvl, int H, int W, int M, int D, T h, T w, int m,   \
                                  T weight, T *acc)                                            \
{                                                                                              \
    const int h_low = (int)floor((double)h), w_low = (int)floor((double)w);                    \
    const int h_high = h_low + 1, w_high = w_low + 1;                                          \
    const T lh = h - (T)h_low, lw = w - (T)w_low;                                              \
    const T hh = (T)1 - lh, hw = (T)1 - lw;                                                    \
    const long w_stride = (long)M * D, h_stride = (long)W * w_stride;                          \
    const T w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;                            \
    const T *p1 = (h_low >= 0 && w_low >= 0)                                                   \
                      ? lvl + h_low * h_stride + w_low * w_stride + (long)m * D : 0;           \
    const T *p2 = (h_low >= 0 && w_high <= W - 1)                                              \
                      ? lvl + h_low * h_stride + w_high * w_stride + (long)m * D : 0;          \
    const T *p3 = (h_high <= H - 1 && w_low >= 0)                                              \
                      ? lvl + h_high * h_stride + w_low * w_stride + (long)m * D : 0;          \
    const T *p4 = (h_high <= H - 1 && w_high <= W - 1)                                         \
                      ? lvl + h_high * h_stride + w_high * w_stride + (long)m * D : 0;         \
    for (int c = 0; c < D; ++c) {                                                              \
        const T v1 = p1 ? p1[c] : (T)0, v2 = p2 ? p2[c] : (T)0;                                \
        const T v3 = p3 ? p3[c] : (T)0, v4 = p4 ? p4[c] : (T)0;                                \
        const T val = (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);                                 \
        acc[c] += val * weight; /* .cuh:290 */                                                 \
    }                                                                                          \
}                                                                                              \
                                                                                               \
void msda_oracle_forward_##SUFFIX(const T *value, const int64_t *shapes,                       \
                                  const int64_t *starts, const T *loc, const T *aw, int N,     \
                                  int S, int M, int D, int L, int Lq, int P, T *out)           \
{                                                                                              \
    const long qm_stride = (long)M * D;                                                        \
    _Pragma("omp parallel for collapse(2) schedule(static)")                                   \
    for (int n = 0; n < N; ++n)                                                                \
        for (int q = 0; q < Lq; ++q)                                                           \
            for (int m = 0; m < M; ++m) {                                                      \
                const long qm = ((long)n * Lq + q) * M + m;                                    \
                T *o = out + qm * D;                                                           \
                for (int c = 0; c < D; ++c) o[c] = (T)0; /* at::zeros, .cu:54 */               \
                const T *lp = loc + qm * L * P * 2;                                            \
                const T *wp = aw + qm * L * P;                                                 \
                for (int l = 0; l < L; ++l) {                                                  \
                    const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];              \
                    const T *lvl = value + ((long)n * S + starts[l]) * qm_stride;              \
                    for (int p = 0; p < P; ++p) {                                              \
                        const T loc_w = lp[(l * P + p) * 2], loc_h = lp[(l * P + p) * 2 + 1];  \
                        const T weight = wp[l * P + p];                                        \
                        const T h_im = loc_h * (T)H - (T)0.5;                                  \
                        const T w_im = loc_w * (T)W - (T)0.5;                                  \
                        if (h_im > -1 && w_im > -1 && h_im < H && w_im < W)                    \
                            bilinear_fwd_##SUFFIX(lvl, H, W, M, D, h_im, w_im, m, weight, o);  \
                    }                                                                          \
                }                                                                              \
            }                                                                                  \
}                                                                                              \
                                                                                               \
/* ms_deform_im2col_cuda.cuh:87-159; the reduction over the D channels that the reference   */ \
/* does through shared memory (.cuh:376-394) is the plain sum below.                         */ \
static void bilinear_bwd_##SUFFIX(const T *lvl, T *glvl, int H, int W, int M, int D, T h,      \
                                  T w, int m, const T *top_grad, T attn, T *g_loc, T *g_aw)    \
{                                                                                              \
    const int h_low = (int)floor((double)h), w_low = (int)floor((double)w);                    \
    const int h_high = h_low + 1, w_high = w_low + 1;                                          \
    const T lh = h - (T)h_low, lw = w - (T)w_low;                                              \
    const T hh = (T)1 - lh, hw = (T)1 - lw;                                                    \
    const long w_stride = (long)M * D, h_stride = (long)W * w_stride;                          \
    const T w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;                            \
    const int ok1 = (h_low >= 0 && w_low >= 0), ok2 = (h_low >= 0 && w_high <= W - 1);         \
    const int ok3 = (h_high <= H - 1 && w_low >= 0);                                           \
    const int ok4 = (h_high <= H - 1 && w_high <= W - 1);                                      \
    const long o1 = h_low * h_stride + w_low * w_stride + (long)m * D;                         \
    const long o2 = h_low * h_stride + w_high * w_stride + (long)m * D;                        \
    const long o3 = h_high * h_stride + w_low * w_stride + (long)m * D;                        \
    const long o4 = h_high * h_stride + w_high * w_stride + (long)m * D;                       \
    T sum_aw = 0, sum_w = 0, sum_h = 0;                                                        \
    for (int c = 0; c < D; ++c) {                                                              \
        const T tg = top_grad[c];                                                              \
        const T top_grad_value = tg * attn;                                                    \
        T grad_h_weight = 0, grad_w_weight = 0;                                                \
        T v1 = 0, v2 = 0, v3 = 0, v4 = 0;                                                      \
        if (ok1) { v1 = lvl[o1 + c]; grad_h_weight -= hw * v1; grad_w_weight -= hh * v1;       \
                   glvl[o1 + c] += w1 * top_grad_value; }                                      \
        if (ok2) { v2 = lvl[o2 + c]; grad_h_weight -= lw * v2; grad_w_weight += hh * v2;       \
                   glvl[o2 + c] += w2 * top_grad_value; }                                      \
        if (ok3) { v3 = lvl[o3 + c]; grad_h_weight += hw * v3; grad_w_weight -= lh * v3;       \
                   glvl[o3 + c] += w3 * top_grad_value; }                                      \
        if (ok4) { v4 = lvl[o4 + c]; grad_h_weight += lw * v4; grad_w_weight += lh * v4;       \
                   glvl[o4 + c] += w4 * top_grad_value; }                                      \
        const T val = (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);                                 \
        sum_aw += tg * val;                           /* .cuh:156 */                           \
        sum_w += (T)W * grad_w_weight * top_grad_value; /* .cuh:157 */                         \
        sum_h += (T)H * grad_h_weight * top_grad_value; /* .cuh:158 */                         \
    }                                                                                          \
    *g_aw = sum_aw;                                                                            \
    g_loc[0] = sum_w;                                                                          \
    g_loc[1] = sum_h;                                                                          \
}                                                                                              \
                                                                                               \
void msda_oracle_backward_##SUFFIX(const T *value, const int64_t *shapes,                      \
                                   const int64_t *starts, const T *loc, const T *aw,           \
                                   const T *grad_out, int N, int S, int M, int D, int L,       \
                                   int Lq, int P, T *g_value, T *g_loc, T *g_aw)               \
{                                                                                              \
    const long qm_stride = (long)M * D;                                                        \
    memset(g_value, 0, sizeof(T) * (size_t)N * S * M * D);       /* .cu:121 */                 \
    memset(g_loc, 0, sizeof(T) * (size_t)N * Lq * M * L * P * 2); /* .cu:122 */                \
    memset(g_aw, 0, sizeof(T) * (size_t)N * Lq * M * L * P);      /* .cu:123 */                \
    /* (n, m) pairs touch disjoint slices of g_value, so they may run in parallel. */          \
    _Pragma("omp parallel for collapse(2) schedule(static)")                                   \
    for (int n = 0; n < N; ++n)                                                                \
        for (int m = 0; m < M; ++m)                                                            \
            for (int q = 0; q < Lq; ++q) {                                                     \
                const long qm = ((long)n * Lq + q) * M + m;                                    \
                const T *tg = grad_out + qm * D;                                               \
                const T *lp = loc + qm * L * P * 2;                                            \
                const T *wp = aw + qm * L * P;                                                 \
                for (int l = 0; l < L; ++l) {                                                  \
                    const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];              \
                    const long lvl_off = ((long)n * S + starts[l]) * qm_stride;                \
                    for (int p = 0; p < P; ++p) {                                              \
                        const long s = qm * L * P + l * P + p;                                 \
                        const T loc_w = lp[(l * P + p) * 2], loc_h = lp[(l * P + p) * 2 + 1];  \
                        const T h_im = loc_h * (T)H - (T)0.5;                                  \
                        const T w_im = loc_w * (T)W - (T)0.5;                                  \
                        if (h_im > -1 && w_im > -1 && h_im < H && w_im < W)                    \
                            bilinear_bwd_##SUFFIX(value + lvl_off, g_value + lvl_off, H, W,    \
                                                  M, D, h_im, w_im, m, tg, wp[l * P + p],      \
                                                  g_loc + 2 * s, g_aw + s);                    \
                    }                                                                          \
                }                                                                              \
            }                                                                                  \
}

MSDA_ORACLE_DEFINE(float, f32)
MSDA_ORACLE_DEFINE(double, f64)

int msda_oracle_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
