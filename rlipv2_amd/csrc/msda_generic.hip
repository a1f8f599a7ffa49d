// msda_generic.hip -- shape-agnostic MSDA kernels for gfx950 (any M, D, L, P; f32 / f64 / bf16).
//
// Mapping: one 64-lane wavefront owns one (n, q, m) triple at a time; lanes stride over the D
// channels, so the sampling location / attention weight of a sample are wave-uniform (served by
// one broadcast load) and the two channel reductions of the backward pass are wave reductions
// done with cross-lane shuffles -- no LDS, no barriers.
//
// This is the correctness backstop behind the D = 32 fast paths (msda_quad.hip,
// msda_window.hip) and the only path for float64 (the reference's gradcheck type,
// models/ops/test.py:67-82) and for head sizes other than 32 (test.py:89-90 runs
// D in {30, 32, 64, 71, 1025, 2048, 3096}).
//
// Semantics follow the reference kernels (what, not how):
//   sample inclusion rule           ms_deform_im2col_cuda.cuh:285-288
//   4 guarded bilinear corners      ms_deform_im2col_cuda.cuh:33-84
//   backward formulas               ms_deform_im2col_cuda.cuh:87-159
#include "msda_device.h"
#include "msda_internal.h"

namespace msda {

namespace {

constexpr int kBlock = 256;            // 4 wavefronts
constexpr int kWavesPerBlock = kBlock / 64;

template <typename T> struct SampleGeom {
    bool inside;
    int o1, o2, o3, o4;      // element offsets (in units of M*D rows) of the four corners, level-relative
    bool ok1, ok2, ok3, ok4;
    T lh, lw, hh, hw;
};

// geometry of one sample; everything here is wave-uniform
template <typename T>
__device__ __forceinline__ SampleGeom<T> sample_geom(T loc_w, T loc_h, int H, int W)
{
    SampleGeom<T> g;
    const T h_im = loc_h * (T)H - (T)0.5;
    const T w_im = loc_w * (T)W - (T)0.5;
    g.inside = (h_im > (T)-1) && (w_im > (T)-1) && (h_im < (T)H) && (w_im < (T)W);
    const T hs = g.inside ? h_im : (T)0, ws = g.inside ? w_im : (T)0;
    const T hf = floor(hs), wf = floor(ws);
    const int h_low = (int)hf, w_low = (int)wf;
    const int h_high = h_low + 1, w_high = w_low + 1;
    g.lh = hs - hf; g.lw = ws - wf;
    g.hh = (T)1 - g.lh; g.hw = (T)1 - g.lw;
    const bool hl = h_low >= 0, hhv = h_high <= H - 1, wl = w_low >= 0, wh = w_high <= W - 1;
    g.ok1 = g.inside && hl && wl;  g.ok2 = g.inside && hl && wh;
    g.ok3 = g.inside && hhv && wl; g.ok4 = g.inside && hhv && wh;
    const int rl = max(h_low, 0) * W, rh = min(h_high, H - 1) * W;
    const int cl = max(w_low, 0), ch = min(w_high, W - 1);
    g.o1 = rl + cl; g.o2 = rl + ch; g.o3 = rh + cl; g.o4 = rh + ch;
    return g;
}

template <typename T, typename VT>
__global__ __launch_bounds__(kBlock) void generic_forward_kernel(
    const VT *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ starts,
    const T *__restrict__ loc, const T *__restrict__ aw, int N, int S, int M, int D, int L, int Lq, int P,
    VT *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const long nwaves = (long)gridDim.x * kWavesPerBlock;
    const long total = (long)N * Lq * M;
    const long row = (long)M * D;
    for (long qm = wave; qm < total; qm += nwaves) {
        const int m = (int)(qm % M);
        const long n = qm / ((long)Lq * M);
        const T *lp = loc + qm * L * P * 2;
        const T *wp = aw + qm * L * P;
        const VT *vn = value + n * S * row + (long)m * D;
        for (int c0 = 0; c0 < D; c0 += 64) {
            const int c = c0 + lane;
            const bool act = c < D;
            const int cc = act ? c : 0;
            T acc = 0;
            for (int l = 0; l < L; ++l) {
                const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
                const VT *vl = vn + (long)starts[l] * row + cc;
                for (int p = 0; p < P; ++p) {
                    const int s = l * P + p;
                    const SampleGeom<T> g = sample_geom<T>(lp[2 * s], lp[2 * s + 1], H, W);
                    if (!g.inside) continue;
                    const T v1 = g.ok1 ? Elem<T, VT>::ld(vl + (long)g.o1 * row) : (T)0;
                    const T v2 = g.ok2 ? Elem<T, VT>::ld(vl + (long)g.o2 * row) : (T)0;
                    const T v3 = g.ok3 ? Elem<T, VT>::ld(vl + (long)g.o3 * row) : (T)0;
                    const T v4 = g.ok4 ? Elem<T, VT>::ld(vl + (long)g.o4 * row) : (T)0;
                    const T val = g.hh * g.hw * v1 + g.hh * g.lw * v2 + g.lh * g.hw * v3 + g.lh * g.lw * v4;
                    acc += val * wp[s];
                }
            }
            if (act) Elem<T, VT>::st(out + qm * D + c, acc);
        }
    }
}

// GT = type of grad_value (== T: float for bf16 storage)
template <typename T, typename VT>
__global__ __launch_bounds__(kBlock) void generic_backward_kernel(
    const VT *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ starts,
    const T *__restrict__ loc, const T *__restrict__ aw, const VT *__restrict__ grad_out, int N, int S, int M,
    int D, int L, int Lq, int P, T *__restrict__ g_value, T *__restrict__ g_loc, T *__restrict__ g_aw)
{
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const long nwaves = (long)gridDim.x * kWavesPerBlock;
    const long total = (long)N * Lq * M;
    const long row = (long)M * D;
    for (long qm = wave; qm < total; qm += nwaves) {
        const int m = (int)(qm % M);
        const long n = qm / ((long)Lq * M);
        const T *lp = loc + qm * L * P * 2;
        const T *wp = aw + qm * L * P;
        const long img = n * S * row + (long)m * D;
        for (int l = 0; l < L; ++l) {
            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
            const long lvl = img + (long)starts[l] * row;
            for (int p = 0; p < P; ++p) {
                const int s = l * P + p;
                const SampleGeom<T> g = sample_geom<T>(lp[2 * s], lp[2 * s + 1], H, W);
                T s_aw = 0, s_w = 0, s_h = 0;
                if (g.inside) {
                    const T attn = wp[s];
                    const T w1 = g.hh * g.hw, w2 = g.hh * g.lw, w3 = g.lh * g.hw, w4 = g.lh * g.lw;
                    for (int c = lane; c < D; c += 64) {
                        const T tg = Elem<T, VT>::ld(grad_out + qm * D + c);
                        const T tgv = tg * attn;
                        T v1 = 0, v2 = 0, v3 = 0, v4 = 0;
                        if (g.ok1) { v1 = Elem<T, VT>::ld(value + lvl + (long)g.o1 * row + c);
                                     atomic_add(g_value + lvl + (long)g.o1 * row + c, w1 * tgv); }
                        if (g.ok2) { v2 = Elem<T, VT>::ld(value + lvl + (long)g.o2 * row + c);
                                     atomic_add(g_value + lvl + (long)g.o2 * row + c, w2 * tgv); }
                        if (g.ok3) { v3 = Elem<T, VT>::ld(value + lvl + (long)g.o3 * row + c);
                                     atomic_add(g_value + lvl + (long)g.o3 * row + c, w3 * tgv); }
                        if (g.ok4) { v4 = Elem<T, VT>::ld(value + lvl + (long)g.o4 * row + c);
                                     atomic_add(g_value + lvl + (long)g.o4 * row + c, w4 * tgv); }
                        s_aw += tg * (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);
                        s_w += tgv * (g.hh * (v2 - v1) + g.lh * (v4 - v3));
                        s_h += tgv * (g.hw * (v3 - v1) + g.lw * (v4 - v2));
                    }
                }
                // `inside` is wave-uniform, so every lane reaches the shuffles
                s_aw = wave_sum(s_aw);
                s_w = wave_sum(s_w);
                s_h = wave_sum(s_h);
                if (lane == 0) {
                    const long si = qm * L * P + s;
                    g_aw[si] = s_aw;
                    g_loc[2 * si] = (T)W * s_w;
                    g_loc[2 * si + 1] = (T)H * s_h;
                }
            }
        }
    }
}

inline int grid_for(long total_waves)
{
    const long blocks = (total_waves + kWavesPerBlock - 1) / kWavesPerBlock;
    const long cap = 256L * 16;  // 256 CUs x 16 resident blocks is plenty; the rest grid-strides
    return (int)(blocks < 1 ? 1 : (blocks > cap ? cap : blocks));
}

template <typename T, typename VT> void fwd(const Problem &p)
{
    const long total = (long)p.N * p.Lq * p.M;
    hipLaunchKernelGGL((generic_forward_kernel<T, VT>), dim3(grid_for(total)), dim3(kBlock), 0, p.stream,
                       (const VT *)p.value, p.shapes, p.starts, (const T *)p.loc, (const T *)p.aw, p.N, p.S,
                       p.M, p.D, p.L, p.Lq, p.P, (VT *)p.out);
}
template <typename T, typename VT> void bwd(const Problem &p)
{
    const long total = (long)p.N * p.Lq * p.M;
    hipLaunchKernelGGL((generic_backward_kernel<T, VT>), dim3(grid_for(total)), dim3(kBlock), 0, p.stream,
                       (const VT *)p.value, p.shapes, p.starts, (const T *)p.loc, (const T *)p.aw,
                       (const VT *)p.grad_out, p.N, p.S, p.M, p.D, p.L, p.Lq, p.P, (T *)p.g_value, (T *)p.g_loc,
                       (T *)p.g_aw);
}

}  // namespace

void launch_generic_forward(const Problem &p)
{
    switch (p.dtype) {
        case MSDA_F32: fwd<float, float>(p); break;
        case MSDA_F64: fwd<double, double>(p); break;
        default: fwd<float, bf16_t>(p); break;
    }
}

void launch_generic_backward(const Problem &p)
{
    switch (p.dtype) {
        case MSDA_F32: bwd<float, float>(p); break;
        case MSDA_F64: bwd<double, double>(p); break;
        default: bwd<float, bf16_t>(p); break;
    }
}

}  // namespace msda
