// msda_rows.hip -- backward pass of the sampling op with ONE head of 256 channels and few queries: what the decoders'
// cross-attention becomes when sampling and value projection are exchanged (rlipv2_amd/deform_attn.py:
// MSDeformAttn._sampled_projection; reference arithmetic models/ops/modules/ms_deform_attn.py:98-118, gradient formulas
// models/ops/src/cuda/ms_deform_im2col_cuda.cuh:87-159).  value = the image memory itself [N, S, 256], queries = the
// (query, head) pairs (Q = Lq x 8 = 2 400 / 1 200 per image), every sample moves a whole 512-byte row.
//
// The library's other backward routes are built for 32-channel heads; on them this call is 8 x the sampling work or, on
// the generic kernel, 614 k rows of float atomics per call.  Here, with no atomics and a fixed summation order:
//   * rows_scatter_kernel (grad of the memory): a workgroup owns a RANGE of pixels of one (image, level) and keeps their
//     256-channel float32 accumulators in LDS; each of its 4 waves owns 64 channels (lane = channel) and walks ALL the
//     level's samples in order, 64 at a time (lane = sample: geometry, corners inside the range -> ballot), and for every
//     hit adds weight x dz[query] to its channels of the pixel's accumulator -- every accumulator cell has ONE writer and
//     receives its terms in sample order: bit-repeatable for any input, however skewed.  Every row of the gradient is
//     written exactly once (untouched pixels as zeros): no zero-fill pass.
//   * rows_dots_kernel (grads of sampling locations / attention weights): a wave per (image, query'), lane = 4 channels:
//     <dz, corner row> over 256 channels by wave reduction, then the reference's formulas.
// Written in round 5 without a GPU: checked on the lane-level model of tools/emu/ (tests/test_msda_emulated_library.py), NOT yet run on
// hardware; nothing calls it unless deform_attn.sample_then_project is switched on.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rlipv2_msda.h"
#include "once_per_device.h"

#ifndef MSDA_DYNAMIC_LDS
#define MSDA_DYNAMIC_LDS(type, name) extern __shared__ __attribute__((aligned(16))) type name[]
#endif

namespace {

constexpr int RC = 256;                 // channels of a row
constexpr int RTHREADS = 256;           // 4 waves x 64 channels
constexpr int RMAX_RANGE = 128;         // pixels per workgroup: 128 x 256 x 4 B = 128 KB of accumulators
constexpr int RMAX_LEVELS = 8;

struct RowsPlan {
    int H[RMAX_LEVELS], W[RMAX_LEVELS];
    int range[RMAX_LEVELS];             // pixels per workgroup of the level
    int first[RMAX_LEVELS + 1];         // first workgroup of the level (per image)
    int L, per_image;
};

__device__ __forceinline__ float bf(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ uint16_t rne(float f)
{
    const uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
template <typename T> __device__ __forceinline__ float ld(const T *p);
template <> __device__ __forceinline__ float ld<float>(const float *p) { return *p; }
template <> __device__ __forceinline__ float ld<uint16_t>(const uint16_t *p) { return bf(*p); }
// 4 consecutive channels (8-byte / 16-byte aligned) in one load
template <typename T> __device__ __forceinline__ void ld4(const T *p, float (&v)[4]);
template <> __device__ __forceinline__ void ld4<float>(const float *p, float (&v)[4])
{
    const float4 t = *reinterpret_cast<const float4 *>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
template <> __device__ __forceinline__ void ld4<uint16_t>(const uint16_t *p, float (&v)[4])
{
    const uint2 t = *reinterpret_cast<const uint2 *>(p);
    v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
    v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
}
template <typename T> __device__ __forceinline__ void st(T *p, float v);
template <> __device__ __forceinline__ void st<float>(float *p, float v) { *p = v; }
template <> __device__ __forceinline__ void st<uint16_t>(uint16_t *p, float v) { *p = rne(v); }

// one sample of level (H, W): corner pixels (-1: outside the level) and bilinear weights; false: the sample is dropped
// (ms_deform_im2col_cuda.cuh:285-288; NaN -> false)
struct Corners { int pix[4]; float w[4]; float lw, lh; };
__device__ __forceinline__ bool corners_of(float x, float y, int H, int W, Corners &c)
{
    const float Hf = (float)H, Wf = (float)W;
    const float h_im = fmaf(y, Hf, -0.5f), w_im = fmaf(x, Wf, -0.5f);
    if (!((h_im > -1.f) && (w_im > -1.f) && (h_im < Hf) && (w_im < Wf))) return false;
    const float hf = floorf(h_im), wf = floorf(w_im);
    const int iy = (int)hf, ix = (int)wf;
    c.lh = h_im - hf; c.lw = w_im - wf;
    const float hh = 1.f - c.lh, hw = 1.f - c.lw;
    const bool y0 = iy >= 0, y1 = iy + 1 <= H - 1, x0 = ix >= 0, x1 = ix + 1 <= W - 1;
    c.pix[0] = (y0 && x0) ? iy * W + ix : -1;
    c.pix[1] = (y0 && x1) ? iy * W + ix + 1 : -1;
    c.pix[2] = (y1 && x0) ? (iy + 1) * W + ix : -1;
    c.pix[3] = (y1 && x1) ? (iy + 1) * W + ix + 1 : -1;
    c.w[0] = hh * hw; c.w[1] = hh * c.lw; c.w[2] = c.lh * hw; c.w[3] = c.lh * c.lw;
    return true;
}

template <typename T>
__global__ __launch_bounds__(RTHREADS) void rows_scatter_kernel(RowsPlan pl, const int64_t *__restrict__ starts,
                                                                const float *__restrict__ loc, const float *__restrict__ aw,
                                                                const T *__restrict__ dz, int S, int Q, int P,
                                                                T *__restrict__ g_src)
{
    MSDA_DYNAMIC_LDS(float, acc);                                   // [range][256]
    const int tid = threadIdx.x, lane = tid & 63;
    const int n = blockIdx.x / pl.per_image, b = blockIdx.x % pl.per_image;
    int l = 0;
    while (l + 1 < pl.L && b >= pl.first[l + 1]) ++l;
    const int H = pl.H[l], W = pl.W[l], range = pl.range[l];
    const int p0 = (b - pl.first[l]) * range, p1 = min(p0 + range, H * W);
    for (int i = tid; i < (p1 - p0) * RC; i += RTHREADS) acc[i] = 0.f;
    __syncthreads();
    // every wave walks all Q x P samples of (image, level) in order; it owns channels tid (= wave * 64 + lane) of every pixel
    const int samples = Q * P;
    const float *loc_n = loc + (size_t)n * Q * pl.L * P * 2;
    const float *aw_n = aw + (size_t)n * Q * pl.L * P;
    const T *dz_n = dz + (size_t)n * Q * RC;
    for (int s0 = 0; s0 < samples; s0 += 64) {
        const int s = s0 + lane;                                      // this lane's sample: (query', point)
        Corners c;
        bool any = false;
        float a = 0.f;
        int q = 0;
        if (s < samples) {
            q = s / P;
            const int p = s - q * P;
            const size_t e = ((size_t)q * pl.L + l) * P + p;
            if (corners_of(loc_n[e * 2], loc_n[e * 2 + 1], H, W, c)) {
                a = aw_n[e];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const bool in = c.pix[k] >= p0 && c.pix[k] < p1;
                    c.pix[k] = in ? c.pix[k] - p0 : -1;
                    any = any || in;
                }
            }
        }
        unsigned long long hits = __builtin_amdgcn_ballot_w64(any);
        while (hits) {                                                // (wave-uniform: samples in increasing order)
            const int src_lane = __builtin_ctzll(hits);
            hits &= hits - 1;
            const int qh = __builtin_amdgcn_readlane(q, src_lane);
            const float ah = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a), src_lane));
            const float g = ld<T>(dz_n + (size_t)qh * RC + tid) * ah;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int pk = __builtin_amdgcn_readlane(c.pix[k], src_lane);
                const float wk = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c.w[k]), src_lane));
                if (pk >= 0) acc[pk * RC + tid] += wk * g;            // (wave-uniform branch; one writer per cell)
            }
        }
    }
    // this thread's channel of every pixel of the range (a pixel's 256 channels are written by the 256 threads together)
    T *out = g_src + ((size_t)n * S + (size_t)starts[l] + p0) * RC;
    for (int p = 0; p < p1 - p0; ++p) st<T>(out + (size_t)p * RC + tid, acc[p * RC + tid]);
}

__device__ __forceinline__ float wave_sum64(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// grads of the sampling locations / attention weights: a wave per (image, query'), lane = channels 4 lane .. 4 lane + 3
template <typename T>
__global__ __launch_bounds__(RTHREADS) void rows_dots_kernel(RowsPlan pl, const int64_t *__restrict__ starts,
                                                             const T *__restrict__ src, const float *__restrict__ loc,
                                                             const float *__restrict__ aw, const T *__restrict__ dz, int N, int S,
                                                             int Q, int P, float *__restrict__ g_loc, float *__restrict__ g_aw)
{
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * (RTHREADS / 64) + (threadIdx.x >> 6);       // (image, query')
    if (row >= (long)N * Q) return;                                   // (a whole wave)
    const int n = (int)(row / Q);
    float g[4];
    ld4<T>(dz + row * RC + lane * 4, g);
    const T *src_n = src + (size_t)n * S * RC;
    for (int l = 0; l < pl.L; ++l) {
        const int H = pl.H[l], W = pl.W[l];
        const T *lvl = src_n + (size_t)starts[l] * RC;
        for (int p = 0; p < P; ++p) {
            const size_t e = ((size_t)row * pl.L + l) * P + p;
            Corners c;
            float ga = 0.f, gx = 0.f, gy = 0.f;
            if (corners_of(loc[e * 2], loc[e * 2 + 1], H, W, c)) {   // (wave-uniform: every lane reads the same sample)
                float dot[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float d = 0.f;
                    if (c.pix[k] >= 0) {
                        float v[4];
                        ld4<T>(lvl + (size_t)c.pix[k] * RC + lane * 4, v);
#pragma unroll
                        for (int j = 0; j < 4; ++j) d = fmaf(g[j], v[j], d);
                    }
                    dot[k] = wave_sum64(d);
                }
                const float a = aw[e], hh = 1.f - c.lh, hw = 1.f - c.lw;
                ga = c.w[0] * dot[0] + c.w[1] * dot[1] + c.w[2] * dot[2] + c.w[3] * dot[3];
                gx = (float)W * a * (hh * (dot[1] - dot[0]) + c.lh * (dot[3] - dot[2]));
                gy = (float)H * a * (hw * (dot[2] - dot[0]) + c.lw * (dot[3] - dot[1]));
            }
            if (lane == 0) { g_aw[e] = ga; g_loc[e * 2] = gx; g_loc[e * 2 + 1] = gy; }
        }
    }
}

bool make_rows_plan(const int64_t *hs, int L, int S, RowsPlan &pl)
{
    if (!hs || L < 1 || L > RMAX_LEVELS) return false;
    long sum = 0;
    int first = 0;
    pl.L = L;
    for (int l = 0; l < L; ++l) {
        const int64_t H = hs[2 * l], W = hs[2 * l + 1];
        if (H < 1 || W < 1 || H > 8192 || W > 8192) return false;
        pl.H[l] = (int)H; pl.W[l] = (int)W;
        const long px = H * W;
        sum += px;
        // few workgroups on the fine levels (every workgroup scans all samples of its level), a full LDS tile where the
        // records are dense
        int range = px >= 8192 ? 128 : px >= 2048 ? 64 : px >= 512 ? 32 : 16;
        pl.range[l] = range;
        pl.first[l] = first;
        first += (int)((px + range - 1) / range);
    }
    for (int l = L; l <= RMAX_LEVELS; ++l) pl.first[l] = first;
    pl.per_image = first;
    return sum == S;
}

template <typename T>
int launch_rows(const RowsPlan &pl, const void *src, const int64_t *starts, const void *loc, const void *aw, const void *dz, int N,
                int S, int Q, int P, void *g_src, void *g_loc, void *g_aw, hipStream_t stream)
{
    RLIPV2_ONCE_PER_DEVICE((void)hipFuncSetAttribute((const void *)rows_scatter_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                     RMAX_RANGE * RC * 4));
    int max_range = 0;
    for (int l = 0; l < pl.L; ++l) max_range = pl.range[l] > max_range ? pl.range[l] : max_range;
    (void)hipGetLastError();
    hipLaunchKernelGGL((rows_scatter_kernel<T>), dim3(N * pl.per_image), dim3(RTHREADS), max_range * RC * 4, stream, pl, starts,
                       (const float *)loc, (const float *)aw, (const T *)dz, S, Q, P, (T *)g_src);
    const long rows = (long)N * Q;
    hipLaunchKernelGGL((rows_dots_kernel<T>), dim3((unsigned)((rows + RTHREADS / 64 - 1) / (RTHREADS / 64))), dim3(RTHREADS), 0, stream,
                       pl, starts, (const T *)src, (const float *)loc, (const float *)aw, (const T *)dz, N, S, Q, P, (float *)g_loc,
                       (float *)g_aw);
    return hipGetLastError() == hipSuccess ? MSDA_OK : MSDA_ERR_LAUNCH;
}

}  // namespace

extern "C" int msda_rows_backward_supported(int dtype, const int64_t *shapes_host, int N, int S, int C, int L, int Q, int P)
{
    RowsPlan pl;
    return (dtype == MSDA_F32 || dtype == MSDA_BF16) && C == RC && N >= 1 && Q >= 1 && P >= 1 && make_rows_plan(shapes_host, L, S, pl) &&
           (long)N * pl.per_image < (1L << 30) && (long)N * Q * L * P < (1L << 30);
}

extern "C" int msda_rows_backward(int dtype, const void *src, const int64_t *level_start, const int64_t *shapes_host,
                                  const void *sampling_loc, const void *attn_weight, const void *grad_out, int N, int S, int C, int L,
                                  int Q, int P, void *grad_src, void *grad_sampling_loc, void *grad_attn_weight, void *stream)
{
    if (!msda_rows_backward_supported(dtype, shapes_host, N, S, C, L, Q, P)) return MSDA_ERR_BAD_SHAPE;
    if (!src || !level_start || !sampling_loc || !attn_weight || !grad_out || !grad_src || !grad_sampling_loc || !grad_attn_weight)
        return MSDA_ERR_NULL_POINTER;
    RowsPlan pl;
    make_rows_plan(shapes_host, L, S, pl);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MSDA_BF16)
        return launch_rows<uint16_t>(pl, src, level_start, sampling_loc, attn_weight, grad_out, N, S, Q, P, grad_src, grad_sampling_loc,
                                     grad_attn_weight, s);
    return launch_rows<float>(pl, src, level_start, sampling_loc, attn_weight, grad_out, N, S, Q, P, grad_src, grad_sampling_loc,
                              grad_attn_weight, s);
}
