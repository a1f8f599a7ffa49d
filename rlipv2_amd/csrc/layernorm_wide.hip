// layernorm_wide.hip -- LayerNorm over the channel widths of the Swin backbones (96 ... 1 536 channels; reference
// models/swin/swin_transformer.py:304-403: the pre-norm residual block  x = x + branch;  y = norm(x)) as one HBM pass per
// direction, with the residual add in front and the residual's gradient behind (include/rlipv2_norm.h).
//
//   forward :  s = a + b (b optional),  y = LN(s) * gamma + beta          writes y, and s when the caller needs it
//   backward:  dx = LN'(dy) + ds (ds optional: the gradient that reaches s on the residual path)
//
// The 256-channel kernels of add_layernorm.hip (the encoder's post-norm blocks) stay as they are -- their device code is
// pinned to the hardware run behind it.  This file generalises their mapping: a row is LANES lanes x VEC 16-byte vectors
// (C = 8 VEC LANES), lane l owns vectors l, l + LANES, l + 2 LANES, so that every load instruction of a row covers LANES
// consecutive 16-byte pieces; the row statistics are two xor butterflies over LANES lanes, float32, two-pass in registers.
// HBM-bound: forward moves (2-3) x rows x 2C bytes, backward (3-4) x.  The Swin norms are frozen by the reference
// (models/swin/backbone.py:66-69), so there are no gamma / beta gradients here; a caller that trains them keeps PyTorch's op.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rlipv2_msda.h"
#include "../../include/rlipv2_norm.h"

namespace {

constexpr int THREADS = 256;
constexpr int MAX_BLOCKS = 2048;

__device__ __forceinline__ float lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// two floats -> packed bfloat16 pair, round to nearest even (one v_cvt_pk_bf16_f32)
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__device__ __forceinline__ uint32_t pack2(float a, float b)
{
    const f32x2_t f = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2_t));
}

__device__ __forceinline__ void unpack8(const uint4 &v, float *f)
{
    f[0] = lo(v.x); f[1] = hi(v.x); f[2] = lo(v.y); f[3] = hi(v.y);
    f[4] = lo(v.z); f[5] = hi(v.z); f[6] = lo(v.w); f[7] = hi(v.w);
}

__device__ __forceinline__ uint4 pack8(const float *o)
{
    return make_uint4(pack2(o[0], o[1]), pack2(o[2], o[3]), pack2(o[4], o[5]), pack2(o[6], o[7]));
}

// sum over the LANES lanes of a row (rows are aligned groups of LANES lanes: the xor butterfly stays inside the group)
template <int LANES> __device__ __forceinline__ float row_sum(float v)
{
#pragma unroll
    for (int off = LANES / 2; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <int VEC, int LANES, bool HAS_B, bool SUM_OUT>
__global__ __launch_bounds__(THREADS) void wide_forward_kernel(const uint4 *__restrict__ a, const uint4 *__restrict__ b,
                                                               const uint4 *__restrict__ gamma, const uint4 *__restrict__ beta,
                                                               long rows, float eps, uint4 *__restrict__ y,
                                                               uint4 *__restrict__ sum, float *__restrict__ mean,
                                                               float *__restrict__ rstd)
{
    constexpr int C = 8 * VEC * LANES, ROWS_PER_STEP = THREADS / LANES, RV = VEC * LANES;      // RV: 16-byte vectors per row
    const int lane = threadIdx.x & (LANES - 1), slot = threadIdx.x / LANES;
    const long stride = (long)gridDim.x * ROWS_PER_STEP;
    // (the trip count is uniform over the workgroup: a row slot past the end re-reads the last row and stores nothing, so the
    //  butterflies below always run with every lane of the wave)
    for (long r0 = (long)blockIdx.x * ROWS_PER_STEP; r0 < rows; r0 += stride) {
        const bool live = r0 + slot < rows;
        const long r = live ? r0 + slot : rows - 1;
        float x[8 * VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) unpack8(a[r * RV + v * LANES + lane], x + 8 * v);
        if (HAS_B) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                float t[8];
                unpack8(b[r * RV + v * LANES + lane], t);
#pragma unroll
                for (int j = 0; j < 8; ++j) x[8 * v + j] += t[j];
            }
            if (SUM_OUT) {
                // the sum leaves as bfloat16 and the statistics are taken from the ROUNDED sum: y is then exactly the
                // LayerNorm of the tensor the next block reads (what `norm(x + branch)` computes as two ops)
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    const uint4 p = pack8(x + 8 * v);
                    if (live) sum[r * RV + v * LANES + lane] = p;
                    unpack8(p, x + 8 * v);
                }
            }
        }
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 8 * VEC; ++j) s += x[j];
        const float mu = row_sum<LANES>(s) * (1.f / C);
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < 8 * VEC; ++j) { x[j] -= mu; q += x[j] * x[j]; }
        const float rs = rsqrtf(row_sum<LANES>(q) * (1.f / C) + eps);
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            float g[8], bt[8], o[8];
            unpack8(gamma[v * LANES + lane], g);           // (L1-resident: 2C bytes per tensor for the whole grid)
            unpack8(beta[v * LANES + lane], bt);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = x[8 * v + j] * rs * g[j] + bt[j];
            if (live) y[r * RV + v * LANES + lane] = pack8(o);
        }
        if (live && lane == 0) { mean[r] = mu; rstd[r] = rs; }
    }
}

// dx = rstd (g dy - mean(g dy) - xhat mean(g dy xhat)) + ds        (x: the saved input of the norm, i.e. the sum)
template <int VEC, int LANES, bool HAS_DS>
__global__ __launch_bounds__(THREADS) void wide_backward_kernel(const uint4 *__restrict__ dy, const uint4 *__restrict__ ds,
                                                                const uint4 *__restrict__ x_in, const uint4 *__restrict__ gamma,
                                                                const float *__restrict__ mean, const float *__restrict__ rstd,
                                                                long rows, uint4 *__restrict__ dx)
{
    constexpr int C = 8 * VEC * LANES, ROWS_PER_STEP = THREADS / LANES, RV = VEC * LANES;
    const int lane = threadIdx.x & (LANES - 1), slot = threadIdx.x / LANES;
    const long stride = (long)gridDim.x * ROWS_PER_STEP;
    for (long r0 = (long)blockIdx.x * ROWS_PER_STEP; r0 < rows; r0 += stride) {       // (uniform trip count, see the forward)
        const bool live = r0 + slot < rows;
        const long r = live ? r0 + slot : rows - 1;
        float x[8 * VEC], gd[8 * VEC];
        const float mu = mean[r], rs = rstd[r];
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            float d[8], g[8];
            unpack8(x_in[r * RV + v * LANES + lane], x + 8 * v);
            unpack8(dy[r * RV + v * LANES + lane], d);
            unpack8(gamma[v * LANES + lane], g);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                x[8 * v + j] = (x[8 * v + j] - mu) * rs;     // normalised input
                gd[8 * v + j] = d[j] * g[j];
                c1 += gd[8 * v + j];
                c2 += gd[8 * v + j] * x[8 * v + j];
            }
        }
        c1 = row_sum<LANES>(c1) * (1.f / C);
        c2 = row_sum<LANES>(c2) * (1.f / C);
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = rs * (gd[8 * v + j] - c1 - x[8 * v + j] * c2);
            if (HAS_DS) {
                float t[8];
                unpack8(ds[r * RV + v * LANES + lane], t);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] += t[j];
            }
            if (live) dx[r * RV + v * LANES + lane] = pack8(o);
        }
    }
}

struct Shape { int vec, lanes; };

// C = 8 VEC LANES with LANES a power of two <= 64 and VEC in {1, 2, 3}: 96, 128, 192, 384, 512, 768, 1 024, 1 536 (the Swin
// tiny / small / base / large stages); 256 belongs to add_layernorm.hip
bool shape_of(int channels, Shape &s)
{
    switch (channels) {
    case 96: s = {3, 4}; return true;
    case 128: s = {1, 16}; return true;
    case 192: s = {3, 8}; return true;
    case 384: s = {3, 16}; return true;
    case 512: s = {1, 64}; return true;
    case 768: s = {3, 32}; return true;
    case 1024: s = {2, 64}; return true;
    case 1536: s = {3, 64}; return true;
    default: return false;
    }
}

int grid_for(long rows, int lanes)
{
    const int per = THREADS / lanes;
    long blocks = (rows + per - 1) / per;
    return (int)(blocks < MAX_BLOCKS ? (blocks < 1 ? 1 : blocks) : MAX_BLOCKS);
}

bool misaligned(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) != 0; }

template <int VEC, int LANES>
void launch_forward(const void *a, const void *b, const void *gamma, const void *beta, long rows, float eps, void *y, void *sum,
                    float *mean, float *rstd, hipStream_t stream)
{
    const dim3 grid(grid_for(rows, LANES)), block(THREADS);
#define RLIPV2_WIDE_FWD(HAS_B, SUM_OUT)                                                                                 \
    hipLaunchKernelGGL((wide_forward_kernel<VEC, LANES, HAS_B, SUM_OUT>), grid, block, 0, stream, (const uint4 *)a,    \
                       (const uint4 *)b, (const uint4 *)gamma, (const uint4 *)beta, rows, eps, (uint4 *)y, (uint4 *)sum, mean, rstd)
    if (!b) RLIPV2_WIDE_FWD(false, false);
    else if (!sum) RLIPV2_WIDE_FWD(true, false);
    else RLIPV2_WIDE_FWD(true, true);
#undef RLIPV2_WIDE_FWD
}

template <int VEC, int LANES>
void launch_backward(const void *dy, const void *ds, const void *x, const void *gamma, const float *mean, const float *rstd,
                     long rows, void *dx, hipStream_t stream)
{
    const dim3 grid(grid_for(rows, LANES)), block(THREADS);
    if (ds)
        hipLaunchKernelGGL((wide_backward_kernel<VEC, LANES, true>), grid, block, 0, stream, (const uint4 *)dy, (const uint4 *)ds,
                           (const uint4 *)x, (const uint4 *)gamma, mean, rstd, rows, (uint4 *)dx);
    else
        hipLaunchKernelGGL((wide_backward_kernel<VEC, LANES, false>), grid, block, 0, stream, (const uint4 *)dy,
                           (const uint4 *)nullptr, (const uint4 *)x, (const uint4 *)gamma, mean, rstd, rows, (uint4 *)dx);
}

}  // namespace

#define RLIPV2_WIDE_DISPATCH(CALL)                                                                                      \
    switch (channels) {                                                                                                 \
    case 96: CALL(3, 4); break;                                                                                         \
    case 128: CALL(1, 16); break;                                                                                       \
    case 192: CALL(3, 8); break;                                                                                        \
    case 384: CALL(3, 16); break;                                                                                       \
    case 512: CALL(1, 64); break;                                                                                       \
    case 768: CALL(3, 32); break;                                                                                       \
    case 1024: CALL(2, 64); break;                                                                                      \
    default: CALL(3, 64); break;                                                                                        \
    }

extern "C" int layernorm_wide_supported(long rows, int channels)
{
    Shape s;
    return rows >= 1 && shape_of(channels, s);
}

extern "C" int layernorm_wide_forward_bf16(const void *a, const void *b, const void *gamma, const void *beta, long rows,
                                           int channels, float eps, void *y, void *sum, float *mean, float *rstd, void *stream_)
{
    if (!layernorm_wide_supported(rows, channels)) return MSDA_ERR_BAD_SHAPE;
    if (!a || !gamma || !beta || !y || !mean || !rstd) return MSDA_ERR_NULL_POINTER;
    if (sum && !b) return MSDA_ERR_NULL_POINTER;                      // the sum of one tensor is that tensor
    if (misaligned(a) || misaligned(b) || misaligned(gamma) || misaligned(beta) || misaligned(y) || misaligned(sum))
        return MSDA_ERR_ALIGNMENT;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    (void)hipGetLastError();
#define RLIPV2_WIDE_CALL(V, L) launch_forward<V, L>(a, b, gamma, beta, rows, eps, y, sum, mean, rstd, stream)
    RLIPV2_WIDE_DISPATCH(RLIPV2_WIDE_CALL)
#undef RLIPV2_WIDE_CALL
    return hipGetLastError() == hipSuccess ? MSDA_OK : MSDA_ERR_LAUNCH;
}

extern "C" int layernorm_wide_backward_bf16(const void *dy, const void *dsum, const void *x, const void *gamma, const float *mean,
                                            const float *rstd, long rows, int channels, void *dx, void *stream_)
{
    if (!layernorm_wide_supported(rows, channels)) return MSDA_ERR_BAD_SHAPE;
    if (!dy || !x || !gamma || !mean || !rstd || !dx) return MSDA_ERR_NULL_POINTER;
    if (misaligned(dy) || misaligned(dsum) || misaligned(x) || misaligned(gamma) || misaligned(dx)) return MSDA_ERR_ALIGNMENT;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    (void)hipGetLastError();
#define RLIPV2_WIDE_CALL(V, L) launch_backward<V, L>(dy, dsum, x, gamma, mean, rstd, rows, dx, stream)
    RLIPV2_WIDE_DISPATCH(RLIPV2_WIDE_CALL)
#undef RLIPV2_WIDE_CALL
    return hipGetLastError() == hipSuccess ? MSDA_OK : MSDA_ERR_LAUNCH;
}
