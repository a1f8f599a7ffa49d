// add_layernorm.hip -- fused residual add + LayerNorm over 256 channels, forward and backward
// (include/rlipv2_norm.h).  HBM-bound: forward moves 3 x rows x 512 B, backward 4 x rows x 512 B.
//
// Mapping: 32 lanes per row, 8 channels (one 16-byte load) per lane, so a wave64 covers two rows and the
// row reductions are 5 DPP/shuffle steps inside a half-wave; a workgroup of 256 threads walks 8 rows per
// step over a grid-stride range, two steps unrolled so that 4-6 independent 16-byte loads per lane are in
// flight.  Statistics two-pass in registers (mean, then sum of squared deviations), float32.
// Backward additionally keeps per-lane dgamma / dbeta partials for its 8 channels, folds the 8 row slots
// of the workgroup through LDS and writes one [2][256] partial per workgroup; `finish_param_grads` sums
// the partials (a second tiny launch instead of 2 x 256 float atomics per workgroup on the same lines).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rlipv2_msda.h"
#include "../../include/rlipv2_norm.h"

namespace {

constexpr int C = 256, LANES = 32, THREADS = 256, ROWS_PER_STEP = THREADS / LANES;
constexpr int MAX_BLOCKS = 1024;

__device__ __forceinline__ float lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

__device__ __forceinline__ uint32_t pack2(float a, float b)
{
    auto rne = [](float f) -> uint32_t {
        uint32_t u = __float_as_uint(f);
        if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
        return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
    };
    return rne(a) | (rne(b) << 16);
}

__device__ __forceinline__ void unpack8(const uint4 &v, float (&f)[8])
{
    f[0] = lo(v.x); f[1] = hi(v.x); f[2] = lo(v.y); f[3] = hi(v.y);
    f[4] = lo(v.z); f[5] = hi(v.z); f[6] = lo(v.w); f[7] = hi(v.w);
}

// sum over the 32 lanes of a row (xor butterfly stays inside the half-wave)
__device__ __forceinline__ float row_sum(float v)
{
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <bool HAS_B>
__global__ __launch_bounds__(THREADS) void forward_kernel(const uint4 *__restrict__ a, const uint4 *__restrict__ b,
                                                          const uint4 *__restrict__ gamma, const uint4 *__restrict__ beta,
                                                          long rows, float eps, uint4 *__restrict__ y,
                                                          float *__restrict__ mean, float *__restrict__ rstd)
{
    const int lane = threadIdx.x & (LANES - 1), slot = threadIdx.x / LANES;
    float g[8], bt[8];
    unpack8(gamma[lane], g);
    unpack8(beta[lane], bt);
    const long stride = (long)gridDim.x * ROWS_PER_STEP;
    for (long r = (long)blockIdx.x * ROWS_PER_STEP + slot; r < rows; r += stride) {
        float x[8];
        unpack8(a[r * LANES + lane], x);
        if (HAS_B) {
            float t[8];
            unpack8(b[r * LANES + lane], t);
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] += t[j];
        }
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) s += x[j];
        const float mu = row_sum(s) * (1.f / C);
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { x[j] -= mu; q += x[j] * x[j]; }
        const float rs = rsqrtf(row_sum(q) * (1.f / C) + eps);
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = x[j] * rs * g[j] + bt[j];
        y[r * LANES + lane] = make_uint4(pack2(o[0], o[1]), pack2(o[2], o[3]), pack2(o[4], o[5]), pack2(o[6], o[7]));
        if (lane == 0) { mean[r] = mu; rstd[r] = rs; }
    }
}

template <bool HAS_B>
__global__ __launch_bounds__(THREADS) void backward_kernel(const uint4 *__restrict__ dy, const uint4 *__restrict__ a,
                                                           const uint4 *__restrict__ b, const uint4 *__restrict__ gamma,
                                                           const float *__restrict__ mean, const float *__restrict__ rstd,
                                                           long rows, uint4 *__restrict__ dx, float *__restrict__ partial)
{
    __shared__ float red[ROWS_PER_STEP][2 * C];
    const int lane = threadIdx.x & (LANES - 1), slot = threadIdx.x / LANES;
    float g[8];
    unpack8(gamma[lane], g);
    float dg[8], db[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { dg[j] = 0.f; db[j] = 0.f; }
    const long stride = (long)gridDim.x * ROWS_PER_STEP;
    for (long r = (long)blockIdx.x * ROWS_PER_STEP + slot; r < rows; r += stride) {
        float x[8], d[8];
        unpack8(a[r * LANES + lane], x);
        unpack8(dy[r * LANES + lane], d);
        if (HAS_B) {
            float t[8];
            unpack8(b[r * LANES + lane], t);
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] += t[j];
        }
        const float mu = mean[r], rs = rstd[r];
        float c1 = 0.f, c2 = 0.f, gd[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            x[j] = (x[j] - mu) * rs;                       // normalised input
            gd[j] = d[j] * g[j];
            c1 += gd[j];
            c2 += gd[j] * x[j];
            dg[j] += d[j] * x[j];
            db[j] += d[j];
        }
        c1 = row_sum(c1) * (1.f / C);
        c2 = row_sum(c2) * (1.f / C);
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = rs * (gd[j] - c1 - x[j] * c2);
        dx[r * LANES + lane] = make_uint4(pack2(o[0], o[1]), pack2(o[2], o[3]), pack2(o[4], o[5]), pack2(o[6], o[7]));
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        red[slot][lane * 8 + j] = dg[j];
        red[slot][C + lane * 8 + j] = db[j];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * C; c += THREADS) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < ROWS_PER_STEP; ++k) s += red[k][c];
        partial[(size_t)blockIdx.x * 2 * C + c] = s;
    }
}

// dgamma[c] = sum_blocks partial[blk][c], dbeta[c] = sum_blocks partial[blk][C + c].  One workgroup per 8
// columns, 32 lanes per column each summing every 32nd partial row, then a half-wave butterfly
// (64 workgroups x 32 loads per lane; the first version -- 16 workgroups, 256 dependent loads per lane --
// took 76 us, longer than the backward kernel itself).
__global__ __launch_bounds__(256) void finish_param_grads(const float *__restrict__ partial, int blocks,
                                                          uint16_t *__restrict__ dgamma, uint16_t *__restrict__ dbeta)
{
    const int part = threadIdx.x & 31, col = blockIdx.x * 8 + (threadIdx.x >> 5);
    float s = 0.f;
#pragma unroll 4
    for (int k = part; k < blocks; k += 32) s += partial[(size_t)k * 2 * C + col];
    s = row_sum(s);
    if (part == 0) {
        const uint16_t v = (uint16_t)(pack2(s, 0.f) & 0xffffu);
        if (col < C) dgamma[col] = v; else dbeta[col - C] = v;
    }
}

int grid_for(long rows)
{
    long blocks = (rows + ROWS_PER_STEP - 1) / ROWS_PER_STEP;
    return (int)(blocks < MAX_BLOCKS ? (blocks < 1 ? 1 : blocks) : MAX_BLOCKS);
}

bool misaligned(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) != 0; }

}  // namespace

extern "C" int add_layernorm_supported(long rows, int channels) { return rows >= 1 && channels == C; }

extern "C" size_t add_layernorm_workspace_bytes(long rows, int channels)
{
    if (!add_layernorm_supported(rows, channels)) return 0;
    return (size_t)grid_for(rows) * 2 * C * sizeof(float);
}

extern "C" int add_layernorm_forward_bf16(const void *a, const void *b, const void *gamma, const void *beta, long rows,
                                          int channels, float eps, void *y, float *mean, float *rstd, void *stream_)
{
    if (!add_layernorm_supported(rows, channels)) return MSDA_ERR_BAD_SHAPE;
    if (!a || !gamma || !beta || !y || !mean || !rstd) return MSDA_ERR_NULL_POINTER;
    if (misaligned(a) || misaligned(b) || misaligned(gamma) || misaligned(beta) || misaligned(y)) return MSDA_ERR_ALIGNMENT;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const dim3 grid(grid_for(rows)), block(THREADS);
    if (b)
        hipLaunchKernelGGL(forward_kernel<true>, grid, block, 0, stream, (const uint4 *)a, (const uint4 *)b,
                           (const uint4 *)gamma, (const uint4 *)beta, rows, eps, (uint4 *)y, mean, rstd);
    else
        hipLaunchKernelGGL(forward_kernel<false>, grid, block, 0, stream, (const uint4 *)a, (const uint4 *)nullptr,
                           (const uint4 *)gamma, (const uint4 *)beta, rows, eps, (uint4 *)y, mean, rstd);
    return hipGetLastError() == hipSuccess ? MSDA_OK : MSDA_ERR_LAUNCH;
}

extern "C" int add_layernorm_backward_bf16(const void *dy, const void *a, const void *b, const void *gamma,
                                           const float *mean, const float *rstd, long rows, int channels, void *dx,
                                           void *dgamma, void *dbeta, void *workspace, size_t workspace_bytes,
                                           void *stream_)
{
    if (!add_layernorm_supported(rows, channels)) return MSDA_ERR_BAD_SHAPE;
    if (!dy || !a || !gamma || !mean || !rstd || !dx || !dgamma || !dbeta || !workspace) return MSDA_ERR_NULL_POINTER;
    if (workspace_bytes < add_layernorm_workspace_bytes(rows, channels)) return MSDA_ERR_BAD_SHAPE;
    if (misaligned(dy) || misaligned(a) || misaligned(b) || misaligned(gamma) || misaligned(dx) || misaligned(workspace))
        return MSDA_ERR_ALIGNMENT;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int blocks = grid_for(rows);
    float *partial = static_cast<float *>(workspace);
    if (b)
        hipLaunchKernelGGL(backward_kernel<true>, dim3(blocks), dim3(THREADS), 0, stream, (const uint4 *)dy,
                           (const uint4 *)a, (const uint4 *)b, (const uint4 *)gamma, mean, rstd, rows, (uint4 *)dx, partial);
    else
        hipLaunchKernelGGL(backward_kernel<false>, dim3(blocks), dim3(THREADS), 0, stream, (const uint4 *)dy,
                           (const uint4 *)a, (const uint4 *)nullptr, (const uint4 *)gamma, mean, rstd, rows, (uint4 *)dx,
                           partial);
    hipLaunchKernelGGL(finish_param_grads, dim3(2 * C / 8), dim3(256), 0, stream, partial, blocks,
                       static_cast<uint16_t *>(dgamma), static_cast<uint16_t *>(dbeta));
    return hipGetLastError() == hipSuccess ? MSDA_OK : MSDA_ERR_LAUNCH;
}
