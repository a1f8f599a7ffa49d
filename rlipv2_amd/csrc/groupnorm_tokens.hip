// groupnorm_tokens.hip -- GroupNorm(32, 256) of the token-major feature pyramid (include/rlipv2_groupnorm.h).
//
// HBM-bound elementwise / reduction work: forward reads x twice (statistics, apply) and writes out once, backward reads
// dy and x twice and writes dx once; 45.5 MB per pass at batch 4 of the 800x1333 pyramid.  A group is 8 adjacent channels
// = one 16-byte load per (token, group); a thread owns one group and every 8th token of its workgroup's 256 tokens, so a
// wave reads 2 tokens x 512 contiguous bytes per instruction.  Statistics are combined with Chan's formula (per-chunk
// mean / M2, double precision for the handful of chunk terms) in a fixed order: bit-for-bit repeatable, no atomics.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rlipv2_groupnorm.h"
#include "../../include/rlipv2_msda.h"

namespace {

constexpr int C = 256, G = 32, THREADS = 256, TOK = 256, MAXL = 4, TL = THREADS / G;   // TL = 8 token lanes

struct Plan {
    const uint16_t *x[MAXL];
    const uint16_t *gamma[MAXL];
    const uint16_t *beta[MAXL];
    uint16_t *dx[MAXL];
    uint16_t *dgamma[MAXL];
    uint16_t *dbeta[MAXL];
    int hw[MAXL], start[MAXL], cb[MAXL + 1];      // cb: first chunk of the level inside an image's chunk list
    int L, N, S, CT;                              // CT = chunks per image
};

struct Where { int n, l, k, t0, t1, nck; };

__device__ __forceinline__ Where decode(const Plan &p, int b)
{
    Where w;
    w.n = b / p.CT;
    const int c = b % p.CT;
    w.l = 0;
#pragma unroll
    for (int l = 1; l < MAXL; ++l) w.l = (l < p.L && c >= p.cb[l]) ? l : w.l;
    w.k = c - p.cb[w.l];
    w.nck = p.cb[w.l + 1] - p.cb[w.l];
    w.t0 = w.k * TOK;
    w.t1 = min(p.hw[w.l], w.t0 + TOK);
    return w;
}

__device__ __forceinline__ void unpack8(const uint4 &v, float (&f)[8])
{
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f[2 * j] = __uint_as_float(w[j] << 16);
        f[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u);
    }
}

__device__ __forceinline__ uint32_t rne(float f)
{
    const uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;     // NaN stays NaN
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

__device__ __forceinline__ uint4 pack8(const float (&f)[8])
{
    uint4 v;
    v.x = rne(f[0]) | (rne(f[1]) << 16); v.y = rne(f[2]) | (rne(f[3]) << 16);
    v.z = rne(f[4]) | (rne(f[5]) << 16); v.w = rne(f[6]) | (rne(f[7]) << 16);
    return v;
}

// ---- forward, pass 1: per (image, chunk, group) mean and M2 of the chunk's tokens x 8 channels -------------------
__global__ __launch_bounds__(THREADS) void stats_kernel(Plan p, float2 *__restrict__ part)
{
    __shared__ float2 red[TL][G];
    const Where w = decode(p, blockIdx.x);
    const int g = threadIdx.x & (G - 1), tl = threadIdx.x / G;
    const uint16_t *xp = p.x[w.l] + ((size_t)w.n * p.hw[w.l]) * C + g * 8;
    float s = 0.f, ss = 0.f;
    for (int t = w.t0 + tl; t < w.t1; t += TL) {
        float f[8];
        unpack8(*reinterpret_cast<const uint4 *>(xp + (size_t)t * C), f);
#pragma unroll
        for (int j = 0; j < 8; ++j) { s += f[j]; ss = fmaf(f[j], f[j], ss); }
    }
    red[tl][g] = make_float2(s, ss);
    __syncthreads();
    if (tl == 0) {
#pragma unroll
        for (int j = 1; j < TL; ++j) { s += red[j][g].x; ss += red[j][g].y; }
        const float cnt = (float)((w.t1 - w.t0) * 8);
        const float mean = s / cnt;
        part[(size_t)blockIdx.x * G + g] = make_float2(mean, fmaxf(ss - s * mean, 0.f));
    }
}

// the (image, level)'s statistics from its chunks' (mean, M2): every workgroup of the level recomputes them (<= 66 terms)
__device__ __forceinline__ void combine_stats(const Plan &p, const Where &w, const float2 *__restrict__ part, float eps,
                                              float (*stat)[2], double (*scratch)[G][3])
{
    const int g = threadIdx.x & (G - 1), tl = threadIdx.x / G;
    const float2 *pp = part + ((size_t)w.n * p.CT + p.cb[w.l]) * G + g;
    double cnt = 0.0, mean = 0.0, m2 = 0.0;
    for (int k = tl; k < w.nck; k += TL) {
        const float2 v = pp[(size_t)k * G];
        const double c = (double)((min(p.hw[w.l], (k + 1) * TOK) - k * TOK) * 8);
        const double d = (double)v.x - mean, tot = cnt + c;
        mean += d * c / tot;
        m2 += (double)v.y + d * d * cnt * c / tot;
        cnt = tot;
    }
    scratch[tl][g][0] = cnt; scratch[tl][g][1] = mean; scratch[tl][g][2] = m2;
    __syncthreads();
    if (tl == 0) {
        for (int j = 1; j < TL; ++j) {
            const double c = scratch[j][g][0];
            if (c > 0.0) {
                const double d = scratch[j][g][1] - mean, tot = cnt + c;
                mean += d * c / tot;
                m2 += scratch[j][g][2] + d * d * cnt * c / tot;
                cnt = tot;
            }
        }
        stat[g][0] = (float)mean;
        stat[g][1] = (float)(1.0 / sqrt(m2 / cnt + (double)eps));
    }
    __syncthreads();
}

// ---- forward, pass 2 ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(THREADS) void apply_kernel(Plan p, const float2 *__restrict__ part, float eps,
                                                        uint16_t *__restrict__ out, float *__restrict__ mean_out,
                                                        float *__restrict__ rstd_out)
{
    __shared__ float stat[G][2];
    __shared__ double scratch[TL][G][3];
    const Where w = decode(p, blockIdx.x);
    combine_stats(p, w, part, eps, stat, scratch);
    const int g = threadIdx.x & (G - 1), tl = threadIdx.x / G;
    const float mean = stat[g][0], rstd = stat[g][1];
    if (w.k == 0 && tl == 0) {
        mean_out[((size_t)w.l * p.N + w.n) * G + g] = mean;
        rstd_out[((size_t)w.l * p.N + w.n) * G + g] = rstd;
    }
    float ga[8], be[8];
    unpack8(*reinterpret_cast<const uint4 *>(p.gamma[w.l] + g * 8), ga);
    unpack8(*reinterpret_cast<const uint4 *>(p.beta[w.l] + g * 8), be);
    const uint16_t *xp = p.x[w.l] + ((size_t)w.n * p.hw[w.l]) * C + g * 8;
    uint16_t *op = out + ((size_t)w.n * p.S + p.start[w.l]) * C + g * 8;
    for (int t = w.t0 + tl; t < w.t1; t += TL) {
        float f[8];
        unpack8(*reinterpret_cast<const uint4 *>(xp + (size_t)t * C), f);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = fmaf((f[j] - mean) * rstd, ga[j], be[j]);
        *reinterpret_cast<uint4 *>(op + (size_t)t * C) = pack8(f);
    }
}

// ---- backward, pass 1: per (image, chunk, channel) sum dy * x and sum dy ---------------------------------------------
__global__ __launch_bounds__(THREADS) void bstats_kernel(Plan p, const uint16_t *__restrict__ dy,
                                                         float2 *__restrict__ part)
{
    __shared__ float2 red[TL][C];
    const Where w = decode(p, blockIdx.x);
    const int g = threadIdx.x & (G - 1), tl = threadIdx.x / G;
    const uint16_t *xp = p.x[w.l] + ((size_t)w.n * p.hw[w.l]) * C + g * 8;
    const uint16_t *dp = dy + ((size_t)w.n * p.S + p.start[w.l]) * C + g * 8;
    float ds[8], db[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { ds[j] = 0.f; db[j] = 0.f; }
    for (int t = w.t0 + tl; t < w.t1; t += TL) {
        float f[8], d[8];
        unpack8(*reinterpret_cast<const uint4 *>(xp + (size_t)t * C), f);
        unpack8(*reinterpret_cast<const uint4 *>(dp + (size_t)t * C), d);
#pragma unroll
        for (int j = 0; j < 8; ++j) { ds[j] = fmaf(d[j], f[j], ds[j]); db[j] += d[j]; }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[tl][g * 8 + j] = make_float2(ds[j], db[j]);
    __syncthreads();
    const int ch = threadIdx.x;
    float2 s = red[0][ch];
#pragma unroll
    for (int j = 1; j < TL; ++j) { s.x += red[j][ch].x; s.y += red[j][ch].y; }
    part[(size_t)blockIdx.x * C + ch] = s;
}

// ---- backward, pass 2: one workgroup per (image, level): the level's channel sums, the groups' dx coefficients, the
// image's share of dgamma / dbeta ------------------------------------------------------------------------------------
__global__ __launch_bounds__(THREADS) void bfinal_kernel(Plan p, const float2 *__restrict__ part,
                                                         const float *__restrict__ mean, const float *__restrict__ rstd,
                                                         float2 *__restrict__ coef, float2 *__restrict__ dgp)
{
    const int n = blockIdx.x / p.L, l = blockIdx.x % p.L, ch = threadIdx.x, g = ch >> 3;
    const int nck = p.cb[l + 1] - p.cb[l];
    const float2 *pp = part + ((size_t)n * p.CT + p.cb[l]) * C + ch;
    float ds = 0.f, db = 0.f;
#pragma unroll 8
    for (int k = 0; k < nck; ++k) {
        const float2 v = pp[(size_t)k * C];
        ds += v.x; db += v.y;
    }
    const uint32_t gw = reinterpret_cast<const uint16_t *>(p.gamma[l])[ch];
    const float ga = __uint_as_float(gw << 16);
    float dsg = ds * ga, dbg = db * ga;
#pragma unroll
    for (int off = 1; off < 8; off <<= 1) {
        dsg += __shfl_xor(dsg, off, 64);
        dbg += __shfl_xor(dbg, off, 64);
    }
    const float mu = mean[((size_t)l * p.N + n) * G + g], rs = rstd[((size_t)l * p.N + n) * G + g];
    if ((ch & 7) == 0) {
        const float s = 1.f / ((float)p.hw[l] * 8.f);
        const float f1 = (dbg * mu - dsg) * rs * rs * rs * s;
        const float f2 = -f1 * mu - dbg * rs * s;
        coef[((size_t)n * p.L + l) * G + g] = make_float2(f1, f2);
    }
    dgp[((size_t)n * p.L + l) * C + ch] = make_float2((ds - mu * db) * rs, db);
}

// ---- backward, pass 3 -----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(THREADS) void bapply_kernel(Plan p, const uint16_t *__restrict__ dy,
                                                         const float *__restrict__ rstd, const float2 *__restrict__ coef,
                                                         const float2 *__restrict__ dgp)
{
    const Where w = decode(p, blockIdx.x);
    const int g = threadIdx.x & (G - 1), tl = threadIdx.x / G;
    if (w.n == 0 && w.k == 0) {             // parameter gradients of the level: sum over the images, in order
        const int ch = threadIdx.x;
        float a = 0.f, b = 0.f;
        for (int n = 0; n < p.N; ++n) {
            const float2 v = dgp[((size_t)n * p.L + w.l) * C + ch];
            a += v.x; b += v.y;
        }
        p.dgamma[w.l][ch] = (uint16_t)rne(a);
        p.dbeta[w.l][ch] = (uint16_t)rne(b);
    }
    const float rs = rstd[((size_t)w.l * p.N + w.n) * G + g];
    const float2 f = coef[((size_t)w.n * p.L + w.l) * G + g];
    float ga[8];
    unpack8(*reinterpret_cast<const uint4 *>(p.gamma[w.l] + g * 8), ga);
#pragma unroll
    for (int j = 0; j < 8; ++j) ga[j] *= rs;
    const uint16_t *xp = p.x[w.l] + ((size_t)w.n * p.hw[w.l]) * C + g * 8;
    const uint16_t *dp = dy + ((size_t)w.n * p.S + p.start[w.l]) * C + g * 8;
    uint16_t *op = p.dx[w.l] + ((size_t)w.n * p.hw[w.l]) * C + g * 8;
    for (int t = w.t0 + tl; t < w.t1; t += TL) {
        float x[8], d[8];
        unpack8(*reinterpret_cast<const uint4 *>(xp + (size_t)t * C), x);
        unpack8(*reinterpret_cast<const uint4 *>(dp + (size_t)t * C), d);
#pragma unroll
        for (int j = 0; j < 8; ++j) d[j] = fmaf(ga[j], d[j], fmaf(f.x, x[j], f.y));
        *reinterpret_cast<uint4 *>(op + (size_t)t * C) = pack8(d);
    }
}

bool make_plan(Plan &p, int N, const int *hw, int levels)
{
    if (N < 1 || levels < 1 || levels > MAXL || !hw) return false;
    p = Plan();
    p.L = levels; p.N = N;
    int s = 0, c = 0;
    for (int l = 0; l < levels; ++l) {
        if (hw[l] < 1) return false;
        p.hw[l] = hw[l]; p.start[l] = s; p.cb[l] = c;
        s += hw[l];
        c += (hw[l] + TOK - 1) / TOK;
    }
    for (int l = levels; l <= MAXL; ++l) p.cb[l] = c;
    p.S = s; p.CT = c;
    return true;
}

// workspace layout (floats): part [N][CT][C][2] (the forward uses [N][CT][G][2] of it) | coef [N][L][G][2] | dgp [N][L][C][2]
size_t part_floats(const Plan &p) { return (size_t)p.N * p.CT * C * 2; }
size_t coef_floats(const Plan &p) { return (size_t)p.N * p.L * G * 2; }
size_t dgp_floats(const Plan &p) { return (size_t)p.N * p.L * C * 2; }

bool misaligned(const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15u) != 0; }

}  // namespace

extern "C" int groupnorm_tokens_supported(int channels, int groups, int levels)
{
    return channels == C && groups == G && levels >= 1 && levels <= MAXL;
}

extern "C" size_t groupnorm_tokens_workspace_bytes(int N, const int *hw, int levels)
{
    Plan p;
    if (!make_plan(p, N, hw, levels)) return 0;
    return (part_floats(p) + coef_floats(p) + dgp_floats(p)) * sizeof(float);
}

extern "C" int groupnorm_tokens_forward_bf16(const void *const *x, const int *hw, int levels, int N,
                                             const void *const *gamma, const void *const *beta, float eps, void *out,
                                             float *mean, float *rstd, void *workspace, size_t workspace_bytes,
                                             void *stream_)
{
    Plan p;
    if (!make_plan(p, N, hw, levels)) return MSDA_ERR_BAD_SHAPE;
    if (!x || !gamma || !beta || !out || !mean || !rstd || !workspace) return MSDA_ERR_NULL_POINTER;
    if (workspace_bytes < groupnorm_tokens_workspace_bytes(N, hw, levels)) return MSDA_ERR_BAD_SHAPE;
    if (misaligned(out) || misaligned(workspace)) return MSDA_ERR_ALIGNMENT;
    for (int l = 0; l < levels; ++l) {
        if (!x[l] || !gamma[l] || !beta[l]) return MSDA_ERR_NULL_POINTER;
        if (misaligned(x[l]) || misaligned(gamma[l]) || misaligned(beta[l])) return MSDA_ERR_ALIGNMENT;
        p.x[l] = static_cast<const uint16_t *>(x[l]);
        p.gamma[l] = static_cast<const uint16_t *>(gamma[l]);
        p.beta[l] = static_cast<const uint16_t *>(beta[l]);
    }
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    float2 *part = static_cast<float2 *>(workspace);
    const dim3 grid(N * p.CT), block(THREADS);
    hipLaunchKernelGGL(stats_kernel, grid, block, 0, stream, p, part);
    hipLaunchKernelGGL(apply_kernel, grid, block, 0, stream, p, part, eps, static_cast<uint16_t *>(out), mean, rstd);
    return hipGetLastError() == hipSuccess ? MSDA_OK : MSDA_ERR_LAUNCH;
}

extern "C" int groupnorm_tokens_backward_bf16(const void *dy, const void *const *x, const int *hw, int levels, int N,
                                              const void *const *gamma, const float *mean, const float *rstd,
                                              void *const *dx, void *const *dgamma, void *const *dbeta, void *workspace,
                                              size_t workspace_bytes, void *stream_)
{
    Plan p;
    if (!make_plan(p, N, hw, levels)) return MSDA_ERR_BAD_SHAPE;
    if (!dy || !x || !gamma || !mean || !rstd || !dx || !dgamma || !dbeta || !workspace) return MSDA_ERR_NULL_POINTER;
    if (workspace_bytes < groupnorm_tokens_workspace_bytes(N, hw, levels)) return MSDA_ERR_BAD_SHAPE;
    if (misaligned(dy) || misaligned(workspace)) return MSDA_ERR_ALIGNMENT;
    for (int l = 0; l < levels; ++l) {
        if (!x[l] || !gamma[l] || !dx[l] || !dgamma[l] || !dbeta[l]) return MSDA_ERR_NULL_POINTER;
        if (misaligned(x[l]) || misaligned(gamma[l]) || misaligned(dx[l])) return MSDA_ERR_ALIGNMENT;
        p.x[l] = static_cast<const uint16_t *>(x[l]);
        p.gamma[l] = static_cast<const uint16_t *>(gamma[l]);
        p.dx[l] = static_cast<uint16_t *>(dx[l]);
        p.dgamma[l] = static_cast<uint16_t *>(dgamma[l]);
        p.dbeta[l] = static_cast<uint16_t *>(dbeta[l]);
    }
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    float *ws = static_cast<float *>(workspace);
    float2 *part = reinterpret_cast<float2 *>(ws);
    float2 *coef = reinterpret_cast<float2 *>(ws + part_floats(p));
    float2 *dgp = reinterpret_cast<float2 *>(ws + part_floats(p) + coef_floats(p));
    const dim3 grid(N * p.CT), block(THREADS);
    const uint16_t *dy16 = static_cast<const uint16_t *>(dy);
    hipLaunchKernelGGL(bstats_kernel, grid, block, 0, stream, p, dy16, part);
    hipLaunchKernelGGL(bfinal_kernel, dim3(N * levels), block, 0, stream, p, part, mean, rstd, coef, dgp);
    hipLaunchKernelGGL(bapply_kernel, grid, block, 0, stream, p, dy16, rstd, coef, dgp);
    return hipGetLastError() == hipSuccess ? MSDA_OK : MSDA_ERR_LAUNCH;
}
