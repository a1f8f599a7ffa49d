// alif_attention.hip -- the bi-directional attention core of the ALIF fusion (RLIPv2_BiMultiHeadAttention, reference
// models/fuse_helper.py:365-466) as ONE kernel for gfx950: both directions share the logits q k^T; the vision side
// takes a softmax over the text tokens, the language side a softmax over the vision tokens of the transposed logits
// minus their row maximum (:399-400), each followed by attention dropout and its value product.
//
//   S   = Q K^T                      [Tv, Tl]   (Q already scaled and carrying the positional term)
//   P_v = softmax_j(S)               [Tv, Tl]   out_v = drop(P_v) V_l      [Tv, 256]
//   P_l = softmax_i(S^T)             [Tl, Tv]   out_l = drop(P_l) V_v      [Tl, 256]
//
// (The reference adds a constant 1.0 to every logit when its bool masks arrive, SURVEY.md Q1: softmax is unchanged.
//  The optional clamps / stable_softmax_2d are off in every RLIPv2 script; the host falls back to PyTorch for them.)
//
// One workgroup (4 waves) per (image, head): with `fusion_last_vis` the problem is Tv = 273 vision tokens of the last
// level x Tl = 64 label texts x head_dim 256 -- 27 MFLOP, launch-bound as ~10 PyTorch launches.  All three products
// run on v_mfma_f32_32x32x16_bf16 with operands loaded straight from global memory in fragment layout (16 B per
// lane): Q / K rows are k-contiguous as they come out of the projections, and the host hands the value projections
// over TRANSPOSED ([E, tokens], a GEMM with swapped operands), which makes them k-contiguous for P V as well -- no
// LDS transposes.  The logits live in LDS as float32 (both softmaxes read them, row-wise and column-wise), the
// probabilities as bfloat16 in the A-fragment layout of their product.  Probabilities BEFORE dropout are also written
// to global memory for the backward pass.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rlipv2_alif.h"
#include "once_per_device.h"
#include "../../include/rlipv2_msda.h"

// (tools/emu/ compiles this file for the CPU against a lane-level model of the workgroup and defines the macro itself)
#ifndef MSDA_DYNAMIC_LDS
#define MSDA_DYNAMIC_LDS(type, name) extern __shared__ __attribute__((aligned(16))) type name[]
#endif

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int HD = 256;                 // head dimension (embed_dim 2048 / 8 heads)
constexpr int TL = 64;                  // text tokens, padded
constexpr int TV_MAX = 288;             // vision tokens, padded to a multiple of 32
constexpr int THREADS = 256;
constexpr int S_STRIDE = TL + 1;        // floats per row of the logits (odd: a thread per row reads conflict-free)
constexpr int PV_STRIDE = TL + 8;       // bf16 per row of P_v  [Tv][Tl]
constexpr int PL_STRIDE = TV_MAX + 8;   // bf16 per row of P_l  [Tl][Tv]
constexpr int OFF_PV = TV_MAX * S_STRIDE * 4;
constexpr int OFF_PL = OFF_PV + TV_MAX * PV_STRIDE * 2;
constexpr int OFF_RED = OFF_PL + TL * PL_STRIDE * 2;          // softmax statistics: row max / 1/sum [2][TV_MAX], column max / 1/sum
constexpr int LDS_BYTES = OFF_RED + (2 * TV_MAX + 2 * TL + 8 * TL) * 4;  // [2][TL], partials [8][TL]
// operand staging (rows padded by 16 B so that the 32 rows a fragment read touches fall into different banks):
//   K   [64 rows][256 + 8] bf16 in the P_l region during phase 1 (P_l is written in phase 2)
//   V_l^T [256 rows][64 + 8] bf16 in the logits' region during phase 3 (the logits are dead after phase 2)
constexpr int K_STRIDE = HD + 8;
constexpr int VL_STRIDE = TL + 8;
static_assert(TL * K_STRIDE * 2 <= TL * PL_STRIDE * 2, "K staging fits the P_l region");
static_assert(HD * VL_STRIDE * 2 <= OFF_PV, "V_l^T staging fits the logits' region");

__device__ __forceinline__ float bf(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ uint16_t rne(float f)
{
    const uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

union Frag {
    uint4 u;
    bf16x8 v;
};

__device__ __forceinline__ Frag load_frag(const uint16_t *p, bool ok)
{
    Frag f;
    f.u = ok ? *reinterpret_cast<const uint4 *>(p) : make_uint4(0u, 0u, 0u, 0u);
    return f;
}

// C/D layout of v_mfma_f32_32x32x16: lane -> column (lane & 31), register r -> row (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

template <bool DROP>
__global__ __launch_bounds__(THREADS) void alif_forward_kernel(
    const uint16_t *__restrict__ q, const uint16_t *__restrict__ k, const uint16_t *__restrict__ vlT,
    const uint16_t *__restrict__ vvT, const uint8_t *__restrict__ keep_v, const uint8_t *__restrict__ keep_l,
    float keep_scale, int H, int Tv, int Tl, int Tvp, uint16_t *__restrict__ out_v, uint16_t *__restrict__ out_l,
    uint16_t *__restrict__ p_v, uint16_t *__restrict__ p_l)
{
    MSDA_DYNAMIC_LDS(unsigned char, lds);
    float *S = reinterpret_cast<float *>(lds);
    uint16_t *Pv = reinterpret_cast<uint16_t *>(lds + OFF_PV);
    uint16_t *Pl = reinterpret_cast<uint16_t *>(lds + OFF_PL);
    float *red = reinterpret_cast<float *>(lds + OFF_RED);

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int E = H * HD;
    const int li = lane & 31, kg = (lane >> 5) * 8;
    const int row_tiles = Tvp / 32;
    const uint16_t *qb = q + (size_t)b * Tv * E + h * HD;
    const uint16_t *kb = k + (size_t)b * Tl * E + h * HD;
    const uint16_t *vlb = vlT + ((size_t)b * E + h * HD) * TL;
    const uint16_t *vvb = vvT + ((size_t)b * E + h * HD) * Tvp;
    const size_t bh = (size_t)b * H + h;

    // ---- phase 1: S = Q K^T, a wave per 32-row tile, both 32-column tiles -----------------------------------------
    // K is staged in LDS once (coalesced 16-byte loads); a wave loads all 16 A fragments of its row tile in one go
    // (one memory latency per tile instead of one per k-step) and reads the B fragments from LDS.
    {
        uint16_t *Ks = Pl;
        for (int i = tid; i < TL * (HD / 8); i += THREADS) {
            const int j = i / (HD / 8), c8 = i % (HD / 8);
            const uint4 v = j < Tl ? *reinterpret_cast<const uint4 *>(kb + (size_t)j * E + c8 * 8) : make_uint4(0u, 0u, 0u, 0u);
            *reinterpret_cast<uint4 *>(Ks + j * K_STRIDE + c8 * 8) = v;
        }
        __syncthreads();
        for (int rt = wave; rt < row_tiles; rt += 4) {
            f32x16 acc[2];
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
            const int i = rt * 32 + li;
            Frag a[HD / 16];
#pragma unroll
            for (int ks = 0; ks < HD / 16; ++ks) a[ks] = load_frag(qb + (size_t)i * E + ks * 16 + kg, i < Tv);
#pragma unroll
            for (int ks = 0; ks < HD / 16; ++ks) {
                Frag b0, b1;
                b0.u = *reinterpret_cast<const uint4 *>(Ks + li * K_STRIDE + ks * 16 + kg);
                b1.u = *reinterpret_cast<const uint4 *>(Ks + (32 + li) * K_STRIDE + ks * 16 + kg);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks].v, b0.v, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks].v, b1.v, acc[1], 0, 0, 0);
            }
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) S[(rt * 32 + acc_row(r, lane)) * S_STRIDE + c * 32 + li] = acc[c][r];
        }
    }
    __syncthreads();

    // ---- phase 2: both softmaxes.  Statistics first (row maxima / sums by a thread per vision row, column maxima / sums
    // by 4 threads per text token -- serial passes over LDS, no shuffles), then two element-wise sweeps in which every
    // global access (saved probabilities, keep masks) is a wave reading / writing consecutive addresses ---------------
    float *rmax = red, *rinv = red + TV_MAX, *cmax = red + 2 * TV_MAX, *cinv = cmax + TL, *part = cinv + TL;
    for (int i = tid; i < Tv; i += THREADS) {
        const float *srow = S + i * S_STRIDE;
        float mx = -INFINITY;
        for (int j = 0; j < Tl; ++j) mx = fmaxf(mx, srow[j]);
        float sum = 0.f;
        for (int j = 0; j < Tl; ++j) sum += __expf(srow[j] - mx);
        rmax[i] = mx;
        rinv[i] = 1.f / sum;
    }
    {
        const int j = tid & 63, pt = tid >> 6;
        float mx = -INFINITY;
        if (j < Tl)
            for (int i = pt; i < Tv; i += 4) mx = fmaxf(mx, S[i * S_STRIDE + j]);
        part[pt * TL + j] = mx;
        __syncthreads();
        mx = fmaxf(fmaxf(part[j], part[TL + j]), fmaxf(part[2 * TL + j], part[3 * TL + j]));
        float sum = 0.f;
        if (j < Tl)
            for (int i = pt; i < Tv; i += 4) sum += __expf(S[i * S_STRIDE + j] - mx);
        part[(4 + pt) * TL + j] = sum;
        __syncthreads();
        if (pt == 0) {
            cmax[j] = mx;
            cinv[j] = 1.f / (part[4 * TL + j] + part[5 * TL + j] + part[6 * TL + j] + part[7 * TL + j]);
        }
    }
    __syncthreads();
    // (the keep-mask bytes of a batch of elements are requested together, ahead of their use: inside a conditional a
    //  load is not hoisted by the compiler, and a serial loop of dependent ~1 us global loads is what made the first
    //  version of this phase take ~70 us)
    {
        constexpr int BATCH = 8;
        const int j = tid & 63;
        for (int i0 = tid >> 6; i0 < Tvp; i0 += 4 * BATCH) {        // P_v, [vision][text] order: a wave per row
            uint8_t kb[BATCH];
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int i = i0 + 4 * u;
                kb[u] = (DROP && i < Tv && j < Tl) ? keep_v[(bh * Tv + i) * Tl + j] : (uint8_t)1;
            }
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int i = i0 + 4 * u;
                if (i >= Tvp) break;
                uint16_t pb = 0;
                if (i < Tv && j < Tl) {
                    pb = rne(__expf(S[i * S_STRIDE + j] - rmax[i]) * rinv[i]);   // the probability as bfloat16 (what the backward sees)
                    p_v[(bh * Tv + i) * Tl + j] = pb;
                    if (DROP) pb = kb[u] ? rne(bf(pb) * keep_scale) : (uint16_t)0;
                }
                Pv[i * PV_STRIDE + j] = pb;
            }
        }
    }
    for (int j = wave; j < TL; j += THREADS / 64) {                 // P_l, [text][vision] order: a wave per text token
        uint16_t *prow = Pl + j * PL_STRIDE;
        const float mx = cmax[j], inv = cinv[j];
        constexpr int STEPS = (TV_MAX + 63) / 64;
        uint8_t kb[STEPS];
#pragma unroll
        for (int u = 0; u < STEPS; ++u) {
            const int i = lane + 64 * u;
            kb[u] = (DROP && i < Tv && j < Tl) ? keep_l[(bh * Tl + j) * Tv + i] : (uint8_t)1;
        }
#pragma unroll
        for (int u = 0; u < STEPS; ++u) {
            const int i = lane + 64 * u;
            if (i >= Tvp) break;
            uint16_t pb = 0;
            if (i < Tv && j < Tl) {
                pb = rne(__expf(S[i * S_STRIDE + j] - mx) * inv);       // (bank (i + j) mod 64: conflict-free)
                p_l[(bh * Tl + j) * Tv + i] = pb;
                if (DROP) pb = kb[u] ? rne(bf(pb) * keep_scale) : (uint16_t)0;
            }
            prow[i] = pb;
        }
    }
    __syncthreads();

    // ---- phase 3: out_v = P_v V_l  (K = 64 text tokens), a wave per 32-row tile, 8 column tiles of 32 channels ------
    // V_l^T (256 channel rows x 64 tokens, 32 KB) is staged over the dead logits; phase 4's first B fragments are
    // requested before, so that they travel underneath this phase.
    {
        uint16_t *Vs = reinterpret_cast<uint16_t *>(lds);
        for (int i = tid; i < HD * (TL / 8); i += THREADS) {
            const int c = i / (TL / 8), j8 = i % (TL / 8);
            *reinterpret_cast<uint4 *>(Vs + c * VL_STRIDE + j8 * 8) = *reinterpret_cast<const uint4 *>(vlb + (size_t)c * TL + j8 * 8);
        }
        __syncthreads();
        for (int rt = wave; rt < row_tiles; rt += 4) {
            Frag a[TL / 16];
#pragma unroll
            for (int ks = 0; ks < TL / 16; ++ks)
                a[ks].u = *reinterpret_cast<const uint4 *>(Pv + (rt * 32 + li) * PV_STRIDE + ks * 16 + kg);
#pragma unroll 2
            for (int nt = 0; nt < HD / 32; ++nt) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < TL / 16; ++ks) {
                    Frag bf_;
                    bf_.u = *reinterpret_cast<const uint4 *>(Vs + (nt * 32 + li) * VL_STRIDE + ks * 16 + kg);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks].v, bf_.v, acc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int i = rt * 32 + acc_row(r, lane);
                    if (i < Tv) out_v[((size_t)b * Tv + i) * E + h * HD + nt * 32 + li] = rne(acc[r]);
                }
            }
        }
    }
    // ---- phase 4: out_l = P_l V_v  (K = Tv vision tokens), wave w: channels [64 w, 64 w + 64), both text row tiles -----
    {
        f32x16 acc[2][2];
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[jt][c][r] = 0.f;
        const uint16_t *v0 = vvb + (size_t)(wave * 64 + li) * Tvp, *v1 = v0 + (size_t)32 * Tvp;
#pragma unroll 6
        for (int ks = 0; ks < Tvp / 16; ++ks) {
            Frag a0, a1;
            a0.u = *reinterpret_cast<const uint4 *>(Pl + li * PL_STRIDE + ks * 16 + kg);
            a1.u = *reinterpret_cast<const uint4 *>(Pl + (32 + li) * PL_STRIDE + ks * 16 + kg);
            const Frag b0 = load_frag(v0 + ks * 16 + kg, true), b1 = load_frag(v1 + ks * 16 + kg, true);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0.v, b0.v, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0.v, b1.v, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.v, b0.v, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.v, b1.v, acc[1][1], 0, 0, 0);
        }
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int j = jt * 32 + acc_row(r, lane);
                    if (j < Tl) out_l[((size_t)b * Tl + j) * E + h * HD + wave * 64 + c * 32 + li] = rne(acc[jt][c][r]);
                }
    }
}

// Backward of the two softmaxes (and of the dropout in front of the value products): from the saved probabilities and
// the gradients of the (dropped) probabilities to the gradient of the shared logits,
//   dS[i][j] = P_v[i][j] (g_v[i][j] - sum_j' P_v[i][j'] g_v[i][j']) + P_l[j][i] (g_l[j][i] - sum_i' P_l[j][i'] g_l[j][i'])
// with g = dP * keep * keep_scale; also writes the dropped probabilities P * keep * keep_scale the value-projection
// gradients need.  One workgroup per (image, head), float32 arithmetic, ~12 PyTorch launches otherwise.
constexpr int BWD_THREADS = 1024;

__device__ __forceinline__ float wave_sum64(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <bool DROP>
__global__ __launch_bounds__(BWD_THREADS) void alif_softmax_backward_kernel(
    const uint16_t *__restrict__ p_v, const uint16_t *__restrict__ p_l, const uint16_t *__restrict__ d_pv,
    const uint16_t *__restrict__ d_pl, const uint8_t *__restrict__ keep_v, const uint8_t *__restrict__ keep_l,
    float keep_scale, int Tv, int Tl, uint16_t *__restrict__ d_s, uint16_t *__restrict__ pd_v, uint16_t *__restrict__ pd_l)
{
    // every global access is a wave reading / writing consecutive elements; the language-side term, which lives in
    // [text][vision] order, reaches the [vision][text] output through an LDS tile
    __shared__ float row_v[TV_MAX];              // sum_j P_v g_v per vision row
    __shared__ float row_l[TL];                  // sum_i P_l g_l per text row
    __shared__ float term_l[TL][TV_MAX + 1];     // P_l (g_l - row_l) in [text][vision] order
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    constexpr int WAVES = BWD_THREADS / 64;
    const size_t base = (size_t)blockIdx.x * Tv * Tl;
    const uint16_t *pv = p_v + base, *pl = p_l + base, *gv = d_pv + base, *gl = d_pl + base;
    const uint8_t *kv = DROP ? keep_v + base : nullptr, *kl = DROP ? keep_l + base : nullptr;
    auto grad = [&](const uint16_t *g, const uint8_t *kp, int e) {
        const float x = bf(g[e]);
        return DROP ? (kp[e] ? x * keep_scale : 0.f) : x;
    };
    for (int i = wave; i < Tv; i += WAVES) {                     // vision rows: lane = text token
        const int e = i * Tl + lane;
        const float acc = wave_sum64(lane < Tl ? bf(pv[e]) * grad(gv, kv, e) : 0.f);
        if (lane == 0) row_v[i] = acc;
    }
    for (int j = wave; j < Tl; j += WAVES) {                     // text rows: lanes stride over the vision tokens
        float acc = 0.f;
        for (int i = lane; i < Tv; i += 64) acc = fmaf(bf(pl[j * Tv + i]), grad(gl, kl, j * Tv + i), acc);
        acc = wave_sum64(acc);
        if (lane == 0) row_l[j] = acc;
    }
    __syncthreads();
    for (int e = tid; e < Tv * Tl; e += BWD_THREADS) {           // language-side term, [text][vision] order
        const int j = e / Tv, i = e % Tv;
        const float plv = bf(pl[e]);
        term_l[j][i] = plv * (grad(gl, kl, e) - row_l[j]);
        if (DROP) pd_l[base + e] = kl[e] ? rne(plv * keep_scale) : (uint16_t)0;
    }
    __syncthreads();
    for (int e = tid; e < Tv * Tl; e += BWD_THREADS) {           // [vision][text] order
        const int i = e / Tl, j = e % Tl;
        const float pvv = bf(pv[e]);
        d_s[base + e] = rne(pvv * (grad(gv, kv, e) - row_v[i]) + term_l[j][i]);
        if (DROP) pd_v[base + e] = kv[e] ? rne(pvv * keep_scale) : (uint16_t)0;
    }
}

bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

extern "C" {

int alif_attention_supported(int B, int H, int Tv, int Tl, int head_dim)
{
    return B > 0 && H > 0 && head_dim == HD && Tl >= 1 && Tl <= TL && Tv >= 1 && Tv <= TV_MAX &&
           (long)B * H < (1L << 20);
}

int alif_attention_padded_tv(int Tv) { return (Tv + 31) / 32 * 32; }

int alif_attention_forward_bf16(const void *q, const void *k, const void *values_l_t, const void *values_v_t,
                                const void *keep_v, const void *keep_l, float keep_scale, int B, int H, int Tv, int Tl,
                                void *out_v, void *out_l, void *probs_v, void *probs_l, void *stream)
{
    if (!alif_attention_supported(B, H, Tv, Tl, HD)) return MSDA_ERR_BAD_SHAPE;
    if (!q || !k || !values_l_t || !values_v_t || !out_v || !out_l || !probs_v || !probs_l) return MSDA_ERR_NULL_POINTER;
    if ((keep_v == nullptr) != (keep_l == nullptr)) return MSDA_ERR_NULL_POINTER;
    if (!(aligned16(q) && aligned16(k) && aligned16(values_l_t) && aligned16(values_v_t))) return MSDA_ERR_ALIGNMENT;
    const int Tvp = alif_attention_padded_tv(Tv);
    hipStream_t s = (hipStream_t)stream;
    RLIPV2_ONCE_PER_DEVICE(      // more than 64 KB of dynamic LDS has to be asked for
        (void)hipFuncSetAttribute((const void *)alif_forward_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute((const void *)alif_forward_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    (void)hipGetLastError();
    if (keep_v)
        hipLaunchKernelGGL(alif_forward_kernel<true>, dim3(B * H), dim3(THREADS), LDS_BYTES, s, (const uint16_t *)q,
                           (const uint16_t *)k, (const uint16_t *)values_l_t, (const uint16_t *)values_v_t,
                           (const uint8_t *)keep_v, (const uint8_t *)keep_l, keep_scale, H, Tv, Tl, Tvp, (uint16_t *)out_v,
                           (uint16_t *)out_l, (uint16_t *)probs_v, (uint16_t *)probs_l);
    else
        hipLaunchKernelGGL(alif_forward_kernel<false>, dim3(B * H), dim3(THREADS), LDS_BYTES, s, (const uint16_t *)q,
                           (const uint16_t *)k, (const uint16_t *)values_l_t, (const uint16_t *)values_v_t,
                           (const uint8_t *)nullptr, (const uint8_t *)nullptr, 1.f, H, Tv, Tl, Tvp, (uint16_t *)out_v,
                           (uint16_t *)out_l, (uint16_t *)probs_v, (uint16_t *)probs_l);
    return hipGetLastError() == hipSuccess ? MSDA_OK : MSDA_ERR_LAUNCH;
}

}  // extern "C"

extern "C" int alif_attention_softmax_backward_bf16(const void *probs_v, const void *probs_l, const void *d_probs_v,
                                                    const void *d_probs_l, const void *keep_v, const void *keep_l,
                                                    float keep_scale, int B, int H, int Tv, int Tl, void *d_logits,
                                                    void *dropped_v, void *dropped_l, void *stream)
{
    if (!alif_attention_supported(B, H, Tv, Tl, HD)) return MSDA_ERR_BAD_SHAPE;
    if (!probs_v || !probs_l || !d_probs_v || !d_probs_l || !d_logits) return MSDA_ERR_NULL_POINTER;
    if ((keep_v == nullptr) != (keep_l == nullptr)) return MSDA_ERR_NULL_POINTER;
    if (keep_v && (!dropped_v || !dropped_l)) return MSDA_ERR_NULL_POINTER;
    hipStream_t s = (hipStream_t)stream;
    (void)hipGetLastError();
    if (keep_v)
        hipLaunchKernelGGL(alif_softmax_backward_kernel<true>, dim3(B * H), dim3(BWD_THREADS), 0, s, (const uint16_t *)probs_v,
                           (const uint16_t *)probs_l, (const uint16_t *)d_probs_v, (const uint16_t *)d_probs_l,
                           (const uint8_t *)keep_v, (const uint8_t *)keep_l, keep_scale, Tv, Tl, (uint16_t *)d_logits,
                           (uint16_t *)dropped_v, (uint16_t *)dropped_l);
    else
        hipLaunchKernelGGL(alif_softmax_backward_kernel<false>, dim3(B * H), dim3(BWD_THREADS), 0, s, (const uint16_t *)probs_v,
                           (const uint16_t *)probs_l, (const uint16_t *)d_probs_v, (const uint16_t *)d_probs_l,
                           (const uint8_t *)nullptr, (const uint8_t *)nullptr, 1.f, Tv, Tl, (uint16_t *)d_logits,
                           (uint16_t *)nullptr, (uint16_t *)nullptr);
    return hipGetLastError() == hipSuccess ? MSDA_OK : MSDA_ERR_LAUNCH;
}
