// token_gemm.hip -- weight / bias gradient of a Linear applied to token-major activations
// (include/rlipv2_linear.h):  dW[M,K] = dY[T,M]^T X[T,K],  db[M] = sum_t dY[t,:],  bf16 in, f32 accumulate.
//
// Shape: T = 88 892 tokens, M, K in {256, 384, 1024}.  Arithmetic intensity M*K/(M+K) flop/byte = 128..205,
// below the MI355X ridge (2.5 PFLOP/s / 8 TB/s = 312), so the bound is HBM: dY and X are streamed once.
//
// One workgroup (256 threads = 4 waves, 2x2 waves of 64x64 outputs = one 128x128 tile of dW) walks a chunk
// of consecutive tokens 32 rows at a time:
//   * global -> LDS by LDS-DMA (global_load_lds_dwordx4): no staging registers and no ds_write pass (a
//     ds_write_b128 costs 13 LDS-path cycles per wave; the register-staged first version of this kernel
//     spent more time storing tiles than the MFMAs spent consuming them).  One DMA instruction fills one
//     1 KB "row group" = 4 token rows x 128 columns; three LDS stages, DMAs issued two steps ahead and
//     waited for with a counted vmcnt, one raw s_barrier per step;
//   * both MFMA operands need 8 consecutive-k (= token) values of one column per lane, the transpose of
//     what memory holds; ds_read_b64_tr_b16 delivers that: a 16-lane group reads a [4 tokens][16 columns]
//     block and every lane receives one column.  dY and X are read with the same lane->token map, so the
//     k order inside an MFMA is consistent by construction;
//   * the DMA's LDS side is lane-linear, so the bank skew is applied on the SOURCE side: 16-byte piece c of
//     row r of a group is stored at slot 16 r + (c ^ 4 r); the 4 rows x 4 pieces a 32-lane half reads with
//     one transpose-read then cover all 64 banks exactly once;
//   * v_mfma_f32_32x32x16_bf16, 2x2 per wave per 16 tokens, 64 accumulator VGPRs;
//   * the per-chunk 128x128 partial goes to the workspace with plain stores and `reduce_partials` sums the
//     chunks (fp32 atomics to one shared dW measured 3x slower, profiles/r01_ubench_global_atomics.txt);
//     the same pass adds the T % 32 tail rows, so the main loop has no partial step;
//   * bias gradient: the tile_k == 0 workgroups also add up the dY tile they have in LDS anyway.
// Workgroups that share a chunk are placed on the same XCD (they re-read the chunk through that XCD's L2).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/rlipv2_linear.h"
#include "../../include/rlipv2_msda.h"
#include "msda_device.h"
#include "msda_internal.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void global_void;

constexpr int BM = 128, BN = 128, BK = 32, THREADS = 256;
constexpr int GROUP_BYTES = 1024;               // 4 token rows x 128 columns, one DMA instruction
constexpr int TILE_BYTES = BK / 4 * GROUP_BYTES;  // one operand, one stage: 8 KB
constexpr int STAGE_BYTES = 2 * TILE_BYTES;
constexpr int STAGES = 3;
constexpr int LDS_BYTES = STAGES * STAGE_BYTES;  // 48 KB
constexpr int NUM_XCD = 8;
constexpr int WAVES_PER_SIMD = 3;               // workgroups per CU (4 waves each, one per SIMD)

__device__ __forceinline__ float bf16_to_float(uint32_t bits16) { return __uint_as_float(bits16 << 16); }

// LDS reads are inline asm on purpose: hipcc's waitcnt pass makes every LDS read it can see wait for ALL
// outstanding LDS-DMA (vmcnt(0)), which would serialise the three-stage pipeline; the reads below are
// ordered against the DMAs by the counted vmcnt + barrier in the main loop instead.
template <int OFF>
__device__ __forceinline__ s16x4 lds_tr_read(unsigned addr)
{
#ifndef MSDA_EMU
    s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
#else
    return emu_tr_read(addr, OFF);                     // (host model, tools/emu/)
#endif
}

template <int OFF>
__device__ __forceinline__ u32x4 lds_read_b128(unsigned addr)
{
    u32x4 v;
#ifndef MSDA_EMU
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
#else
    std::memcpy(&v, emu::lds_ptr(addr + OFF), 16);
#endif
    return v;
}

// the 8 transpose-reads of one 16-token k-step: A fragments 0/1 and B fragments 0/1, rows 0-3 | 4-7 of
// the lane's 8 tokens (two consecutive row groups)
struct StepFragments { s16x4 r[8]; };

template <int OFF>
__device__ __forceinline__ void read_step(StepFragments &s, unsigned a0, unsigned a1, unsigned b0, unsigned b1)
{
    s.r[0] = lds_tr_read<OFF>(a0);
    s.r[1] = lds_tr_read<OFF + GROUP_BYTES>(a0);
    s.r[2] = lds_tr_read<OFF>(a1);
    s.r[3] = lds_tr_read<OFF + GROUP_BYTES>(a1);
    s.r[4] = lds_tr_read<OFF + TILE_BYTES>(b0);
    s.r[5] = lds_tr_read<OFF + TILE_BYTES + GROUP_BYTES>(b0);
    s.r[6] = lds_tr_read<OFF + TILE_BYTES>(b1);
    s.r[7] = lds_tr_read<OFF + TILE_BYTES + GROUP_BYTES>(b1);
}

__device__ __forceinline__ void wait_step(StepFragments &s)
{
#ifndef MSDA_EMU
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(s.r[0]), "+v"(s.r[1]), "+v"(s.r[2]), "+v"(s.r[3]), "+v"(s.r[4]), "+v"(s.r[5]), "+v"(s.r[6]),
                   "+v"(s.r[7]));
#else
    (void)s;
#endif
}

__device__ __forceinline__ void mfma_step(const StepFragments &s, f32x16 (&acc)[2][2])
{
    union { s16x4 h[2]; bf16x8 v; } a[2], b[2];
    a[0].h[0] = s.r[0]; a[0].h[1] = s.r[1]; a[1].h[0] = s.r[2]; a[1].h[1] = s.r[3];
    b[0].h[0] = s.r[4]; b[0].h[1] = s.r[5]; b[1].h[0] = s.r[6]; b[1].h[1] = s.r[7];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i].v, b[j].v, acc[i][j], 0, 0, 0);
}

// DIRECT != 0 (only when ONE chunk covers all tokens and there are no tail rows): the tile is final, so it is written
// straight to dW / db -- bf16 (1) or float32 (2) -- and the second pass is not launched at all.  That is the case of the
// 256- and 320-token Linears of the text layers: ~100 of the 230 weight gradients of a train step.
template <int dbg, int DIRECT = 0>
__global__ __launch_bounds__(THREADS, WAVES_PER_SIMD) void wgrad_kernel(
    const uint16_t *__restrict__ dy, const uint16_t *__restrict__ x, int M, int K, int steps_total,
    int steps_per_chunk, int chunks, int bias_parts, float *__restrict__ partial, float *__restrict__ bias_partial)
{
    MSDA_DYNAMIC_LDS_ALIGNED(char, smem, 1024);
    const int tiles_k = K / BN, tiles = (M / BM) * tiles_k;
    const int total = chunks * tiles;
    // physical workgroup b runs on XCD b % 8: give every XCD a contiguous range of logical ids so that
    // the `tiles` workgroups of one chunk share an L2
    const int per_xcd = (total + NUM_XCD - 1) / NUM_XCD;
    const int logical = (blockIdx.x % NUM_XCD) * per_xcd + blockIdx.x / NUM_XCD;
    if (logical >= total) return;
    const int chunk = logical / tiles, tile = logical % tiles;
    const int tm = tile / tiles_k, tk = tile % tiles_k;
    const int s0 = chunk * steps_per_chunk;
    const int nsteps = min(steps_total, s0 + steps_per_chunk) - s0;      // >= 1 by construction of the plan

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // DMA role: wave w fills row groups w and w + 4 of both operand tiles.  Lane l carries the piece that
    // belongs at LDS slot l of the group: row l >> 4, 16-byte piece (l & 15) ^ 4 (l >> 4).
    const int drow = lane >> 4, dpiece = (lane & 15) ^ (4 * (lane >> 4));
    const uint16_t *a_src = dy + ((size_t)s0 * BK + 4 * wave + drow) * M + (size_t)tm * BM + dpiece * 8;
    const uint16_t *b_src = x + ((size_t)s0 * BK + 4 * wave + drow) * K + (size_t)tk * BN + dpiece * 8;
    const size_t a_step = (size_t)BK * M, b_step = (size_t)BK * K;          // elements per 32-token step
    const size_t a_half = (size_t)16 * M, b_half = (size_t)16 * K;          // row group w + 4 is 16 rows on

    // fragment role: lane (p = l & 15, g = l >> 4) reads row r = p >> 2 of row group 2 (l >> 5) (+1), the
    // 8 bytes at columns wave_col + 32 f + 16 (g & 1) + 4 (p & 3): piece index c, stored at slot c ^ 4 r
    const int p = lane & 15, g = lane >> 4, r = p >> 2;
    auto frag_addr = [&](int wave_piece, int f) {
        const int c = wave_piece + 4 * f + 2 * (g & 1) + ((p & 3) >> 1);
        return (lane >> 5) * 2 * GROUP_BYTES + r * 256 + ((c ^ (4 * r)) * 16) + (p & 1) * 8;
    };
    const unsigned lds0 = MSDA_LDS_BYTE_ADDR(smem);
    const unsigned a0 = lds0 + frag_addr(wm * 8, 0), a1 = lds0 + frag_addr(wm * 8, 1);
    const unsigned b0 = lds0 + frag_addr(wn * 8, 0), b1 = lds0 + frag_addr(wn * 8, 1);
    const unsigned bias_addr = lds0 + wave * GROUP_BYTES + lane * 16;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    float bsum[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bsum[j] = 0.f;
    // Bias gradient: the `bias_parts` (1, 2, 4 or 8; 1 when DIRECT) workgroups tk < bias_parts of a (chunk, tm) row share
    // the column sums of the dY tile they all hold: part tk takes row groups [tk, tk + 1) * 8 / bias_parts of every step.
    // (With one workgroup doing all of it -- 2 x b128 + ~40 VALU per thread per step next to 8 MFMAs -- the tk == 0
    //  workgroups ran 1.5x longer than the others and the kernel waited for them: tools/wgrad_big.py, ablation arms.)
    const bool want_bias = (tk < bias_parts) && bias_partial != nullptr;
    const int rg_lo = tk * (8 / bias_parts), rg_hi = rg_lo + 8 / bias_parts;
    const bool bias_h0 = want_bias && wave >= rg_lo && wave < rg_hi;              // wave-uniform
    const bool bias_h1 = want_bias && wave + 4 >= rg_lo && wave + 4 < rg_hi;

    // Always four DMAs per wave per call, so the vmcnt arithmetic below is exact; steps past the end of
    // the chunk re-load its last step into a stage nobody reads again.
    auto issue = [&](int step, int stage) {
        const int st = min(step, nsteps - 1);
        const uint16_t *a = a_src + (size_t)st * a_step;
        const uint16_t *b = b_src + (size_t)st * b_step;
        char *dst = smem + stage * STAGE_BYTES + wave * GROUP_BYTES;
        MSDA_GLOBAL_LOAD_LDS16(a, dst);
        MSDA_GLOBAL_LOAD_LDS16(a + a_half, dst + 4 * GROUP_BYTES);
        MSDA_GLOBAL_LOAD_LDS16(b, dst + TILE_BYTES);
        MSDA_GLOBAL_LOAD_LDS16(b + b_half, dst + TILE_BYTES + 4 * GROUP_BYTES);
    };
    auto compute = [&](auto stage_c) {
        constexpr int SB = decltype(stage_c)::value * STAGE_BYTES;
        StepFragments f0, f1;
        if (dbg & 8) {                                          // DMA only
        } else if (dbg & 2) {
            for (int q = 0; q < 8; ++q) { f0.r[q] = s16x4{1, 2, 3, 4}; f1.r[q] = s16x4{1, 2, 3, 4}; }
            mfma_step(f0, acc);
            mfma_step(f1, acc);
        } else if (dbg & 4) {
            read_step<SB>(f0, a0, a1, b0, b1);
            wait_step(f0);
            read_step<SB + 4 * GROUP_BYTES>(f1, a0, a1, b0, b1);
            wait_step(f1);
            for (int q = 0; q < 8; ++q) acc[0][0][q] += (float)(f0.r[q][0] ^ f1.r[q][1]);
        } else {
        read_step<SB>(f0, a0, a1, b0, b1);
        wait_step(f0);
        read_step<SB + 4 * GROUP_BYTES>(f1, a0, a1, b0, b1);      // in flight behind the first four MFMAs
        mfma_step(f0, acc);
        wait_step(f1);
        mfma_step(f1, acc);
        }
        // thread t re-reads slot t & 63 of row groups (t >> 6) and (t >> 6) + 4 of the dY tile: always
        // the same 8 columns (piece (t & 15) ^ 4 ((t >> 4) & 3)), one row of each group
        auto bias_add = [&](const u32x4 &v) {
            const uint32_t w[4] = {v[0], v[1], v[2], v[3]};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bsum[2 * j] += __uint_as_float(w[j] << 16);
                bsum[2 * j + 1] += __uint_as_float(w[j] & 0xffff0000u);
            }
        };
        if (bias_h0) {
            u32x4 v = lds_read_b128<SB>(bias_addr);
            MSDA_ASM_WAIT_LGKM(v);
            bias_add(v);
        }
        if (bias_h1) {
            u32x4 v = lds_read_b128<SB + 4 * GROUP_BYTES>(bias_addr);
            MSDA_ASM_WAIT_LGKM(v);
            bias_add(v);
        }
    };
    // iteration i: this wave's DMAs of step i have landed once at most the 4 of step i+1 are outstanding;
    // after the barrier everybody's have, and everybody is done reading step i-1's stage, which is where
    // step i+2 goes.
    auto body = [&](int i, auto stage_c) {
        constexpr int stage = decltype(stage_c)::value;
        if (dbg & 1) MSDA_ASM_WAIT_VM(); else
        MSDA_ASM_WAIT_VMCNT(4);
        __builtin_amdgcn_s_barrier();
        MSDA_ASM_FENCE();
        if (!(dbg & 1)) issue(i + 2, (stage + 2) % STAGES);
        compute(stage_c);
    };

    issue(0, 0);
    issue(1, 1);
    for (int i = 0;;) {
        body(i, std::integral_constant<int, 0>{});
        if (++i >= nsteps) break;
        body(i, std::integral_constant<int, 1>{});
        if (++i >= nsteps) break;
        body(i, std::integral_constant<int, 2>{});
        if (++i >= nsteps) break;
    }
    MSDA_ASM_WAIT_VM();                                   // the two surplus DMAs still target our LDS
    __builtin_amdgcn_s_barrier();

    // partial[chunk][m][k]: accumulator register q of lane l is row 8*(q/4) + 4*(l/32) + q%4, column l%32
    auto rne = [](float f) -> uint16_t {
        uint32_t u = __float_as_uint(f);
        if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);     // NaN stays NaN
        return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
    };
    const size_t tile_off = ((size_t)(DIRECT ? 0 : chunk) * M + (size_t)tm * BM + wm * 64) * K + (size_t)tk * BN + wn * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = i * 32 + 8 * (q >> 2) + 4 * (lane >> 5) + (q & 3);
                const size_t e = tile_off + (size_t)row * K + j * 32 + (lane & 31);
                if (DIRECT == 1) reinterpret_cast<uint16_t *>(partial)[e] = rne(acc[i][j][q]);   // `partial` IS dW here
                else partial[e] = acc[i][j][q];
            }

    if (want_bias) {                                           // (all waves are past their last LDS read)
        float *red = reinterpret_cast<float *>(smem);          // [16 threads per column piece][128 columns]
        const int piece = (tid & 15) ^ (4 * ((tid >> 4) & 3));
#pragma unroll
        for (int j = 0; j < 8; ++j) red[(tid >> 4) * BM + piece * 8 + j] = bsum[j];
        __syncthreads();
        if (tid < BM) {
            float s = 0.f;
#pragma unroll
            for (int rg = 0; rg < 16; ++rg) s += red[rg * BM + tid];
            const size_t e = (size_t)(DIRECT ? 0 : chunk * bias_parts + tk) * M + tm * BM + tid;
            if (DIRECT == 1) reinterpret_cast<uint16_t *>(bias_partial)[e] = rne(s);              // `bias_partial` IS db
            else bias_partial[e] = s;
        }
    }
}

// out[e] = sum_c partial[c][e] + the contribution of the tail rows [tail0, T) the main kernel left out.
// A workgroup owns 256/CG consecutive float4s and splits the chunks (and the tail rows) over CG thread groups:
// with many chunks (CG = 16) every thread has only chunks/16 independent loads in flight and the grid is n/64
// workgroups (a thread-per-element loop over 128 chunks was latency-bound at ~30 us); with a handful of chunks
// (small token counts) CG = 4 or 1 keeps all 256 threads busy instead of 15 idle groups out of 16.
// BIAS: n = M and the tail term is dy[t][e..e+3]; otherwise e = m*K + k and it is dy[t][m] * x[t][k..k+3].
template <bool F32, bool BIAS, int CG>
__device__ __forceinline__ void reduce_body(const float *__restrict__ partial, int chunks, size_t n, int block,
                                            const uint16_t *__restrict__ dy, const uint16_t *__restrict__ x, int tail0,
                                            int T, int M, int K, void *__restrict__ out, float4 *red)
{
    constexpr int Q = 256 / CG;
    const int q = threadIdx.x % Q, cg = threadIdx.x / Q;
    const size_t e = ((size_t)block * Q + q) * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e < n) {
#pragma unroll 4
        for (int c = cg; c < chunks; c += CG) {
            const float4 v = *reinterpret_cast<const float4 *>(partial + (size_t)c * n + e);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        const int m = BIAS ? (int)e : (int)(e / K), k = BIAS ? 0 : (int)(e % K);
        for (int t = tail0 + cg; t < T; t += CG) {
            if (BIAS) {
                const uint2 d = *reinterpret_cast<const uint2 *>(dy + (size_t)t * M + m);
                s.x += bf16_to_float(d.x & 0xffffu); s.y += bf16_to_float(d.x >> 16);
                s.z += bf16_to_float(d.y & 0xffffu); s.w += bf16_to_float(d.y >> 16);
            } else {
                const float d = bf16_to_float(dy[(size_t)t * M + m]);
                const uint2 xv = *reinterpret_cast<const uint2 *>(x + (size_t)t * K + k);
                s.x += d * bf16_to_float(xv.x & 0xffffu); s.y += d * bf16_to_float(xv.x >> 16);
                s.z += d * bf16_to_float(xv.y & 0xffffu); s.w += d * bf16_to_float(xv.y >> 16);
            }
        }
    }
    if (CG > 1) {
        red[cg * Q + q] = s;
        __syncthreads();
        if (cg != 0) return;
#pragma unroll
        for (int g2 = 1; g2 < CG; ++g2) {
            const float4 v = red[g2 * Q + q];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    if (e >= n) return;
    if (F32) {
        *reinterpret_cast<float4 *>(static_cast<float *>(out) + e) = s;
    } else {
        auto rne = [](float f) -> uint32_t {
            uint32_t u = __float_as_uint(f);
            if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;     // NaN stays NaN
            return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
        };
        uint2 o;
        o.x = rne(s.x) | (rne(s.y) << 16);
        o.y = rne(s.z) | (rne(s.w) << 16);
        *reinterpret_cast<uint2 *>(static_cast<uint16_t *>(out) + e) = o;
    }
}

// one launch for both results: workgroups [0, dw_blocks) reduce the weight-gradient partials, the rest the
// bias-gradient partials (db may be absent: then the grid is dw_blocks)
template <bool F32, int CG>
__global__ __launch_bounds__(256) void reduce_partials(const float *__restrict__ partial,
                                                       const float *__restrict__ bias_partial, int chunks, int bias_slots,
                                                       int dw_blocks,
                                                       const uint16_t *__restrict__ dy, const uint16_t *__restrict__ x,
                                                       int tail0, int T, int M, int K, void *__restrict__ dw,
                                                       void *__restrict__ db)
{
    __shared__ float4 red[256];
    if ((int)blockIdx.x < dw_blocks)
        reduce_body<F32, false, CG>(partial, chunks, (size_t)M * K, blockIdx.x, dy, x, tail0, T, M, K, dw, red);
    else
        reduce_body<F32, true, CG>(bias_partial, bias_slots, (size_t)M, blockIdx.x - dw_blocks, dy, x, tail0, T, M, K, db, red);
}

struct Plan { int chunks, steps_per_chunk, steps_total, tiles, bias_parts; size_t partial_floats, bias_floats; };

bool make_plan(int T, int M, int K, Plan &pl)
{
    if (T < 1 || M < BM || K < BN || M % BM || K % BN) return false;
    pl.steps_total = T / BK;                                   // whole 32-token steps; the rest is the tail
    pl.tiles = (M / BM) * (K / BN);
    // two workgroups per CU in flight (a third fits, but its extra partial sums cost more than it hides)
    static int target = msda::ablation_env("RLIPV2_WGRAD_BLOCKS", 512);   // measured: 512 beats 768 (fewer partials) and 256
    int chunks = (target + pl.tiles - 1) / pl.tiles;
    // a workgroup that runs only a step or two pays the pipeline prologue and a partial-sum slot for nothing
    static int min_steps = msda::ablation_env("RLIPV2_WGRAD_MINSTEPS", 8);   // measured (tools/wgrad_small.py): 8 beats 1-4 and 16 on the 256-1200-token shapes
    if (chunks > pl.steps_total / min_steps) chunks = pl.steps_total / min_steps;
    if (chunks < 1) chunks = 1;
    pl.steps_per_chunk = pl.steps_total ? (pl.steps_total + chunks - 1) / chunks : 0;
    pl.chunks = pl.steps_total ? (pl.steps_total + pl.steps_per_chunk - 1) / pl.steps_per_chunk : 0;
    // at least one slot so that the workspace is never empty
    pl.partial_floats = (size_t)(pl.chunks ? pl.chunks : 1) * M * K;
    // workgroups of one (chunk, tm) row that share the bias gradient's column sums (power of two, <= 8 row groups)
    const int tiles_k = K / BN;
    pl.bias_parts = tiles_k >= 8 ? 8 : tiles_k >= 4 ? 4 : tiles_k >= 2 ? 2 : 1;
    if (pl.chunks <= 1) pl.bias_parts = 1;                     // (the one-chunk kernel writes db itself)
    pl.bias_floats = (size_t)(pl.chunks ? pl.chunks : 1) * pl.bias_parts * M;
    return true;
}

}  // namespace

extern "C" size_t linear_wgrad_workspace_bytes(int T, int M, int K)
{
    Plan pl;
    if (!make_plan(T, M, K, pl)) return 0;
    return (pl.partial_floats + pl.bias_floats) * sizeof(float);
}

extern "C" int linear_wgrad_supported(int T, int M, int K)
{
    Plan pl;
    return make_plan(T, M, K, pl) ? 1 : 0;
}

extern "C" int linear_wgrad_bf16(const void *dy, const void *x, int T, int M, int K, void *dw, void *db, int out_f32,
                                 void *workspace, size_t workspace_bytes, void *stream_)
{
    Plan pl;
    if (!make_plan(T, M, K, pl)) return MSDA_ERR_BAD_SHAPE;
    if (!dy || !x || !dw || !workspace) return MSDA_ERR_NULL_POINTER;
    if (workspace_bytes < (pl.partial_floats + pl.bias_floats) * sizeof(float)) return MSDA_ERR_BAD_SHAPE;
    if ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dw) |
         reinterpret_cast<uintptr_t>(workspace)) & 15u)
        return MSDA_ERR_ALIGNMENT;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const uint16_t *dy16 = static_cast<const uint16_t *>(dy), *x16 = static_cast<const uint16_t *>(x);
    float *partial = static_cast<float *>(workspace);
    float *bias_partial = partial + pl.partial_floats;
    if (pl.chunks == 1 && T % BK == 0) {        // one chunk, no tail rows: final results straight from the main kernel
        const int grid = ((pl.tiles + NUM_XCD - 1) / NUM_XCD) * NUM_XCD;
        if (out_f32)
            hipLaunchKernelGGL((wgrad_kernel<0, 2>), dim3(grid), dim3(THREADS), LDS_BYTES, stream, dy16, x16, M, K,
                               pl.steps_total, pl.steps_per_chunk, pl.chunks, 1, static_cast<float *>(dw),
                               static_cast<float *>(db));
        else
            hipLaunchKernelGGL((wgrad_kernel<0, 1>), dim3(grid), dim3(THREADS), LDS_BYTES, stream, dy16, x16, M, K,
                               pl.steps_total, pl.steps_per_chunk, pl.chunks, 1, static_cast<float *>(dw),
                               static_cast<float *>(db));
        return hipGetLastError() == hipSuccess ? MSDA_OK : MSDA_ERR_LAUNCH;
    }
    if (pl.chunks > 0) {
        const int total = pl.chunks * pl.tiles;
        const int grid = ((total + NUM_XCD - 1) / NUM_XCD) * NUM_XCD;
        static int dbg = msda::ablation_env("RLIPV2_WGRAD_DBG", 0);      // ablation builds only (skips work)
#define LAUNCH(D) hipLaunchKernelGGL(wgrad_kernel<D>, dim3(grid), dim3(THREADS), LDS_BYTES, stream, dy16, x16, M, K, \
                                     pl.steps_total, pl.steps_per_chunk, pl.chunks, pl.bias_parts, partial, db ? bias_partial : nullptr)
        switch (dbg) {
        case 1: LAUNCH(1); break;
        case 2: LAUNCH(2); break;
        case 3: LAUNCH(3); break;
        case 4: LAUNCH(4); break;
        case 5: LAUNCH(5); break;
        case 7: LAUNCH(7); break;
        case 8: LAUNCH(8); break;
        default: LAUNCH(0); break;
        }
#undef LAUNCH
    }
    const size_t n = (size_t)M * K;
    const int tail0 = pl.steps_total * BK;
    // chunk groups per workgroup of the second pass (tail rows count like chunks: they are split the same way)
    const int terms = pl.chunks + (T - tail0 + 3) / 4;
    const int cg = terms >= 16 ? 16 : terms >= 4 ? 4 : 1, q = 256 / cg;
    const int gw = (int)((n / 4 + q - 1) / q), gb = db ? (int)(((size_t)M / 4 + q - 1) / q) : 0;
#define REDUCE(F, C) hipLaunchKernelGGL((reduce_partials<F, C>), dim3(gw + gb), dim3(256), 0, stream, partial, bias_partial, \
                                        pl.chunks, pl.chunks * pl.bias_parts, gw, dy16, x16, tail0, T, M, K, dw, db)
    if (out_f32) { if (cg == 16) REDUCE(true, 16); else if (cg == 4) REDUCE(true, 4); else REDUCE(true, 1); }
    else         { if (cg == 16) REDUCE(false, 16); else if (cg == 4) REDUCE(false, 4); else REDUCE(false, 1); }
#undef REDUCE
    return hipGetLastError() == hipSuccess ? MSDA_OK : MSDA_ERR_LAUNCH;
}
