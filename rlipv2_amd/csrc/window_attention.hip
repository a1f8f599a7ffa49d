// window_attention.hip -- the window attention of the Swin backbones (BASELINE configs 4-5; reference
// models/swin/swin_transformer.py:221-301, WindowAttention.forward) as ONE kernel per direction for gfx950:
//
//   S = scale Q K^T + relative-position bias (+ the shift mask of the window)      [N, N], N = window_size^2 <= 64
//   P = softmax_j(S),   O = P V                                                     head_dim 32
//
// As PyTorch ops that is matmul + add + float32 softmax + cast + matmul (and their backward), each a pass over the
// [B, windows, heads, N, N] attention tensor -- 40 M elements per block at 800 x 1333 in stage 0.  Here a WAVE owns one
// (window, head) at a time and nothing N x N ever leaves the CU:
//   * S^T = K Q^T on v_mfma_f32_32x32x16_bf16 with the operand fragments loaded straight from the packed qkv tensor (16
//     bytes per lane, k-contiguous as the projection leaves them): in the C layout of that instruction a LANE holds one
//     QUERY (column) and its registers run over the KEYS (rows), so the softmax is 2 x 32 in-lane operations and one
//     exchange with the other half-wave -- no cross-lane reduction trees;
//   * the relative-position bias is handed over padded to [64 queries][64 keys] with -30000 in the padded KEY columns (that
//     is also the padding mask): a lane reads its query's row, 4 consecutive keys per 16-byte load, from a 16 KB table that
//     stays in the cache; the shift mask of the few windows that have one comes from a compact table of distinct masks;
//   * P goes through the wave's own LDS tile as bfloat16 (a lane writes the row of its query, 8 bytes at a time; the A
//     fragments of P V are 16-byte reads of that row), V through a transposed LDS tile; O = P V on the same instruction.
// The backward recomputes S and P from Q and K (8 MFMAs) instead of saving them:
//   dV = P^T dO,  dP^T = V dO^T,  dS = P o (dP - rowsum(P o dP)),  dQ = scale dS K,  dK = scale dS^T Q
// with dS leaving through LDS in both orientations.  Written in round 5 without a GPU: checked on the lane-level model
// of tools/emu/ against float32 PyTorch (tests/test_dense_emulated.py), NOT yet run on hardware; the host side keeps it
// behind the self-checked route `fused_window_attention` (rlipv2_amd/routes.py).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rlipv2_msda.h"
#include "../../include/rlipv2_swin.h"

#ifndef MSDA_DYNAMIC_LDS
#define MSDA_DYNAMIC_LDS(type, name) extern __shared__ __attribute__((aligned(16))) type name[]
#endif
#ifndef MSDA_ASM_FENCE
#define MSDA_ASM_FENCE() asm volatile("" ::: "memory")
#endif
// the lanes of ONE wave hand data to each other through LDS: program order is enough on the hardware (a wave's LDS
// operations complete in order), the compiler must not move memory operations across; a wave barrier in the host model
#define WATT_WAVE_SYNC() do { MSDA_ASM_FENCE(); __builtin_amdgcn_wave_barrier(); MSDA_ASM_FENCE(); } while (0)

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int HD = 32;                  // head dimension (every Swin preset: C / heads = 32)
constexpr int NP = 64;                  // tokens per window, padded (window_size <= 8)
constexpr int WAVES = 2, THREADS = WAVES * 64;    // (LDS per wave decides the waves per CU: small workgroups pack best)
constexpr int P_STRIDE = NP + 8;        // bf16 per row of an [NP][NP] LDS tile (+ 16 bytes: rows on different banks)
constexpr int T_STRIDE = NP + 8;        // bf16 per row of a transposed operand tile [HD][NP]
constexpr int CODE_BYTES = NP * 4;      // per wave: the row code of each of the window's tokens
constexpr int FWD_WAVE_LDS = (NP * P_STRIDE + HD * T_STRIDE) * 2 + CODE_BYTES;      // P | V^T | codes
constexpr int BWD_WAVE_LDS = (2 * NP * P_STRIDE + HD * T_STRIDE) * 2 + CODE_BYTES; // X^T (P^T, then dS^T) | dS | dO^T, then K^T, then Q^T | codes

// two floats -> packed bfloat16 pair, round to nearest even: ONE v_cvt_pk_bf16_f32 on gfx950 (the bit-twiddling form is 8
// VALU instructions per value, and the probabilities alone are 64 values per lane and window)
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__device__ __forceinline__ uint32_t pack2(float a, float b)
{
    const f32x2_t f = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2_t));
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
__device__ __forceinline__ Frag lds_frag(const uint16_t *p)
{
    Frag f;
    f.u = *reinterpret_cast<const uint4 *>(p);
    return f;
}

// C / D layout of v_mfma_f32_32x32x16: lane -> column (lane & 31), register r -> row (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
// (alif_attention.hip runs the same instruction with the same layouts on the hardware)
__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// the task of a wave: head `head`, windows first, first + step, ... < windows
struct Task { int head, first, step; bool any; };

__device__ __forceinline__ Task task_of(int heads, int windows)
{
    const int g = blockIdx.x * WAVES + (threadIdx.x >> 6);          // global wave index: head fastest
    const int chunks = (gridDim.x * WAVES) / heads;                  // whole groups of `heads` waves
    Task t;
    t.head = g % heads;
    t.first = g / heads;
    t.step = chunks;
    t.any = t.first < chunks && t.first < windows;
    return t;
}

// Where the rows of a window's tokens live.  map == nullptr: the tensors are window-major, token t of window w is row w N + t
// (the layout the reference's window_partition produces).  map != nullptr: the tensors stay in IMAGE order [B, H W, .] and
// map[(w mod windows_per_image) N + t] is the token's row inside its image, or -(slot + 1) for a token of the zero padding
// (window_partition after F.pad, models/swin/swin_transformer.py:362-379): its q / k / v are the projection's bias (`pad_row`),
// its output is dropped and its gradient goes to row `slot` of a small side buffer (it belongs to the bias) -- pad, cyclic
// shift and window partition / reverse become addressing, no copies.
struct RowMap { const int *map; int rows_per_image, pads_per_image, windows_per_image; };

// code of token `tok` of window w: >= 0 a row, -1 no such token (tok >= N), <= -2 the pad slot -(code + 2)
__device__ __forceinline__ int token_code(const RowMap &rm, int w, int N, int tok)
{
    if (tok >= N) return -1;
    if (!rm.map) return w * N + tok;
    const int b = w / rm.windows_per_image, wl = w - b * rm.windows_per_image;
    const int r = rm.map[wl * N + tok];
    return r >= 0 ? b * rm.rows_per_image + r : -2 - (b * rm.pads_per_image + (-r - 1));
}

// row of a token in an input matrix: `mat` for real rows, the single row `pad` for padding tokens (nullptr: zeros), nullptr otherwise
__device__ __forceinline__ const uint16_t *in_row(const uint16_t *mat, const uint16_t *pad, int row_stride, int code)
{
    return code >= 0 ? mat + (size_t)code * row_stride : (code <= -2 ? pad : nullptr);
}

// operand fragments of one 32-token tile: [k-step] (16 channels each), zeros for tokens without a row
__device__ __forceinline__ void load_tile_frags(const uint16_t *mat, const uint16_t *pad, int row_stride, const int *codes, int tile,
                                                int lane, Frag (&f)[2])
{
    const int kg = (lane >> 5) * 8;
    const uint16_t *row = in_row(mat, pad, row_stride, codes[tile * 32 + (lane & 31)]);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) f[ks] = load_frag(row ? row + ks * 16 + kg : nullptr, row != nullptr);
}

// x^T[jt][r] = sum over the 32 channels of  rows[key jt * 32 + acc_row(r)] . cols[this tile's query lane & 31]
__device__ __forceinline__ void product_keys_by_queries(const Frag (&rows)[2][2], const Frag (&cols)[2], f32x16 (&x)[2])
{
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) x[jt][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) x[jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rows[jt][ks].v, cols[ks].v, x[jt], 0, 0, 0);
    }
}

// logits of one query tile (s[jt][r]: key jt * 32 + acc_row(r), query it * 32 + (lane & 31)), in place:
// s <- softmax over the keys of  scale s + bias (+ mask).  The bias of the wave's head is read from the (cache-resident)
// table for every window: holding it in 64 registers halved the waves a SIMD can keep.
__device__ __forceinline__ void softmax_keys(f32x16 (&s)[2], float scale, const float *bias_h, const float *mask_w, int it, int lane)
{
    // the lane's query row of the padded [64 queries][64 keys] table: registers 4 g .. 4 g + 3 of key tile jt are the 4
    // consecutive keys jt * 32 + 8 g + 4 (lane >> 5) + 0..3 -- one 16-byte load each
    const int q = it * 32 + (lane & 31), half = lane >> 5;
    float add[2][16];
    const float4 *brow = reinterpret_cast<const float4 *>(bias_h + q * NP);
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 b = brow[jt * 8 + 2 * g + half];
            add[jt][4 * g] = b.x; add[jt][4 * g + 1] = b.y; add[jt][4 * g + 2] = b.z; add[jt][4 * g + 3] = b.w;
        }
    if (mask_w) {                                                    // (wave-uniform)
        const float4 *mrow = reinterpret_cast<const float4 *>(mask_w + q * NP);
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 b = mrow[jt * 8 + 2 * g + half];
                add[jt][4 * g] += b.x; add[jt][4 * g + 1] += b.y; add[jt][4 * g + 2] += b.z; add[jt][4 * g + 3] += b.w;
            }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[jt][r] = s[jt][r] * scale + add[jt][r];
            mx = fmaxf(mx, s[jt][r]);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));                          // the other half-wave holds the other 32 keys
    float sum = 0.f;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[jt][r] = __expf(s[jt][r] - mx);
            sum += s[jt][r];
        }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.f / sum;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[jt][r] *= inv;
}

// x^T registers of query tile `it` (lane = query, registers = keys) -> LDS tile rows[query][key] as bfloat16: the lane's 4
// consecutive keys of a register group leave as one 8-byte store
__device__ __forceinline__ void store_rows(uint16_t *tile, const f32x16 (&x)[2], int it, int lane)
{
    const int li = lane & 31, half = lane >> 5;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int j0 = jt * 32 + 8 * g + 4 * half;
            *reinterpret_cast<uint2 *>(tile + (it * 32 + li) * P_STRIDE + j0) =
                make_uint2(pack2(x[jt][4 * g], x[jt][4 * g + 1]), pack2(x[jt][4 * g + 2], x[jt][4 * g + 3]));
        }
}

// x^T registers of query tile `it` -> LDS tile cols[key][query] as bfloat16 (for a fixed register the 32 lanes of a
// half-wave write 32 consecutive queries)
__device__ __forceinline__ void store_cols(uint16_t *tile, const f32x16 (&x)[2], int it, int lane)
{
    const int li = lane & 31;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) tile[(jt * 32 + acc_row(r, lane)) * P_STRIDE + it * 32 + li] = (uint16_t)(pack2(x[jt][r], 0.f) & 0xffffu);
}

// LDS tile rows[a][b] -> LDS tile cols[b][a]; lane = row a
__device__ __forceinline__ void transpose_tile(uint16_t *dst, const uint16_t *src, int lane)
{
#pragma unroll
    for (int c = 0; c < NP / 8; ++c) {
        const uint4 v = *reinterpret_cast<const uint4 *>(src + lane * P_STRIDE + c * 8);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            dst[(c * 8 + 2 * e) * P_STRIDE + lane] = (uint16_t)(w[e] & 0xffffu);
            dst[(c * 8 + 2 * e + 1) * P_STRIDE + lane] = (uint16_t)(w[e] >> 16);
        }
    }
}

// the window's rows [token][32 channels] in global memory -> transposed LDS tile [channel][token]; lane = token
__device__ __forceinline__ void stage_transposed(uint16_t *tile, const uint16_t *mat, const uint16_t *pad, int row_stride,
                                                 const int *codes, int lane)
{
    const uint16_t *row = in_row(mat, pad, row_stride, codes[lane]);
    uint4 v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = row ? *reinterpret_cast<const uint4 *>(row + c * 8) : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const uint32_t w[4] = {v[c].x, v[c].y, v[c].z, v[c].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            tile[(c * 8 + 2 * e) * T_STRIDE + lane] = (uint16_t)(w[e] & 0xffffu);
            tile[(c * 8 + 2 * e + 1) * T_STRIDE + lane] = (uint16_t)(w[e] >> 16);
        }
    }
}

// row tile `rt` of  A [NP x NP, LDS rows] x B^T [32 x NP, transposed LDS tile]  -> the tokens' rows in `dst` (padding tokens: row
// `slot` of `dst_pad`, or nowhere), scaled: o[r] = token rt * 32 + acc_row(r), channel lane & 31
__device__ __forceinline__ void product_tile_store(const uint16_t *a_tile, const uint16_t *bt_tile, int rt, uint16_t *dst,
                                                   uint16_t *dst_pad, int row_stride, const int *codes, float scale, int lane)
{
    const int li = lane & 31, kg = (lane >> 5) * 8;
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < NP / 16; ++ks) {
        const Frag a = lds_frag(a_tile + (rt * 32 + li) * P_STRIDE + ks * 16 + kg);
        const Frag b = lds_frag(bt_tile + li * T_STRIDE + ks * 16 + kg);
        o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, o, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int code = codes[rt * 32 + acc_row(r, lane)];
        uint16_t *row = code >= 0 ? dst + (size_t)code * row_stride : (code <= -2 && dst_pad ? dst_pad + (size_t)(-2 - code) * row_stride : nullptr);
        if (row) row[li] = (uint16_t)(pack2(o[r] * scale, 0.f) & 0xffffu);
    }
}

__global__ __launch_bounds__(THREADS) void window_attention_forward_kernel(
    const uint16_t *__restrict__ qkv, const uint16_t *__restrict__ pad_row, RowMap rm, const float *__restrict__ bias_t,
    const float *__restrict__ mask_t, const int *__restrict__ mask_id, int windows, int heads, int N, float scale,
    uint16_t *__restrict__ out)
{
    MSDA_DYNAMIC_LDS(unsigned char, lds);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint16_t *Pl = reinterpret_cast<uint16_t *>(lds + wave * FWD_WAVE_LDS);
    uint16_t *Vt = Pl + NP * P_STRIDE;
    int *codes = reinterpret_cast<int *>(Vt + HD * T_STRIDE);
    const Task t = task_of(heads, windows);
    if (!t.any) return;
    const int C = heads * HD, row_stride = 3 * C;
    const float *bias_h = bias_t + (size_t)t.head * NP * NP;
    const uint16_t *qh = qkv + t.head * HD, *ph = pad_row ? pad_row + t.head * HD : nullptr;
    for (int w = t.first; w < windows; w += t.step) {
        const int mid = mask_id ? mask_id[w % rm.windows_per_image] : -1;
        const float *mask_w = mid >= 0 ? mask_t + (size_t)mid * NP * NP : nullptr;
        WATT_WAVE_SYNC();                                            // the previous window's tiles and codes are no longer read
        codes[lane] = token_code(rm, w, N, lane);
        WATT_WAVE_SYNC();
        Frag kf[2][2];
        load_tile_frags(qh + C, ph ? ph + C : nullptr, row_stride, codes, 0, lane, kf[0]);
        load_tile_frags(qh + C, ph ? ph + C : nullptr, row_stride, codes, 1, lane, kf[1]);
        stage_transposed(Vt, qh + 2 * C, ph ? ph + 2 * C : nullptr, row_stride, codes, lane);
#pragma unroll 1
        for (int it = 0; it < 2; ++it) {                             // one tile of 32 queries at a time (register budget)
            Frag qf[2];
            load_tile_frags(qh, ph, row_stride, codes, it, lane, qf);
            f32x16 s[2];
            product_keys_by_queries(kf, qf, s);
            softmax_keys(s, scale, bias_h, mask_w, it, lane);
            store_rows(Pl, s, it, lane);
            WATT_WAVE_SYNC();
            product_tile_store(Pl, Vt, it, out + t.head * HD, nullptr, C, codes, 1.f, lane);
        }
    }
}

// d_qkv of one (window, head) from qkv and d_out; see the file header for the formulas
__global__ __launch_bounds__(THREADS) void window_attention_backward_kernel(
    const uint16_t *__restrict__ qkv, const uint16_t *__restrict__ pad_row, RowMap rm, const uint16_t *__restrict__ d_out,
    const float *__restrict__ bias_t, const float *__restrict__ mask_t, const int *__restrict__ mask_id, int windows, int heads,
    int N, float scale, uint16_t *__restrict__ d_qkv, uint16_t *__restrict__ d_pad)
{
    MSDA_DYNAMIC_LDS(unsigned char, lds);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint16_t *Xt = reinterpret_cast<uint16_t *>(lds + wave * BWD_WAVE_LDS);     // [key][query]: P^T, later dS^T
    uint16_t *dSl = Xt + NP * P_STRIDE;                                         // [query][key]
    uint16_t *Tt = dSl + NP * P_STRIDE;                                         // [channel][token]: dO^T, then K^T, then Q^T
    int *codes = reinterpret_cast<int *>(Tt + HD * T_STRIDE);
    const Task t = task_of(heads, windows);
    if (!t.any) return;
    const int C = heads * HD, row_stride = 3 * C;
    const float *bias_h = bias_t + (size_t)t.head * NP * NP;
    const uint16_t *qh = qkv + t.head * HD, *ph = pad_row ? pad_row + t.head * HD : nullptr;
    const uint16_t *doh = d_out + t.head * HD;
    uint16_t *gh = d_qkv + t.head * HD, *gp = d_pad ? d_pad + t.head * HD : nullptr;
    for (int w = t.first; w < windows; w += t.step) {
        const int mid = mask_id ? mask_id[w % rm.windows_per_image] : -1;
        const float *mask_w = mid >= 0 ? mask_t + (size_t)mid * NP * NP : nullptr;
        WATT_WAVE_SYNC();                                            // the previous window's tiles and codes are no longer read
        codes[lane] = token_code(rm, w, N, lane);
        WATT_WAVE_SYNC();
        Frag kf[2][2], vf[2][2];
        load_tile_frags(qh + C, ph ? ph + C : nullptr, row_stride, codes, 0, lane, kf[0]);
        load_tile_frags(qh + C, ph ? ph + C : nullptr, row_stride, codes, 1, lane, kf[1]);
        load_tile_frags(qh + 2 * C, ph ? ph + 2 * C : nullptr, row_stride, codes, 0, lane, vf[0]);
        load_tile_frags(qh + 2 * C, ph ? ph + 2 * C : nullptr, row_stride, codes, 1, lane, vf[1]);
        stage_transposed(Tt, doh, nullptr, C, codes, lane);          // (the output of a padding token is dropped: zero gradient)
#pragma unroll 1
        for (int it = 0; it < 2; ++it) {                             // one tile of 32 queries at a time (register budget)
            Frag qf[2], gf[2];
            load_tile_frags(qh, ph, row_stride, codes, it, lane, qf);
            load_tile_frags(doh, nullptr, C, codes, it, lane, gf);
            f32x16 p[2], dp[2];
            product_keys_by_queries(kf, qf, p);
            softmax_keys(p, scale, bias_h, mask_w, it, lane);
            product_keys_by_queries(vf, gf, dp);                     // dP^T = V dO^T: same layout as P^T
            // dS = P o (dP - sum over the keys of P dP), per query (lane), in place in dp
            float dot = 0.f;
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int r = 0; r < 16; ++r) dot += p[jt][r] * dp[jt][r];
            dot += __shfl_xor(dot, 32, 64);
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int r = 0; r < 16; ++r) dp[jt][r] = p[jt][r] * (dp[jt][r] - dot);
            store_cols(Xt, p, it, lane);                             // P^T for dV = P^T dO
            store_rows(dSl, dp, it, lane);
        }
        WATT_WAVE_SYNC();
#pragma unroll 1
        for (int rt = 0; rt < 2; ++rt)                               // dV[key][ch] = sum_query P^T[key][query] dO[query][ch]
            product_tile_store(Xt, Tt, rt, gh + 2 * C, gp ? gp + 2 * C : nullptr, row_stride, codes, 1.f, lane);
        WATT_WAVE_SYNC();
        stage_transposed(Tt, qh + C, ph ? ph + C : nullptr, row_stride, codes, lane);
        transpose_tile(Xt, dSl, lane);                               // P^T has been read: its tile takes dS^T
        WATT_WAVE_SYNC();
#pragma unroll 1
        for (int rt = 0; rt < 2; ++rt)                               // dQ[query][ch] = scale sum_key dS[query][key] K[key][ch]
            product_tile_store(dSl, Tt, rt, gh, gp, row_stride, codes, scale, lane);
        WATT_WAVE_SYNC();
        stage_transposed(Tt, qh, ph, row_stride, codes, lane);
        WATT_WAVE_SYNC();
#pragma unroll 1
        for (int rt = 0; rt < 2; ++rt)                               // dK[key][ch] = scale sum_query dS^T[key][query] Q[query][ch]
            product_tile_store(Xt, Tt, rt, gh + C, gp ? gp + C : nullptr, row_stride, codes, scale, lane);
    }
}

bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

int grid_for(int windows, int heads)
{
    // enough waves for every CU (256 CUs x 8) in whole groups of `heads` waves; a wave keeps its head's bias in registers
    int chunks = 4096 / heads;
    if (chunks < 1) chunks = 1;
    if (chunks > windows) chunks = windows;
    return (chunks * heads + WAVES - 1) / WAVES;
}

}  // namespace

extern "C" int window_attention_supported(int windows, int heads, int tokens, int head_dim)
{
    return windows >= 1 && heads >= 1 && heads <= 1024 && tokens >= 1 && tokens <= NP && head_dim == HD &&
           (long)windows * tokens * heads * HD * 3 < (1L << 31);
}

namespace {
int launch_forward(const void *qkv, const void *pad_row, const RowMap &rm, const float *bias_t, const float *mask_t, const int *mask_id,
                   int windows, int heads, int tokens, float scale, void *out, void *stream)
{
    static_assert(WAVES * BWD_WAVE_LDS <= 64 * 1024 && WAVES * FWD_WAVE_LDS <= 64 * 1024, "dynamic LDS within the default limit");
    (void)hipGetLastError();
    hipLaunchKernelGGL(window_attention_forward_kernel, dim3(grid_for(windows, heads)), dim3(THREADS), WAVES * FWD_WAVE_LDS,
                       (hipStream_t)stream, (const uint16_t *)qkv, (const uint16_t *)pad_row, rm, bias_t, mask_t, mask_id, windows, heads,
                       tokens, scale, (uint16_t *)out);
    return hipGetLastError() == hipSuccess ? MSDA_OK : MSDA_ERR_LAUNCH;
}

int launch_backward(const void *qkv, const void *pad_row, const RowMap &rm, const void *d_out, const float *bias_t, const float *mask_t,
                    const int *mask_id, int windows, int heads, int tokens, float scale, void *d_qkv, void *d_pad, void *stream)
{
    (void)hipGetLastError();
    hipLaunchKernelGGL(window_attention_backward_kernel, dim3(grid_for(windows, heads)), dim3(THREADS), WAVES * BWD_WAVE_LDS,
                       (hipStream_t)stream, (const uint16_t *)qkv, (const uint16_t *)pad_row, rm, (const uint16_t *)d_out, bias_t, mask_t,
                       mask_id, windows, heads, tokens, scale, (uint16_t *)d_qkv, (uint16_t *)d_pad);
    return hipGetLastError() == hipSuccess ? MSDA_OK : MSDA_ERR_LAUNCH;
}

bool rows_ok(const int *rowmap, int windows, int windows_per_image, int rows_per_image, int pads_per_image, int tokens)
{
    return rowmap && windows_per_image >= 1 && windows % windows_per_image == 0 && rows_per_image >= 1 && pads_per_image >= 0 &&
           (long)(windows / windows_per_image) * rows_per_image < (1L << 30) &&
           (long)rows_per_image + pads_per_image == (long)windows_per_image * tokens;
}
}  // namespace

extern "C" int window_attention_forward_bf16(const void *qkv, const float *bias_t, const float *mask_t, const int *mask_id,
                                             int windows, int windows_per_image, int heads, int tokens, float scale, void *out,
                                             void *stream)
{
    if (!window_attention_supported(windows, heads, tokens, HD) || windows_per_image < 1) return MSDA_ERR_BAD_SHAPE;
    if (!qkv || !bias_t || !out || ((mask_t == nullptr) != (mask_id == nullptr))) return MSDA_ERR_NULL_POINTER;
    if (!aligned16(qkv) || !aligned16(out)) return MSDA_ERR_ALIGNMENT;
    const RowMap rm = {nullptr, 0, 0, windows_per_image};
    return launch_forward(qkv, nullptr, rm, bias_t, mask_t, mask_id, windows, heads, tokens, scale, out, stream);
}

extern "C" int window_attention_backward_bf16(const void *qkv, const void *d_out, const float *bias_t, const float *mask_t,
                                              const int *mask_id, int windows, int windows_per_image, int heads, int tokens,
                                              float scale, void *d_qkv, void *stream)
{
    if (!window_attention_supported(windows, heads, tokens, HD) || windows_per_image < 1) return MSDA_ERR_BAD_SHAPE;
    if (!qkv || !d_out || !bias_t || !d_qkv || ((mask_t == nullptr) != (mask_id == nullptr))) return MSDA_ERR_NULL_POINTER;
    if (!aligned16(qkv) || !aligned16(d_out) || !aligned16(d_qkv)) return MSDA_ERR_ALIGNMENT;
    const RowMap rm = {nullptr, 0, 0, windows_per_image};
    return launch_backward(qkv, nullptr, rm, d_out, bias_t, mask_t, mask_id, windows, heads, tokens, scale, d_qkv, nullptr, stream);
}

extern "C" int window_attention_rows_forward_bf16(const void *qkv, const void *pad_row, const int *rowmap, int rows_per_image,
                                                  int pads_per_image, const float *bias_t, const float *mask_t, const int *mask_id,
                                                  int windows, int windows_per_image, int heads, int tokens, float scale, void *out,
                                                  void *stream)
{
    if (!window_attention_supported(windows, heads, tokens, HD) ||
        !rows_ok(rowmap, windows, windows_per_image, rows_per_image, pads_per_image, tokens))
        return MSDA_ERR_BAD_SHAPE;
    if (!qkv || !bias_t || !out || ((mask_t == nullptr) != (mask_id == nullptr))) return MSDA_ERR_NULL_POINTER;
    if (!aligned16(qkv) || !aligned16(out) || !aligned16(pad_row)) return MSDA_ERR_ALIGNMENT;
    const RowMap rm = {rowmap, rows_per_image, pads_per_image, windows_per_image};
    return launch_forward(qkv, pad_row, rm, bias_t, mask_t, mask_id, windows, heads, tokens, scale, out, stream);
}

extern "C" int window_attention_rows_backward_bf16(const void *qkv, const void *pad_row, const int *rowmap, int rows_per_image,
                                                   int pads_per_image, const void *d_out, const float *bias_t, const float *mask_t,
                                                   const int *mask_id, int windows, int windows_per_image, int heads, int tokens,
                                                   float scale, void *d_qkv, void *d_pad, void *stream)
{
    if (!window_attention_supported(windows, heads, tokens, HD) ||
        !rows_ok(rowmap, windows, windows_per_image, rows_per_image, pads_per_image, tokens))
        return MSDA_ERR_BAD_SHAPE;
    if (!qkv || !d_out || !bias_t || !d_qkv || ((mask_t == nullptr) != (mask_id == nullptr))) return MSDA_ERR_NULL_POINTER;
    if (pads_per_image > 0 && pad_row && !d_pad) return MSDA_ERR_NULL_POINTER;
    if (!aligned16(qkv) || !aligned16(d_out) || !aligned16(d_qkv) || !aligned16(pad_row) || !aligned16(d_pad)) return MSDA_ERR_ALIGNMENT;
    const RowMap rm = {rowmap, rows_per_image, pads_per_image, windows_per_image};
    return launch_backward(qkv, pad_row, rm, d_out, bias_t, mask_t, mask_id, windows, heads, tokens, scale, d_qkv, d_pad, stream);
}
