// token_gemm.hip -- weight / bias gradient of a Linear applied to token-major activations
// (include/rlipv2_linear.h):  dW[M,K] = dY[T,M]^T X[T,K],  db[M] = sum_t dY[t,:],  bf16 in, f32 accumulate.
//
// Shape: T = 88 892 tokens, M, K in {256, 384, 1024}.  Arithmetic intensity M*K/(M+K) flop/byte = 128..205,
// below the MI355X ridge (2.5 PFLOP/s / 8 TB/s = 312), so the bound is HBM: dY and X are streamed once.
//
// Layout of one workgroup (256 threads = 4 waves, 2x2 waves of 64x64 outputs = one 128x128 tile of dW):
//   * a chunk of consecutive tokens [r0, r1) is walked 32 rows at a time; each thread carries 2 x 16 bytes
//     of the dY tile and 2 x 16 bytes of the X tile in registers (issued one step ahead), and stores them
//     row-major into LDS with a 320-byte row pitch;
//   * both MFMA operands need 8 consecutive-k (= token) values of one column per lane, i.e. the transpose
//     of what memory holds; gfx950's ds_read_b64_tr_b16 delivers exactly that: a 16-lane group reads a
//     [4 tokens][16 columns] block and every lane receives one column.  Since dY and X are read with the
//     same lane->token map, the k order inside an MFMA is consistent by construction.  The 320-byte pitch
//     puts the 4 rows x 2 column blocks of a 32-lane half on 64 distinct banks;
//   * v_mfma_f32_32x32x16_bf16, 2x2 per wave per 16 tokens, 64 accumulator VGPRs;
//   * the per-chunk 128x128 partial goes to the workspace with plain stores (fp32 atomics to a shared dW
//     measured 3x slower than stores + one pass of `reduce_partials`, profiles/r01_ubench_global_atomics.txt);
//   * bias gradient: the threads of the tile_k == 0 workgroups add up the dY registers they stage anyway.
// Workgroups that share a chunk are placed on the same XCD (they re-read the chunk through that XCD's L2).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rlipv2_linear.h"
#include "../../include/rlipv2_msda.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int BM = 128, BN = 128, BK = 32, THREADS = 256;
constexpr int ROW_BYTES = 320;                 // 128 bf16 + 64 bytes of skew
constexpr int TILE_BYTES = BK * ROW_BYTES;     // one operand, one stage
constexpr int STAGE_BYTES = 2 * TILE_BYTES;
constexpr int LDS_BYTES = 2 * STAGE_BYTES;     // 40 960
constexpr int NUM_XCD = 8;

__device__ __forceinline__ float bf16_to_float(uint32_t bits16) { return __uint_as_float(bits16 << 16); }

__device__ __forceinline__ bf16x8 read_fragment(const char *lane_base)
{
    union { s16x4 h[2]; bf16x8 v; } u;
    u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(lane_base));
    u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(lane_base + 4 * ROW_BYTES));
    return u.v;
}

__global__ __launch_bounds__(THREADS, 2) void wgrad_kernel(const uint16_t *__restrict__ dy,
                                                           const uint16_t *__restrict__ x, int T, int M, int K,
                                                           int rows_per_chunk, int chunks, float *__restrict__ partial,
                                                           float *__restrict__ bias_partial)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tiles_k = K / BN, tiles = (M / BM) * tiles_k;
    const int total = chunks * tiles;
    // physical workgroup b runs on XCD b % 8: give every XCD a contiguous range of logical ids so that
    // the `tiles` workgroups of one chunk share an L2
    const int per_xcd = (total + NUM_XCD - 1) / NUM_XCD;
    const int logical = (blockIdx.x % NUM_XCD) * per_xcd + blockIdx.x / NUM_XCD;
    if (logical >= total) return;
    const int chunk = logical / tiles, tile = logical % tiles;
    const int tm = tile / tiles_k, tk = tile % tiles_k;
    const int r0 = chunk * rows_per_chunk;
    const int r1 = min(T, r0 + rows_per_chunk);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // staging role: 16 threads per 256-byte tile row, rows srow and srow + 16
    const int srow = tid >> 4, spiece = tid & 15;
    const uint16_t *a_src = dy + (size_t)tm * BM + spiece * 8;
    const uint16_t *b_src = x + (size_t)tk * BN + spiece * 8;
    const int st_off = srow * ROW_BYTES + spiece * 16;
    // fragment role: 16-lane group g reads rows (lane>>5)*8 + (p>>2) (+4), columns 16*(g&1) + 4*(p&3)
    const int p = lane & 15, g = lane >> 4;
    const int frag_off = ((lane >> 5) * 8 + (p >> 2)) * ROW_BYTES + (16 * (g & 1) + 4 * (p & 3)) * 2;
    const char *a_frag = smem + frag_off + wm * 128;
    const char *b_frag = smem + TILE_BYTES + frag_off + wn * 128;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float bsum[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bsum[j] = 0.f;
    const bool want_bias = (tk == 0) && bias_partial != nullptr;

    uint4 ra[2], rb[2];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = k0 + srow + 16 * i;
            if (row < r1) {
                ra[i] = *reinterpret_cast<const uint4 *>(a_src + (size_t)row * M);
                rb[i] = *reinterpret_cast<const uint4 *>(b_src + (size_t)row * K);
            } else {
                ra[i] = make_uint4(0, 0, 0, 0);
                rb[i] = make_uint4(0, 0, 0, 0);
            }
        }
    };
    auto lstore = [&](int stage) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            char *d = smem + stage * STAGE_BYTES + st_off + i * 16 * ROW_BYTES;
            *reinterpret_cast<uint4 *>(d) = ra[i];
            *reinterpret_cast<uint4 *>(d + TILE_BYTES) = rb[i];
        }
        if (want_bias) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const uint32_t w[4] = {ra[i].x, ra[i].y, ra[i].z, ra[i].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    bsum[2 * j] += bf16_to_float(w[j] & 0xffffu);
                    bsum[2 * j + 1] += bf16_to_float(w[j] >> 16);
                }
            }
        }
    };

    if (r0 < r1) {
        gload(r0);
        lstore(0);
    }
    __syncthreads();
    int stage = 0;
    for (int k0 = r0; k0 < r1; k0 += BK, stage ^= 1) {
        const bool more = k0 + BK < r1;
        if (more) gload(k0 + BK);
        const int sb = stage * STAGE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[2], bfr[2];
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                af[f] = read_fragment(a_frag + sb + ks * 16 * ROW_BYTES + f * 64);
                bfr[f] = read_fragment(b_frag + sb + ks * 16 * ROW_BYTES + f * 64);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        if (more) lstore(stage ^ 1);
        __syncthreads();
    }

    // partial[chunk][m][k]: accumulator register r of lane l is row 8*(r/4) + 4*(l/32) + r%4, column l%32
    float *out = partial + ((size_t)chunk * M + (size_t)tm * BM + wm * 64) * K + (size_t)tk * BN + wn * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
                out[(size_t)row * K + j * 32 + (lane & 31)] = acc[i][j][r];
            }

    if (want_bias) {
        float *red = reinterpret_cast<float *>(smem);          // [16 row groups][128 columns]
#pragma unroll
        for (int j = 0; j < 8; ++j) red[srow * BM + spiece * 8 + j] = bsum[j];
        __syncthreads();
        if (tid < BM) {
            float s = 0.f;
#pragma unroll
            for (int rg = 0; rg < 16; ++rg) s += red[rg * BM + tid];
            bias_partial[(size_t)chunk * M + tm * BM + tid] = s;
        }
    }
}

// out[e] = sum_c partial[c][e]; four elements per thread; bf16 (round-to-nearest-even) or f32 output
template <bool F32>
__global__ __launch_bounds__(256) void reduce_partials(const float *__restrict__ partial, int chunks, size_t n,
                                                       void *__restrict__ out)
{
    const size_t e = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (e >= n) return;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
    for (int c = 0; c < chunks; ++c) {
        const float4 v = *reinterpret_cast<const float4 *>(partial + (size_t)c * n + e);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
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

struct Plan { int chunks, rows_per_chunk, tiles; size_t partial_floats, bias_floats; };

bool make_plan(int T, int M, int K, Plan &pl)
{
    if (T < 1 || M < BM || K < BN || M % BM || K % BN) return false;
    pl.tiles = (M / BM) * (K / BN);
    // two workgroups per CU (512 in flight); a chunk is a whole number of 32-token steps
    static int target = [] { const char *e = getenv("RLIPV2_WGRAD_BLOCKS"); return e ? atoi(e) : 512; }();
    int chunks = (target + pl.tiles - 1) / pl.tiles;
    const int steps = (T + BK - 1) / BK;
    if (chunks > steps) chunks = steps;
    if (chunks < 1) chunks = 1;
    const int steps_per_chunk = (steps + chunks - 1) / chunks;
    pl.rows_per_chunk = steps_per_chunk * BK;
    pl.chunks = (T + pl.rows_per_chunk - 1) / pl.rows_per_chunk;
    pl.partial_floats = (size_t)pl.chunks * M * K;
    pl.bias_floats = (size_t)pl.chunks * M;
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
    float *partial = static_cast<float *>(workspace);
    float *bias_partial = db ? partial + pl.partial_floats : nullptr;
    const int total = pl.chunks * pl.tiles;
    const int grid = ((total + NUM_XCD - 1) / NUM_XCD) * NUM_XCD;
    hipLaunchKernelGGL(wgrad_kernel, dim3(grid), dim3(THREADS), LDS_BYTES, stream,
                       static_cast<const uint16_t *>(dy), static_cast<const uint16_t *>(x), T, M, K,
                       pl.rows_per_chunk, pl.chunks, partial, bias_partial);
    const size_t n = (size_t)M * K;
    if (out_f32) {
        hipLaunchKernelGGL(reduce_partials<true>, dim3((n / 4 + 255) / 256), dim3(256), 0, stream, partial, pl.chunks, n, dw);
        if (db) hipLaunchKernelGGL(reduce_partials<true>, dim3(((size_t)M / 4 + 255) / 256), dim3(256), 0, stream,
                                   bias_partial, pl.chunks, (size_t)M, db);
    } else {
        hipLaunchKernelGGL(reduce_partials<false>, dim3((n / 4 + 255) / 256), dim3(256), 0, stream, partial, pl.chunks, n, dw);
        if (db) hipLaunchKernelGGL(reduce_partials<false>, dim3(((size_t)M / 4 + 255) / 256), dim3(256), 0, stream,
                                   bias_partial, pl.chunks, (size_t)M, db);
    }
    return hipGetLastError() == hipSuccess ? MSDA_OK : MSDA_ERR_LAUNCH;
}
