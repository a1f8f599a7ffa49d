// msda_sparse.hip -- grad_value of the MSDA backward pass for FEW queries (the decoders' cross-attention: Lq = 150 / 300
// box queries sampling the 22 223-pixel pyramid), gfx950, D = 32, L = 4, P = 4.  Same contract as msda_dest.hip: every
// row of grad_value has exactly one writer, no floating-point atomics, a fixed summation order.
//
// Why a second formulation: the tile-based destination pass pays ~9 000 cycles of fixed cost (work-queue pop, mask row,
// prefix, barriers) per (image, head, 16x8 tile) item; with 300 queries every one of the 6 432 items is non-empty but
// holds only ~8 (query, level) groups -- 105-145 us per call, 12 calls per train step.  Here ONE workgroup owns a whole
// (image, head, level): it has at most Lq * 16 corner records (4 800), which fit LDS together with the head's grad_out
// rows of all queries, so the records are counting-sorted by PIXEL in LDS (16-bit counters packed two per word, integer
// atomics; the waves scatter in turns, so the order inside a pixel's stretch is fixed) and each stretch is summed by a
// DPP quad straight from the staged grad_out rows.  grad_value is zero-filled first; only touched rows
// are written.
#include "msda_device.h"
#include "msda_internal.h"

namespace msda {

namespace {

constexpr int kL = 4, kP = 4, kD = 32;
constexpr int kThreads = 1024;
constexpr int kWaves = kThreads / 64;
constexpr int kLdsBudget = 160 * 1024;

struct SparsePlan {
    int H[kL], W[kL];
    int Lq;
    int off_rec, off_cnt, off_off, off_misc; // LDS byte offsets (grad_out rows at 0)
};

template <typename VT> struct Row;
template <> struct Row<bf16_t> { static constexpr int kBytes = 64; };
template <> struct Row<float> { static constexpr int kBytes = 128; };

struct Foot {
    int h_low, w_low;
    float lh, lw;
    bool inside;
};
__device__ __forceinline__ Foot footprint(float x, float y, int H, int W)
{
    Foot f;
    const float h_im = fmaf(y, (float)H, -0.5f), w_im = fmaf(x, (float)W, -0.5f);
    f.inside = (h_im > -1.f) && (w_im > -1.f) && (h_im < (float)H) && (w_im < (float)W);   // .cuh:285 (NaN -> false)
    const float hs = f.inside ? h_im : 0.f, ws = f.inside ? w_im : 0.f;
    const float hf = floorf(hs), wf = floorf(ws);
    f.h_low = (int)hf; f.w_low = (int)wf;
    f.lh = hs - hf; f.lw = ws - wf;
    return f;
}

__device__ __forceinline__ int wave_inclusive_scan(int v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int u = __shfl_up(v, off, 64);
        if (lane >= off) v += u;
    }
    return v;
}

template <typename OT> __device__ __forceinline__ void store_out8(OT *p, const float (&acc)[8]);
template <> __device__ __forceinline__ void store_out8<float>(float *p, const float (&acc)[8]) { Vec8<float>::store(p, acc); }
template <> __device__ __forceinline__ void store_out8<bf16_t>(bf16_t *p, const float (&acc)[8]) { Vec8<bf16_t>::store(p, acc); }

template <typename VT, typename OT>
__global__ __launch_bounds__(kThreads) void sparse_dest_kernel(SparsePlan pl, const int64_t *__restrict__ starts,
                                                               const float *__restrict__ loc, const float *__restrict__ aw,
                                                               const VT *__restrict__ grad_out, OT *__restrict__ g_value,
                                                               int N, int S, int M)
{
    MSDA_DYNAMIC_LDS(unsigned char, lds);
    constexpr int ROWB = Row<VT>::kBytes, PPR = ROWB / 16;
    unsigned char *grow = lds;
    uint2 *rec = reinterpret_cast<uint2 *>(lds + pl.off_rec);          // {record id, weight} sorted by pixel
    uint32_t *cnt = reinterpret_cast<uint32_t *>(lds + pl.off_cnt);     // two 16-bit counters per word
    uint32_t *off = reinterpret_cast<uint32_t *>(lds + pl.off_off);     // two 16-bit offsets per word
    int *misc = reinterpret_cast<int *>(lds + pl.off_misc);

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int NM = N * M;
    const int l = blockIdx.x / NM;                 // the largest level's workgroups start first
    const int nm = blockIdx.x % NM, n = nm / M, m = nm % M;
    const int H = pl.H[l], W = pl.W[l], npix = H * W, Lq = pl.Lq;
    const int words = (npix + 1) >> 1;

    // ---- phase A: this head's grad_out rows of all queries into LDS; counters cleared --------------------------------
    {
        const unsigned char *src = reinterpret_cast<const unsigned char *>(grad_out) + ((size_t)n * Lq * M + m) * ROWB;
        for (int i = tid; i < Lq * PPR; i += kThreads)
            *reinterpret_cast<uint4 *>(grow + i * 16) =
                *reinterpret_cast<const uint4 *>(src + (size_t)(i / PPR) * M * ROWB + (i % PPR) * 16);
        for (int i = tid; i < words; i += kThreads) cnt[i] = 0u;
    }
    __syncthreads();
    // ---- phase B: one thread per sample (at most two rounds): corners -> (pixel, weight), counted per pixel ------------
    int pix[2][4];
    float wgt[2][4];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int c = 0; c < 4; ++c) pix[r][c] = -1;
        const int s = tid + r * kThreads;
        if (s < Lq * kP) {
            const int q = s / kP, p = s % kP;
            const long sidx = ((((long)n * Lq + q) * M + m) * kL + l) * kP + p;
            const float2 xy = reinterpret_cast<const float2 *>(loc)[sidx];
            const float a = aw[sidx];
            const Foot f = footprint(xy.x, xy.y, H, W);
            const float hh = 1.f - f.lh, hw = 1.f - f.lw;
            const float cw[4] = {hh * hw * a, hh * f.lw * a, f.lh * hw * a, f.lh * f.lw * a};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int cy = f.h_low + (c >> 1), cx = f.w_low + (c & 1);
                if (f.inside && cy >= 0 && cy < H && cx >= 0 && cx < W) {
                    pix[r][c] = cy * W + cx;
                    wgt[r][c] = cw[c];
                    atomicAdd(&cnt[pix[r][c] >> 1], 1u << (16 * (pix[r][c] & 1)));
                }
            }
        }
    }
    __syncthreads();
    // ---- phase C: exclusive prefix of the counters over the pixels (each thread an even-sized contiguous chunk) -------
    {
        const int per = ((words + kThreads - 1) / kThreads);           // words per thread
        const int w0 = min(words, tid * per), w1 = min(words, w0 + per);
        int total = 0;
        for (int i = w0; i < w1; ++i) total += (int)(cnt[i] & 0xffffu) + (int)(cnt[i] >> 16);
        const int inc = wave_inclusive_scan(total);
        if (lane == 63) misc[wave] = inc;
        __syncthreads();
        int run = inc - total;
        for (int w = 0; w < wave; ++w) run += misc[w];
        for (int i = w0; i < w1; ++i) {
            const int c0 = (int)(cnt[i] & 0xffffu), c1 = (int)(cnt[i] >> 16);
            off[i] = (uint32_t)run | ((uint32_t)(run + c0) << 16);
            run += c0 + c1;
        }
    }
    __syncthreads();
    // ---- phase D: records to their pixel's stretch.  The waves take turns (16 short barrier-separated steps): the slot a
    // record gets inside its pixel's stretch is then wave-major, and inside a wave instruction the LDS atomics are served
    // in lane order -- a fixed order, so the sums below do not depend on timing (the same property msda_dest.hip's
    // per-wave histograms rely on) ------------------------------------------------------------------------------------
    for (int turn = 0; turn < kWaves; ++turn) {
        if (wave == turn) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (pix[r][c] >= 0) {
                        const int sh = 16 * (pix[r][c] & 1);
                        const uint32_t old = atomicAdd(&off[pix[r][c] >> 1], 1u << sh);
                        const int pos = (int)((old >> sh) & 0xffffu);
                        rec[pos] = make_uint2((uint32_t)((tid + r * kThreads) * 4 + c), __float_as_uint(wgt[r][c]));
                    }
        }
        __syncthreads();
    }
    // ---- phase E: a quad per touched pixel: sum its stretch, one 64 / 128-byte row out -----------------------
    {
        const int quad = tid >> 2, sub = tid & 3;
        const long pix0 = (long)n * S + (long)starts[l];
        for (int p = quad; p < npix; p += kThreads / 4) {
            const int sh = 16 * (p & 1);
            const int c = (int)((cnt[p >> 1] >> sh) & 0xffffu);
            if (c == 0) continue;
            const int end = (int)((off[p >> 1] >> sh) & 0xffffu), beg = end - c;
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int i = beg; i < end; ++i) {
                const uint2 r = rec[i];
                const int q = (int)(r.x >> 4);                          // id = (q * P + point) * 4 + corner
                Vec8<VT>::fma(__uint_as_float(r.y), Vec8<VT>::load_raw(reinterpret_cast<const VT *>(grow + q * ROWB) + sub * 8), acc);
            }
            store_out8<OT>(g_value + ((pix0 + p) * M + m) * kD + sub * 8, acc);
        }
    }
}

bool make_plan(const Problem &p, const int64_t *hs, SparsePlan &pl, bool f32)
{
    if (!hs || p.L != kL || p.P != kP || p.D != kD) return false;
    long sum = 0;
    int maxpix = 0;
    for (int l = 0; l < kL; ++l) {
        const int64_t H = hs[2 * l], W = hs[2 * l + 1];
        if (H < 1 || W < 1 || H * W >= 65536) return false;             // 16-bit pixel offsets / counters
        pl.H[l] = (int)H; pl.W[l] = (int)W;
        sum += H * W;
        maxpix = maxpix > (int)(H * W) ? maxpix : (int)(H * W);
    }
    if (sum != p.S) return false;
    if (p.Lq < 1 || p.Lq * kP > 2 * kThreads || p.Lq * kP * 4 >= 65536) return false;
    pl.Lq = p.Lq;
    const int rowb = f32 ? 128 : 64;
    pl.off_rec = (p.Lq * rowb + 15) / 16 * 16;
    pl.off_cnt = pl.off_rec + p.Lq * kP * 4 * 8;
    pl.off_off = pl.off_cnt + ((maxpix + 1) / 2) * 4;
    pl.off_misc = (pl.off_off + ((maxpix + 1) / 2) * 4 + 15) / 16 * 16;
    return pl.off_misc + 256 <= kLdsBudget;
}

}  // namespace

// few queries, the whole level's records of one (image, head) fit LDS
bool sparse_dest_supports(const Problem &p, const int64_t *shapes_host)
{
    if (p.dtype != MSDA_F32 && p.dtype != MSDA_BF16) return false;
    if ((long)p.N * p.S * p.M * kD >= (1L << 31)) return false;
    SparsePlan pl;
    return make_plan(p, shapes_host, pl, p.dtype == MSDA_F32);
}

void launch_sparse_dest(const Problem &p, const int64_t *shapes_host, bool out_bf16)
{
    SparsePlan pl;
    make_plan(p, shapes_host, pl, p.dtype == MSDA_F32);
    const int lds_bytes = pl.off_misc + 256;
    const size_t out_bytes = (size_t)p.N * p.S * p.M * kD * ((p.dtype == MSDA_F32 || !out_bf16) ? 4 : 2);
    (void)hipMemsetAsync(p.g_value, 0, out_bytes, p.stream);
    const dim3 grid(p.N * p.M * kL), block(kThreads);
#define MSDA_SPARSE(VT, OT)                                                                                          \
    do {                                                                                                             \
        RLIPV2_ONCE_PER_DEVICE((void)hipFuncSetAttribute((const void *)sparse_dest_kernel<VT, OT>,                   \
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBudget));   \
        hipLaunchKernelGGL((sparse_dest_kernel<VT, OT>), grid, block, lds_bytes, p.stream, pl, p.starts,             \
                           (const float *)p.loc, (const float *)p.aw, (const VT *)p.grad_out, (OT *)p.g_value, p.N,  \
                           p.S, p.M);                                                                                \
    } while (0)
    if (p.dtype == MSDA_F32) MSDA_SPARSE(float, float);
    else if (out_bf16) MSDA_SPARSE(bf16_t, bf16_t);
    else MSDA_SPARSE(bf16_t, float);
#undef MSDA_SPARSE
}

}  // namespace msda
