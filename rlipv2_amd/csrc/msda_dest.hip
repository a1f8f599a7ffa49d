// msda_dest.hip -- grad_value of the MSDA backward pass as a DESTINATION-stationary pass for gfx950
// (D = 32, L = 4, P = 4): every 128-byte row of grad_value is produced by exactly one workgroup, summed
// in registers in a fixed order and written once with a plain store -- no floating-point atomics, no
// zero-fill, bit-for-bit repeatable.  (Semantics: reference ms_deform_im2col_cuda.cuh:87-159, the
// `atomicAdd(grad_value + ptr, w * top_grad_value)` lines; the reference's result depends on atomic order.)
//
// Why: the source-stationary scatter of msda_window.hip (a query tile sorts its own records and flushes
// every touched row with a global float atomic) touched every row of grad_value ~14 times per call --
// 498 MB of read-modify-write traffic for a 91 MB tensor (profiles/r01_pmc_msda.json) -- and its walk spent
// ~13 instructions per record on ONE channel per lane.
//
// Three kernels (the pyramid shape is known on the host here: the caller passes a host copy of
// spatial_shapes, which also restores the reference's sum(H*W) == Len_in check, ms_deform_attn.py:96):
//
//   bin_kernel      one workgroup per (image, SOURCE tile of 16x16 queries, head): marks, per destination
//                   tile (16x16 pixels of a sampled level), WHICH of its 256 queries have a bilinear corner
//                   in that tile: a 256-bit mask per (destination tile, source tile), built with LDS
//                   integer ORs (order-independent) and written to the workspace.  Reads sampling_loc once.
//   dest_kernel     one workgroup of 1024 threads per (image, head, DESTINATION tile[, source part]):
//                   a DPP quad owns one pixel of the tile and 8 channels per lane in registers.  The set bits of
//                   the tile's mask row are enumerated in a fixed order (source tile, then query) in passes
//                   of 256 (query, level) groups; a pass = one thread per (group, point): locate the query
//                   (rank -> source tile by binary search of the popcount prefix, -> bit by select), load
//                   its location / weight / grad_out piece, stage the grad_out row in LDS, compute the 4
//                   corners, counting-sort the <= 4096 corner records by pixel (LDS integer atomics on
//                   PER-WAVE histograms, so the order inside a pixel's list is wave-major = deterministic),
//                   then every quad walks its pixel's list: per record one 8-byte read (weight, row), one
//                   16-byte read (8 channels of grad_out) and 8 FMAs.
//                   Coarse levels receive as many samples as fine ones on far fewer pixels; their tiles are
//                   split into `parts` by rank range and the partial tiles are summed by
//   combine_kernel  in part order (deterministic).
//
// Measured (MI355X, batch 4, 800x1333 pyramid, bf16): see DESIGN.md section 4 and profiles/r02_*.
#include <cstdio>
#include <cstdlib>

#include "msda_device.h"
#include "msda_internal.h"

#ifdef MSDA_DEST_TIMELINE     // cycle totals per phase of workgroup 0 (timeline builds only, tools/dest_timeline.py)
__device__ unsigned long long dest_ts[16];
#define DTS(k)                                                                           \
    do {                                                                                 \
        if (threadIdx.x == 0 && blockIdx.x == 0) {                                       \
            const unsigned long long t_ = clock64();                                     \
            dest_ts[k] += t_ - ts_last; ts_last = t_;                                    \
        }                                                                                \
    } while (0)
extern "C" int msda_debug_dest_timeline(void *host, int reset)
{
    if (reset) { unsigned long long z[16] = {0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(dest_ts), z, sizeof(z)); }
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(dest_ts), sizeof(dest_ts));
}
#else
#define DTS(k) do { } while (0)
#endif

namespace msda {

namespace {

constexpr int kL = 4, kP = 4, kD = 32;
constexpr int kTile = 16;                    // source tiles: 16 x 16 queries (one 256-bit mask); destination tiles: 16 x TH pixels
constexpr int kSrcQ = kTile * kTile;         // 256 queries per source tile
constexpr int kMaxSrcTiles = 512;            // mask row held in LDS
// destination tile of 16 x TH pixels = TH * 16 quads = TH waves; a pass sorts TH * 16 (query, level) groups
template <int TH> struct Geo {
    static constexpr int kPix = kTile * TH;
    static constexpr int kThreads = kPix * 4;
    static constexpr int kGroups = kPix;
    static constexpr int kWaves = kThreads / 64;
    static constexpr int kRec = kGroups * kP * 4;
};
constexpr int kPartSamples = 8192;           // target samples per (destination tile, part)
constexpr int kMaxParts = 32;

struct DestPlan {                            // by-value kernel argument, built from the host copy of spatial_shapes
    int H[kL], W[kL];
    int stx[kL], sbase[kL];                  // SOURCE tiles (16 x 16 queries) per row, first source tile of the level
    int th;                                  // destination tile height (8 or 16 pixels; width 16)
    int tx[kL], ty[kL];                      // destination tiles per row / column
    int tbase[kL];                           // first destination tile of the level (levels in order 0..L-1)
    int parts[kL];                           // source parts per destination tile of the level
    int ibase[kL], nitems[kL];               // items of the level in processing order (coarsest level first)
    int pbase[kL];                           // first partial slot of a split level (slot = tile * parts + part)
    int cbase[kL];                           // first combine block of a split level (one per tile), -1 if unsplit
    int Td, Ts;                              // destination tiles, source tiles
    int items, pslots, ctiles;
    int queues;                              // item queues: 8 (one per XCD) when N * M divides, else 1
    int tiled;                               // source tiles are 16x16 patches of the pyramid (Lq == S)
};

// ---- sample geometry shared by bin_kernel and dest_kernel (must agree bit for bit) --------------------------------
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

// source tile s -> first query and row pitch (tiled) ; untiled: 256 consecutive queries
__device__ __forceinline__ void source_tile(const DestPlan &pl, const int64_t *__restrict__ starts, int s, int &q0,
                                            int &pitch, int &rows, int &cols)
{
    if (pl.tiled) {
        int lq = 0;
#pragma unroll
        for (int l = 1; l < kL; ++l) lq = s >= pl.sbase[l] ? l : lq;
        const int t = s - pl.sbase[lq];
        const int sy0 = (t / pl.stx[lq]) * kTile, sx0 = (t % pl.stx[lq]) * kTile;
        pitch = pl.W[lq];
        q0 = (int)starts[lq] + sy0 * pitch + sx0;
        rows = min(kTile, pl.H[lq] - sy0);
        cols = min(kTile, pl.W[lq] - sx0);
    } else {
        q0 = s * kSrcQ; pitch = kTile; rows = kTile; cols = kTile;     // bit b -> query q0 + b (checked against Lq)
    }
}

// ------------------------------------------------------------------------------------------------------------------
// bin_kernel
// ------------------------------------------------------------------------------------------------------------------
// The body of one item = one (image, source tile, head) as a MACRO over the item index: bin_kernel (ITEM = blockIdx.x) compiles
// from exactly the tokens it always had -- its device code is pinned to a hardware run (tests/test_isa_manifest.py) --, the
// ablation build's bin_queue_kernel strides over the items with the same text.
#define MSDA_BIN_ITEM(ITEM) \
    const int tid = threadIdx.x;                                                                                      \
    const int m = (ITEM) % M;                                                                                         \
    const int s = ((ITEM) / M) % pl.Ts;                                                                               \
    const int n = (ITEM) / (M * pl.Ts);                                                                               \
    for (int i = tid; i < pl.Td * 8; i += 256) bmask[i] = 0u;                                                         \
    int q0, pitch, rows, cols;                                                                                        \
    source_tile(pl, starts, s, q0, pitch, rows, cols);                                                                \
    __syncthreads();                                                                                                  \
    /* 8 lanes read the 128 bytes of one (query, head): lane c holds points (2c & 3, +1) of level c / 2 */            \
    const int chunk = tid & 7, l = chunk >> 1;                                                                        \
    const int H = pl.H[l], W = pl.W[l], txl = pl.tx[l], tb = pl.tbase[l];                                             \
_Pragma("unroll 2")                                                                                                   \
    for (int pass = 0; pass < 8; ++pass) {                                                                            \
        const int ql = pass * 32 + (tid >> 3);                                                                        \
        const int qy = ql >> 4, qx = ql & 15;                                                                         \
        const int q = pl.tiled ? q0 + qy * pitch + qx : q0 + ql;                                                      \
        const bool live = pl.tiled ? (qy < rows && qx < cols) : (q < Lq);                                             \
        if (!live) continue;                                                                                          \
        const float4 v = reinterpret_cast<const float4 *>(loc)[(((long)n * Lq + q) * M + m) * 8 + chunk];             \
        const uint32_t bit = 1u << (ql & 31);                                                                         \
        const int word = ql >> 5;                                                                                     \
_Pragma("unroll")                                                                                                     \
        for (int k = 0; k < 2; ++k) {                                                                                 \
            const Foot f = footprint(k ? v.z : v.x, k ? v.w : v.y, H, W);                                             \
            if (!f.inside) continue;                                                                                  \
            /* the (up to) four tiles under the 2x2 footprint */                                                      \
            const int y0 = max(f.h_low, 0), y1 = min(f.h_low + 1, H - 1);                                             \
            const int x0 = max(f.w_low, 0), x1 = min(f.w_low + 1, W - 1);                                             \
            const int ta = pl.th == 8 ? y0 >> 3 : y0 >> 4, tb_ = pl.th == 8 ? y1 >> 3 : y1 >> 4, tc = x0 >> 4, td = x1 >> 4; \
            atomicOr(&bmask[(tb + ta * txl + tc) * 8 + word], bit);                                                   \
            if (td != tc) atomicOr(&bmask[(tb + ta * txl + td) * 8 + word], bit);                                     \
            if (tb_ != ta) {                                                                                          \
                atomicOr(&bmask[(tb + tb_ * txl + tc) * 8 + word], bit);                                              \
                if (td != tc) atomicOr(&bmask[(tb + tb_ * txl + td) * 8 + word], bit);                                \
            }                                                                                                         \
        }                                                                                                             \
    }                                                                                                                 \
    __syncthreads();                                                                                                  \
    /* masks[(n, m)][destination tile][source tile][8 words] */                                                       \
    const long nm = (long)n * M + m;                                                                                  \
    for (int i = tid; i < pl.Td * 8; i += 256)                                                                        \
        masks[((nm * pl.Td + (i >> 3)) * pl.Ts + s) * 8 + (i & 7)] = bmask[i];                                        \
    do { } while (0)
__global__ __launch_bounds__(256) void bin_kernel(DestPlan pl, const int64_t *__restrict__ starts,
                                                  const float *__restrict__ loc, int M, int Lq,
                                                  uint32_t *__restrict__ masks, const int *__restrict__ gate)
{
    MSDA_DYNAMIC_LDS_PLAIN(uint32_t, bmask);          // [Td][8]
    if (gate && *gate == 0) return;                   // the patch pass of msda_patch.hip has taken the call
    MSDA_BIN_ITEM(blockIdx.x);
}

#ifdef MSDA_ABLATION
// Arm RLIPV2_DEST_QUEUE (round 6): the gated launches of the sorting fallback with a FIXED grid whose
// workgroups stride over the items -- a launch that finds its gate word zero is <= 512 empty workgroups instead of one per item
// (13.7 k for the kernels of an N = 4 encoder call).  Same item bodies, same results.
__global__ __launch_bounds__(256) void bin_queue_kernel(DestPlan pl, const int64_t *__restrict__ starts,
                                                        const float *__restrict__ loc, int M, int Lq,
                                                        uint32_t *__restrict__ masks, const int *__restrict__ gate, int items)
{
    MSDA_DYNAMIC_LDS_PLAIN(uint32_t, bmask);
    if (gate && *gate == 0) return;
    for (int item = blockIdx.x; item < items; item += gridDim.x) {
        {
            MSDA_BIN_ITEM(item);
        }
        __syncthreads();                              // the table is zeroed again by the next item
    }
}
#endif

// ------------------------------------------------------------------------------------------------------------------
// dest_kernel
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int select_bit(uint32_t w, int k)      // position of the k-th (0-based) set bit
{
    int pos = 0;
#pragma unroll
    for (int sh = 16; sh >= 1; sh >>= 1) {
        const uint32_t low = w & ((1u << sh) - 1u);
        const int c = __popc(low);
        if (k >= c) { k -= c; w >>= sh; pos += sh; }
        else w = low;
    }
    return pos;
}

// inclusive prefix sum over the 64 lanes on DPP row shifts / broadcasts (gfx9 encodings; the ds_bpermute form of
// __shfl_up costs an LDS round trip per step, six steps per scan, two scans per pass)
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ int dpp_add(int v)
{
    // lanes whose source is outside the row / masked off keep `v + 0` (old = 0, bound_ctrl = false keeps `old`)
    return v + __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, BANK_MASK, false);
}

__device__ __forceinline__ int wave_inclusive_scan(int v)
{
    v = dpp_add<0x111, 0xf, 0xf>(v);           // row_shr:1
    v = dpp_add<0x112, 0xf, 0xf>(v);           // row_shr:2
    v = dpp_add<0x114, 0xf, 0xf>(v);           // row_shr:4
    v = dpp_add<0x118, 0xf, 0xf>(v);           // row_shr:8   -> inclusive within each row of 16
    v = dpp_add<0x142, 0xa, 0xf>(v);           // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xc, 0xf>(v);           // row_bcast:31 into rows 2 and 3
    return v;
}

template <typename VT> struct GRow;           // a staged grad_out row in LDS: 32 channels + 16 bytes of padding
template <> struct GRow<bf16_t> {
    static constexpr int kStride = 80;        // bytes
    static constexpr int kPiece = 16;         // bytes per lane (8 channels)
    typedef uint4 piece;
    static __device__ __forceinline__ piece load_global(const bf16_t *p) { return *reinterpret_cast<const uint4 *>(p); }
    static __device__ __forceinline__ void store_lds(unsigned char *p, const piece &v) { *reinterpret_cast<uint4 *>(p) = v; }
    static __device__ __forceinline__ void fma_piece(float w, const uint4 &r, float (&acc)[8])
    {
        acc[0] = fmaf(w, bf16_lo(r.x), acc[0]); acc[1] = fmaf(w, bf16_hi(r.x), acc[1]);
        acc[2] = fmaf(w, bf16_lo(r.y), acc[2]); acc[3] = fmaf(w, bf16_hi(r.y), acc[3]);
        acc[4] = fmaf(w, bf16_lo(r.z), acc[4]); acc[5] = fmaf(w, bf16_hi(r.z), acc[5]);
        acc[6] = fmaf(w, bf16_lo(r.w), acc[6]); acc[7] = fmaf(w, bf16_hi(r.w), acc[7]);
    }
    static __device__ __forceinline__ void fma(float w, const unsigned char *p, float (&acc)[8])
    {
        fma_piece(w, *reinterpret_cast<const uint4 *>(p), acc);
    }
};
template <> struct GRow<float> {
    static constexpr int kStride = 144;
    static constexpr int kPiece = 32;
    struct piece { float4 a, b; };
    static __device__ __forceinline__ piece load_global(const float *p)
    {
        piece v;
        v.a = *reinterpret_cast<const float4 *>(p);
        v.b = *reinterpret_cast<const float4 *>(p + 4);
        return v;
    }
    static __device__ __forceinline__ void store_lds(unsigned char *p, const piece &v)
    {
        *reinterpret_cast<float4 *>(p) = v.a;
        *reinterpret_cast<float4 *>(p + 16) = v.b;
    }
    static __device__ __forceinline__ void fma_piece(float w, const piece &r, float (&acc)[8])
    {
        const float4 a = r.a, b = r.b;
        acc[0] = fmaf(w, a.x, acc[0]); acc[1] = fmaf(w, a.y, acc[1]); acc[2] = fmaf(w, a.z, acc[2]);
        acc[3] = fmaf(w, a.w, acc[3]); acc[4] = fmaf(w, b.x, acc[4]); acc[5] = fmaf(w, b.y, acc[5]);
        acc[6] = fmaf(w, b.z, acc[6]); acc[7] = fmaf(w, b.w, acc[7]);
    }
    static __device__ __forceinline__ void fma(float w, const unsigned char *p, float (&acc)[8])
    {
        fma_piece(w, *reinterpret_cast<const piece *>(p), acc);
    }
};

// LDS carve-up (bytes; every offset a multiple of 16)
template <typename VT, int TH> struct DestLds {
    typedef Geo<TH> G;
    static constexpr int kOffG = 0;                                           // [2 pass parities][groups] staged rows
    static constexpr int kGBytes = G::kGroups * GRow<VT>::kStride;
    static constexpr int kOffRec = kOffG + 2 * kGBytes;
    static constexpr int kOffHist = kOffRec + G::kRec * 8;                    // int [2 pass parities][waves][pixels]
    static constexpr int kHistInts = G::kWaves * G::kPix;
    static constexpr int kOffPix = kOffHist + 2 * kHistInts * 4;              // int pixoff[pixels], tot[pixels]
    static constexpr int kOffMisc = kOffPix + 2 * G::kPix * 4;                // int [32]: wave sums, item, ...
    static constexpr int kOffSrc = kOffMisc + 128;                      // per source tile: mask[8], off, q0, pitch
    static int bytes(int Ts) { return kOffSrc + Ts * 32 + (Ts + 4) * 4 * 3; }
};

template <typename OT> __device__ __forceinline__ void store_out8(OT *p, const float (&acc)[8]);
template <> __device__ __forceinline__ void store_out8<float>(float *p, const float (&acc)[8]) { Vec8<float>::store(p, acc); }
template <> __device__ __forceinline__ void store_out8<bf16_t>(bf16_t *p, const float (&acc)[8]) { Vec8<bf16_t>::store(p, acc); }

template <typename VT, typename OT, int TH, int WAVES>
__global__ __launch_bounds__(Geo<TH>::kThreads, WAVES) void dest_kernel(
    DestPlan pl, const int64_t *__restrict__ starts, const float *__restrict__ loc, const float *__restrict__ aw,
    const VT *__restrict__ grad_out, const uint32_t *__restrict__ masks, int *__restrict__ counter,
    OT *__restrict__ g_value, float *__restrict__ partials, int N, int S, int M, int Lq, const int *__restrict__ gate)
{
    MSDA_DYNAMIC_LDS(unsigned char, lds);
    if (gate && *gate == 0) return;                   // the patch pass of msda_patch.hip has taken the call
    typedef DestLds<VT, TH> LD;
    typedef Geo<TH> G;
    constexpr int kPix = G::kPix, kThreads = G::kThreads, kGroups = G::kGroups, kWaves = G::kWaves;
    static_assert(kThreads * 16 >= 2 * kWaves * kPix * 4, "hist is cleared with one 16-byte store per thread");
    // Staged rows and per-wave histograms are double-buffered by pass parity: a wave that has finished walking pass c
    // starts phase 1 of pass c + 1 (stage rows, histogram atomics) while slower quads still walk -- the walk needs no
    // closing barrier.  (rec / pixoff / tot are rewritten only behind the next pass's first barrier.)
    unsigned char *gl0 = lds + LD::kOffG;
    uint2 *rec = reinterpret_cast<uint2 *>(lds + LD::kOffRec);
    int *hist0 = reinterpret_cast<int *>(lds + LD::kOffHist);
    int *pixoff = reinterpret_cast<int *>(lds + LD::kOffPix);
    int *tot = pixoff + kPix;
    int *misc = reinterpret_cast<int *>(lds + LD::kOffMisc);          // [0..15] wave sums, [16] item
    uint32_t *mrow = reinterpret_cast<uint32_t *>(lds + LD::kOffSrc);
    int *off = reinterpret_cast<int *>(mrow + pl.Ts * 8);             // [Ts + 1]
    int *sq0 = off + pl.Ts + 4;
    int *spitch = sq0 + pl.Ts + 4;

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int grp = tid >> 2, sub = tid & 3;                           // pass: (group, point); walk: (pixel, channel octet)
    const int NM = N * M;
    const int Ts = pl.Ts;

    unsigned long long ts_last = clock64();
    (void)ts_last;
    // Item queues.  pl.queues == 8: one queue per XCD (workgroup b runs on XCD b % 8) holding the items of NM / 8
    // consecutive (image, head) pairs -- neighbouring destination tiles of one (image, head) share most of their source
    // rows (a query's 4 points of a level straddle ~2.5 tiles), and with ONE queue those re-reads were spread over all
    // eight L2s (every XCD saw all N * M slices: 80 MB of live rows against 4 MB of L2).  A workgroup that finds its
    // queue empty helps with the next XCD's (the tail only).
    const int nq = pl.queues, NMq = NM / nq;
    int queue = (int)(blockIdx.x % (unsigned)nq), tried = 0;
    for (;;) {
        if (tid == 0) misc[16] = atomicAdd(counter + queue * 4, 1);
        __syncthreads();
        DTS(0);
        const int item = misc[16];
        if (item >= pl.items / nq) {
            if (++tried >= nq) break;
            queue = (queue + 1) % nq;
            __syncthreads();                 // (everybody has read misc[16])
            continue;
        }

        // ---- decode the item: destination level (coarsest first), tile, part, (image, head) -----------------------
        int l = 0;
#pragma unroll
        for (int k = 1; k < kL; ++k) l = (item >= pl.ibase[k] / nq && item < (pl.ibase[k] + pl.nitems[k]) / nq) ? k : l;
        const int local = item - pl.ibase[l] / nq;
        const int nm = queue * NMq + local % NMq;
        const int rest = local / NMq;
        const int nparts = pl.parts[l];
        const int part = rest % nparts, d = rest / nparts;
        const int n = nm / M, m = nm % M;
        const int H = pl.H[l], W = pl.W[l];
        const int ty0 = (d / pl.tx[l]) * TH, tx0 = (d % pl.tx[l]) * kTile;

        // ---- the tile's mask row, popcount prefix over the source tiles --------------------------------------------
        {
            const uint32_t *mg = masks + ((size_t)((long)nm * pl.Td + pl.tbase[l] + d) * Ts) * 8;
            for (int i = tid; i < Ts * 8; i += kThreads) mrow[i] = mg[i];
            if (tid * 4 < 2 * kWaves * kPix) reinterpret_cast<int4 *>(hist0)[tid] = make_int4(0, 0, 0, 0);
        }
        __syncthreads();
        {
            int c = 0;
            if (tid < Ts) {
#pragma unroll
                for (int i = 0; i < 8; ++i) c += __popc(mrow[tid * 8 + i]);
                int q0, pitch, rows, cols;
                source_tile(pl, starts, tid, q0, pitch, rows, cols);
                sq0[tid] = q0; spitch[tid] = pitch;
            }
            const int inc = wave_inclusive_scan(c);
            if (lane == 63) misc[wave] = inc;
            __syncthreads();
            int base = 0;
            for (int w = 0; w < wave; ++w) base += misc[w];
            if (tid < Ts) off[tid] = base + inc - c;
            if (tid == Ts - 1) off[Ts] = base + inc;
        }
        __syncthreads();
        DTS(1);
        const int R = off[Ts];
        const int r_begin = (int)((long)R * part / nparts), r_end = (int)((long)R * (part + 1) / nparts);

        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const long nq_base = (long)n * Lq;
        // Passes take the tile's groups with a STRIDE (pass c = ranks c, c + npass, c + 2 npass, ...): consecutive ranks
        // come from one source tile and land in one corner of the destination tile (a few quads would walk long lists
        // while the others idle -- measured 9 300 of 23 000 cycles per pass waiting for the slowest wave); a strided pass
        // draws from every source tile and spreads over all 256 pixels.  Still a fixed order.
        const int cnt = r_end - r_begin;
        const int npass = (cnt + kGroups - 1) / kGroups;

        // What a thread needs from global memory for one pass, loaded a whole pass ahead.  A thread's ranks in
        // successive passes are consecutive (c + grp * npass), so after one binary search + bit select it walks the
        // mask row with a cursor: next set bit of the current word, else the next non-zero word (register work, an
        // LDS read now and then) -- the per-pass search was ~2 000 cycles of dependent LDS latency.
        struct Fetch {
            float2 xy;
            float a;
            typename GRow<VT>::piece gp;
            bool valid;
        };
        int cu_lo = 0, cu_wi = 0, cu_bit = 0, cu_r = 0, cu_hi = 0;
        uint32_t cu_rem = 0u;
        auto load = [&](bool valid) {
            Fetch f;
            f.valid = valid;
            f.xy = make_float2(0.f, 0.f); f.a = 0.f; f.gp = typename GRow<VT>::piece();
            if (valid) {
                const int b = cu_wi * 32 + cu_bit;
                const int q = pl.tiled ? sq0[cu_lo] + (b >> 4) * spitch[cu_lo] + (b & 15) : sq0[cu_lo] + b;
                const long qm = (nq_base + q) * M + m;
                const long sidx = (qm * kL + l) * kP + sub;
                f.xy = reinterpret_cast<const float2 *>(loc)[sidx];
                f.a = aw[sidx];
                f.gp = GRow<VT>::load_global(grad_out + qm * kD + sub * 8);
            }
            return f;
        };
        const bool v0 = grp * npass < cnt;
        if (v0) {
            const int r = r_begin + grp * npass;
            int lo = 0, hi = Ts;                       // off[lo] <= r < off[hi]
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (off[mid] <= r) lo = mid; else hi = mid;
            }
            int k = r - off[lo];
            uint32_t ww = 0u;
            bool found = false;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const uint32_t w_i = mrow[lo * 8 + i];
                const int c_ = __popc(w_i);
                if (!found) {
                    if (k < c_) { found = true; cu_wi = i; ww = w_i; }
                    else k -= c_;
                }
            }
            cu_lo = lo; cu_r = r; cu_hi = off[lo + 1];
            cu_bit = select_bit(ww, k);
            cu_rem = ww & ~((2u << cu_bit) - 1u);     // the bits above the current one
        }
        Fetch nxt = load(v0);

        for (int c0 = 0; c0 < npass; ++c0) {
            // ---- pass, phase 1: one thread per (group, point) ------------------------------------------------------
            unsigned char *gl = gl0 + (c0 & 1) * LD::kGBytes;
            int *hist = hist0 + (c0 & 1) * LD::kHistInts;
            const Fetch me = nxt;
#ifdef MSDA_DEST_TIMELINE
            MSDA_ASM_WAIT_VM();
            DTS(12);
#endif
            {   // the next pass's operands start travelling now
                const bool vn = c0 + 1 < npass && c0 + 1 + grp * npass < cnt;
                if (vn) {
                    ++cu_r;
                    if (cu_r >= cu_hi) {                       // source tile exhausted: locate the next non-empty one
                        int lo = cu_lo + 1, hi = Ts;           // off[lo] <= cu_r < off[hi]
                        while (hi - lo > 1) {
                            const int mid = (lo + hi) >> 1;
                            if (off[mid] <= cu_r) lo = mid; else hi = mid;
                        }
                        cu_lo = lo; cu_hi = off[lo + 1]; cu_wi = -1; cu_rem = 0u;
                    }
                    while (cu_rem == 0u) cu_rem = mrow[cu_lo * 8 + (++cu_wi)];      // at most 8 steps
                    cu_bit = __ffs(cu_rem) - 1;
                    cu_rem &= cu_rem - 1u;
                }
                nxt = load(vn);
            }
            DTS(13);
            int rank[4] = {0, 0, 0, 0}, key[4] = {-1, -1, -1, -1};
            float cw[4] = {0.f, 0.f, 0.f, 0.f};
            if (me.valid) {
                GRow<VT>::store_lds(gl + grp * GRow<VT>::kStride + sub * GRow<VT>::kPiece, me.gp);
                const Foot f = footprint(me.xy.x, me.xy.y, H, W);
                const float wgt = f.inside ? me.a : 0.f;
                const float hh = 1.f - f.lh, hw = 1.f - f.lw;
                cw[0] = hh * hw * wgt; cw[1] = hh * f.lw * wgt; cw[2] = f.lh * hw * wgt; cw[3] = f.lh * f.lw * wgt;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int cy = f.h_low + (c >> 1), cx = f.w_low + (c & 1);
                    const int py = cy - ty0, px = cx - tx0;
                    const bool ok = f.inside && cy >= 0 && cy < H && cx >= 0 && cx < W &&
                                    (unsigned)py < (unsigned)TH && (unsigned)px < (unsigned)kTile;
                    if (ok) {
                        key[c] = py * kTile + px;
                        rank[c] = atomicAdd(&hist[wave * kPix + key[c]], 1);
                    }
                }
            }
            DTS(2);
            __syncthreads();
            DTS(3);
            // ---- phase 2: per pixel, prefix over the waves; prefix over the pixels -----------------------------------
            // (the other parity's histograms were last read before the previous pass's third barrier: clear them for the
            //  next pass here, ordered before anybody's next phase 1 by this pass's remaining barriers)
            {
                int *nh = hist0 + ((c0 + 1) & 1) * LD::kHistInts;
                if (tid * 4 < kWaves * kPix) reinterpret_cast<int4 *>(nh)[tid] = make_int4(0, 0, 0, 0);
            }
            if (tid < kPix) {
                int cnt[kWaves];                                   // all the loads first: one LDS round trip, not kWaves
#pragma unroll
                for (int w = 0; w < kWaves; ++w) cnt[w] = hist[w * kPix + tid];
                int run = 0;
#pragma unroll
                for (int w = 0; w < kWaves; ++w) {
                    hist[w * kPix + tid] = run;
                    run += cnt[w];
                }
                tot[tid] = run;
                const int inc = wave_inclusive_scan(run);
                pixoff[tid] = inc - run;
                if (lane == 63) misc[wave] = inc;
            }
            DTS(4);
            __syncthreads();
            DTS(5);
            // ---- phase 3: records to their sorted positions ----------------------------------------------------------
            const int ws0 = misc[0], ws1 = ws0 + misc[1], ws2 = kPix > 128 ? ws1 + misc[2] : 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (key[c] >= 0) {
                    const int kw = key[c] >> 6;
                    const int wpre = kw == 0 ? 0 : kw == 1 ? ws0 : kw == 2 ? ws1 : ws2;
                    const int pos = hist[wave * kPix + key[c]] + rank[c] + pixoff[key[c]] + wpre;
                    rec[pos] = make_uint2(__float_as_uint(cw[c]), (unsigned)(grp * GRow<VT>::kStride));
                }
            }
            DTS(6);
            __syncthreads();
            DTS(7);
            // ---- phase 4: every quad walks the list of its pixel; the next pass's operands travel meanwhile ----------
            {
                const int kw = grp >> 6;
                const int wpre = kw == 0 ? 0 : kw == 1 ? ws0 : kw == 2 ? ws1 : ws2;
                int i = pixoff[grp] + wpre;
                const int e = i + tot[grp];
                const unsigned char *gsub = gl + sub * GRow<VT>::kPiece;
                for (; i + 3 < e; i += 4) {
                    const uint2 ra = rec[i], rb = rec[i + 1], rc = rec[i + 2], rd = rec[i + 3];
                    GRow<VT>::fma(__uint_as_float(ra.x), gsub + ra.y, acc);
                    GRow<VT>::fma(__uint_as_float(rb.x), gsub + rb.y, acc);
                    GRow<VT>::fma(__uint_as_float(rc.x), gsub + rc.y, acc);
                    GRow<VT>::fma(__uint_as_float(rd.x), gsub + rd.y, acc);
                }
                if (i < e) {
                    // the last 1-3 records in ONE round of reads (a record-at-a-time tail is two dependent LDS round trips
                    // per record); same order of additions, absent records skipped
                    const int n_left = e - i;
                    const uint2 ra = rec[i], rb = rec[min(i + 1, e - 1)], rc = rec[min(i + 2, e - 1)];
                    typename GRow<VT>::piece pa = *reinterpret_cast<const typename GRow<VT>::piece *>(gsub + ra.y);
                    typename GRow<VT>::piece pb = *reinterpret_cast<const typename GRow<VT>::piece *>(gsub + rb.y);
                    typename GRow<VT>::piece pc = *reinterpret_cast<const typename GRow<VT>::piece *>(gsub + rc.y);
                    GRow<VT>::fma_piece(__uint_as_float(ra.x), pa, acc);
                    if (n_left > 1) GRow<VT>::fma_piece(__uint_as_float(rb.x), pb, acc);
                    if (n_left > 2) GRow<VT>::fma_piece(__uint_as_float(rc.x), pc, acc);
                }
            }
            DTS(8);
            DTS(9);                          // (no barrier: see the double-buffering note at the top of the kernel)
#ifdef MSDA_DEST_TIMELINE
            if (tid == 0 && blockIdx.x == 0) dest_ts[15] += 1;
#endif
        }
        DTS(10);

        // ---- the tile's rows leave once -------------------------------------------------------------------------------
        const int py = grp >> 4, px = grp & 15;
        if (nparts == 1) {
            const int y = ty0 + py, x = tx0 + px;
            if (y < H && x < W) {
                const long pix = (long)n * S + (long)starts[l] + (long)y * W + x;
                store_out8<OT>(g_value + (pix * M + m) * kD + sub * 8, acc);
            }
        } else {
            float *dst = partials + (((size_t)(pl.pbase[l] + d * nparts + part) * NM + nm) * kPix + grp) * kD + sub * 8;
            Vec8<float>::store(dst, acc);
        }
        DTS(11);
#ifdef MSDA_DEST_TIMELINE
        if (tid == 0 && blockIdx.x == 0) dest_ts[14] += 1;
#endif
    }
}

// sums the partial tiles of a split level in part order and writes the rows
// (the item body as a macro over the item index, for the same reason as MSDA_BIN_ITEM: combine_kernel keeps its tokens)
#define MSDA_COMBINE_ITEM(ITEM) \
    const int NM = N * M;                                                                                             \
    const int nm = (ITEM) % NM, t = (ITEM) / NM;                                                                      \
    int l = 0;                                                                                                        \
_Pragma("unroll")                                                                                                     \
    for (int k = 0; k < kL; ++k) l = (pl.cbase[k] >= 0 && t >= pl.cbase[k] && t < pl.cbase[k] + pl.tx[k] * pl.ty[k]) ? k : l; \
    const int d = t - pl.cbase[l];                                                                                    \
    const int nparts = pl.parts[l];                                                                                   \
    const int grp = threadIdx.x >> 2, sub = threadIdx.x & 3;                                                          \
    constexpr int kPix = Geo<TH>::kPix;                                                                               \
    const int y = (d / pl.tx[l]) * TH + (grp >> 4), x = (d % pl.tx[l]) * kTile + (grp & 15);                          \
    if (y >= pl.H[l] || x >= pl.W[l]) return;                                                                         \
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};                                                          \
    for (int j = 0; j < nparts; ++j) {                                                                                \
        float v[8];                                                                                                   \
        Vec8<float>::load(partials + (((size_t)(pl.pbase[l] + d * nparts + j) * NM + nm) * kPix + grp) * kD + sub * 8, v); \
_Pragma("unroll")                                                                                                     \
        for (int k = 0; k < 8; ++k) acc[k] += v[k];                                                                   \
    }                                                                                                                 \
    const int n = nm / M, m = nm % M;                                                                                 \
    const long pix = (long)n * S + (long)starts[l] + (long)y * pl.W[l] + x;                                           \
    store_out8<OT>(g_value + (pix * M + m) * kD + sub * 8, acc);                                                      \
    do { } while (0)
template <typename OT, int TH>
__global__ __launch_bounds__(Geo<TH>::kThreads) void combine_kernel(DestPlan pl, const int64_t *__restrict__ starts,
                                                           const float *__restrict__ partials,
                                                           OT *__restrict__ g_value, int N, int S, int M,
                                                           const int *__restrict__ gate)
{
    if (gate && *gate == 0) return;
    MSDA_COMBINE_ITEM(blockIdx.x);
}

#ifdef MSDA_ABLATION
template <typename OT, int TH>
__global__ __launch_bounds__(Geo<TH>::kThreads) void combine_queue_kernel(DestPlan pl, const int64_t *__restrict__ starts,
                                                                 const float *__restrict__ partials,
                                                                 OT *__restrict__ g_value, int N, int S, int M,
                                                                 const int *__restrict__ gate, int items)
{
    if (gate && *gate == 0) return;
    for (int item = blockIdx.x; item < items; item += gridDim.x) {
        auto one = [&]() __attribute__((always_inline)) { MSDA_COMBINE_ITEM(item); };    // (the body returns for pixels outside the level)
        one();
    }
}
#endif

// ---- host side -------------------------------------------------------------------------------------------------------
constexpr int kDestTH = 8;                   // destination tile height in use
constexpr int kDestLdsMax = 96 * 1024;       // dynamic LDS bound of bin_kernel / dest_kernel (2 workgroups per CU stay possible)

bool make_plan(const Problem &p, const int64_t *hs, DestPlan &pl)
{
    if (!hs || p.L != kL) return false;
    long sum = 0;
    int tb = 0, sb = 0;
    pl.th = kDestTH;
    for (int l = 0; l < kL; ++l) {
        const int64_t H = hs[2 * l], W = hs[2 * l + 1];
        if (H < 1 || W < 1 || H >= (1 << 20) || W >= (1 << 20)) return false;
        pl.H[l] = (int)H; pl.W[l] = (int)W;
        pl.tx[l] = (int)((W + kTile - 1) / kTile); pl.ty[l] = (int)((H + pl.th - 1) / pl.th);
        pl.tbase[l] = tb;
        tb += pl.tx[l] * pl.ty[l];
        pl.stx[l] = pl.tx[l]; pl.sbase[l] = sb;
        sb += pl.stx[l] * (int)((H + kTile - 1) / kTile);
        sum += H * W;
    }
    if (sum != p.S) return false;
    pl.Td = tb;
    pl.tiled = p.Lq == p.S;
    pl.Ts = pl.tiled ? sb : (p.Lq + kSrcQ - 1) / kSrcQ;
    const int NM = p.N * p.M;
    int items = 0, pslots = 0, ctiles = 0;
    for (int l = kL - 1; l >= 0; --l) {
        const int T = pl.tx[l] * pl.ty[l];
        const double est = (double)p.Lq * kP / T;
        int parts = (int)((est + kPartSamples - 1) / kPartSamples);
        parts = parts < 1 ? 1 : parts > kMaxParts ? kMaxParts : parts;
        pl.parts[l] = parts;
        pl.ibase[l] = items;
        pl.nitems[l] = T * parts * NM;
        items += pl.nitems[l];
        if (parts > 1) {
            pl.pbase[l] = pslots; pslots += T * parts;
            pl.cbase[l] = ctiles; ctiles += T;
        } else {
            pl.pbase[l] = 0; pl.cbase[l] = -1;
        }
    }
    pl.items = items; pl.pslots = pslots; pl.ctiles = ctiles;
    static const int xcd_queues = ablation_env("RLIPV2_DEST_XCD", 1);
    pl.queues = (xcd_queues && NM % 8 == 0) ? 8 : 1;
    return true;
}

constexpr size_t kCtlBytes = 256;
size_t mask_bytes(const Problem &p, const DestPlan &pl) { return (size_t)p.N * p.M * pl.Td * pl.Ts * 32; }
size_t partial_bytes(const Problem &p, const DestPlan &pl)
{
    return (size_t)pl.pslots * p.N * p.M * Geo<kDestTH>::kPix * kD * 4;
}

}  // namespace

bool dest_supports(const Problem &p, const int64_t *shapes_host)
{
    if (p.dtype != MSDA_F32 && p.dtype != MSDA_BF16) return false;
    if (p.D != kD || p.L != kL || p.P != kP) return false;
    if (!quad_supports(p)) return false;                       // K1 is the quad reduce kernel
    DestPlan pl;
    if (!make_plan(p, shapes_host, pl)) return false;
    // LDS: bin_kernel's table is Td * 32 bytes, dest_kernel's carve-up grows with Ts (67 KB for float32 at the 800x1333
    // pyramid); both are asked for with hipFuncSetAttribute at launch and bounded here -- larger pyramids fall back to the
    // sorted scatter (msda_window.hip) instead of failing in the middle of a backward pass
    if (pl.Ts > kMaxSrcTiles || pl.Td > 3072) return false;
    if (DestLds<float, kDestTH>::bytes(pl.Ts) > kDestLdsMax || DestLds<bf16_t, kDestTH>::bytes(pl.Ts) > kDestLdsMax) return false;
    if ((long)pl.items >= (1L << 30)) return false;
    if (mask_bytes(p, pl) > ((size_t)1 << 31)) return false;
    return true;
}

int dest_shapes_consistent(const Problem &p, const int64_t *shapes_host)
{
    if (!shapes_host) return 1;
    long sum = 0;
    for (int l = 0; l < p.L; ++l) sum += shapes_host[2 * l] * shapes_host[2 * l + 1];
    return sum == p.S;
}

static size_t round16(size_t v) { return (v + 15) & ~(size_t)15; }

size_t dest_workspace_bytes(const Problem &p, const int64_t *shapes_host)
{
    DestPlan pl;
    if (!dest_supports(p, shapes_host) || !make_plan(p, shapes_host, pl)) return 0;
    // [control block | tile masks | partial tiles | patch masks of msda_patch.hip (encoder calls, bfloat16)]
    return round16(kCtlBytes + mask_bytes(p, pl) + partial_bytes(p, pl)) + patch_workspace_bytes(p, shapes_host);
}

// K1 + grad_value.  out_bf16: grad_value is bfloat16.
void launch_backward_dest(const Problem &p_in, const Fused *f, const int64_t *shapes_host, void *workspace, bool out_bf16,
                          const void *records, bool records_swap)
{
    Problem p = p_in;                                        // (the records route may point loc / aw at rebuilt copies)
    auto k1 = [&]() { if (f) launch_quad_backward_reduce_fused(p, *f); else launch_quad_backward_reduce(p); };
    if (!records && sparse_dest_supports(p, shapes_host) && ablation_env("RLIPV2_MSDA_SPARSE", 1)) {
        k1();
        launch_sparse_dest(p, shapes_host, out_bf16);      // few queries: per-(image, head, level) pass, see msda_sparse.hip
        return;
    }
    DestPlan pl;
    make_plan(p, shapes_host, pl);
    unsigned char *ws = reinterpret_cast<unsigned char *>(workspace);
    int *counter = reinterpret_cast<int *>(ws);
    uint32_t *masks = reinterpret_cast<uint32_t *>(ws + kCtlBytes);
    float *partials = reinterpret_cast<float *>(ws + kCtlBytes + mask_bytes(p, pl));
    if (hipMemsetAsync(counter, 0, kCtlBytes, p.stream) != hipSuccess) return;   // (the error stays recorded: the ABI reports MSDA_ERR_LAUNCH)
    // Encoder calls with bfloat16 gradients (msda_patch.hip): cell_backward_kernel does K1's work from LDS-resident
    // windows and bins the samples, the matrix-core patch pass produces grad_value; the kernels below then find the gate
    // word zero and return at once.  Only when the binning met a sample outside its cell's reach (gate != 0: the patch
    // pass has returned without writing) does the sorting pass run.
    const int *gate = nullptr;
    void *pws = ws + round16(kCtlBytes + mask_bytes(p, pl) + partial_bytes(p, pl));
    if (records) {
        // the "records" route (msda_cell_records.inc): the forward pass has left every sample's geometry, the patch masks and
        // the group records; nothing is binned here
        const size_t gco = patch_gcell_offset(p, shapes_host);          // (ablation build: the CELLG arm's copy; else 0)
        gate = launch_cell_records_backward(p, f, shapes_host, records, out_bf16, records_swap, gco ? static_cast<unsigned char *>(pws) + gco : nullptr);
        if (!p.loc) {
            // the forward did not save float32 locations / weights: the sorting pass below -- if the gate lets it run at all --
            // reads copies rebuilt from the group records, in the part of the workspace the binning of the product route uses
            // (patch_workspace_bytes >= N * Lq * M * 16 samples * 12 bytes: a group record holds exactly these floats)
            float *loc = reinterpret_cast<float *>(pws);
            float *aw = loc + (size_t)p.N * p.Lq * p.M * kL * kP * 2;
            launch_records_unbin(p, shapes_host, records, loc, aw, gate);
            p.loc = loc; p.aw = aw;
        }
    } else if (patch_workspace_bytes(p, shapes_host) > 0 && ablation_env("RLIPV2_MSDA_PATCH", 1)) {
        int *ctl = counter;
        const bool cell = cell_backward_supports(p, shapes_host) && ablation_env("RLIPV2_MSDA_CELL", 1);
        if (cell) launch_cell_backward(p, f, shapes_host, ctl, pws);
        else k1();
#ifdef MSDA_ABLATION
        // arm RLIPV2_CELL_FAR_RETURN: the cell kernel's workgroups stop once a far sample has been seen; K1, gated on the same
        // word, then writes every gradient of the locations / weights (an empty launch on every other call)
        if (cell && ablation_env("RLIPV2_CELL_FAR_RETURN", 0)) launch_quad_backward_gated(p, f, ctl + 60);
#endif
        const size_t gco = patch_gcell_offset(p, shapes_host);          // (ablation build: the CELLG arm's copy; else 0)
        launch_patch_dest(p, shapes_host, ctl, pws, out_bf16, cell, gco ? static_cast<unsigned char *>(pws) + gco : nullptr);
        gate = ctl + 60;
    } else {
        k1();
    }
    RLIPV2_ONCE_PER_DEVICE((void)hipFuncSetAttribute((const void *)bin_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kDestLdsMax));
#ifdef MSDA_ABLATION
    const bool queue = gate != nullptr && ablation_env("RLIPV2_DEST_QUEUE", 0) != 0;      // arm: fixed grids striding over the items
    if (queue) {
        RLIPV2_ONCE_PER_DEVICE((void)hipFuncSetAttribute((const void *)bin_queue_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kDestLdsMax));
        const int items = p.N * pl.Ts * p.M;
        hipLaunchKernelGGL(bin_queue_kernel, dim3(items < 512 ? items : 512), dim3(256), pl.Td * 32, p.stream, pl, p.starts,
                           (const float *)p.loc, p.M, p.Lq, masks, gate, items);
    } else
#else
    const bool queue = false;
#endif
    hipLaunchKernelGGL(bin_kernel, dim3(p.N * pl.Ts * p.M), dim3(256), pl.Td * 32, p.stream, pl, p.starts,
                       (const float *)p.loc, p.M, p.Lq, masks, gate);
    constexpr int TH = kDestTH, kThreads = Geo<TH>::kThreads;
    static const int per_cu = ablation_env("RLIPV2_DEST_WGS", 2);
    static const int waves = ablation_env("RLIPV2_DEST_WAVES", 4);
    const int grid = pl.items < 256 * per_cu ? pl.items : 256 * per_cu;
#define MSDA_LAUNCH_DEST(VT, OT)                                                                                     \
    do {                                                                                                             \
        const int lds_bytes = DestLds<VT, TH>::bytes(pl.Ts);                                                         \
        auto kern = waves == 8 ? dest_kernel<VT, OT, TH, 8> : waves == 6 ? dest_kernel<VT, OT, TH, 6> : dest_kernel<VT, OT, TH, 4>; \
        RLIPV2_ONCE_PER_DEVICE(                       /* once per instantiation and device, not inside every capture */ \
            (void)hipFuncSetAttribute((const void *)dest_kernel<VT, OT, TH, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, kDestLdsMax); \
            (void)hipFuncSetAttribute((const void *)dest_kernel<VT, OT, TH, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, kDestLdsMax); \
            (void)hipFuncSetAttribute((const void *)dest_kernel<VT, OT, TH, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, kDestLdsMax)); \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), lds_bytes, p.stream, pl, p.starts, (const float *)p.loc, \
                           (const float *)p.aw, (const VT *)p.grad_out, masks, counter, (OT *)p.g_value, partials,  \
                           p.N, p.S, p.M, p.Lq, gate);                                                               \
    } while (0)
    if (p.dtype == MSDA_F32) MSDA_LAUNCH_DEST(float, float);
    else if (out_bf16) MSDA_LAUNCH_DEST(bf16_t, bf16_t);
    else MSDA_LAUNCH_DEST(bf16_t, float);
#undef MSDA_LAUNCH_DEST
    if (pl.ctiles > 0) {
        const int cgrid = pl.ctiles * p.N * p.M;
#ifdef MSDA_ABLATION
        if (queue) {
            const int qgrid = cgrid < 512 ? cgrid : 512;
            if (p.dtype == MSDA_BF16 && out_bf16)
                hipLaunchKernelGGL((combine_queue_kernel<bf16_t, TH>), dim3(qgrid), dim3(kThreads), 0, p.stream, pl, p.starts, partials,
                                   (bf16_t *)p.g_value, p.N, p.S, p.M, gate, cgrid);
            else
                hipLaunchKernelGGL((combine_queue_kernel<float, TH>), dim3(qgrid), dim3(kThreads), 0, p.stream, pl, p.starts, partials,
                                   (float *)p.g_value, p.N, p.S, p.M, gate, cgrid);
            return;
        }
#endif
        (void)queue;
        if (p.dtype == MSDA_BF16 && out_bf16)
            hipLaunchKernelGGL((combine_kernel<bf16_t, TH>), dim3(cgrid), dim3(kThreads), 0, p.stream, pl, p.starts, partials,
                               (bf16_t *)p.g_value, p.N, p.S, p.M, gate);
        else
            hipLaunchKernelGGL((combine_kernel<float, TH>), dim3(cgrid), dim3(kThreads), 0, p.stream, pl, p.starts, partials,
                               (float *)p.g_value, p.N, p.S, p.M, gate);
    }
}

}  // namespace msda
