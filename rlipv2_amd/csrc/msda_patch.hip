// msda_patch.hip -- grad_value of the encoder's MSDA backward pass (Lq == S, bfloat16 grad_out, D = 32, L = P = 4)
// as a wave-autonomous, destination-stationary pass on the matrix cores of gfx950.
// (Semantics: reference ms_deform_im2col_cuda.cuh:87-159, the `atomicAdd(grad_value + ptr, w * top_grad_value)`
//  lines of ms_deform_attn_col2im_bilinear; the reference's result depends on atomic order, this one does not.)
//
// Why (round 3): the workgroup-cooperative pass of msda_dest.hip routes every bilinear corner record to the DPP quad
// that owns its pixel through an LDS counting sort -- four barriers and ~10 dependent LDS round trips per pass of
// 2 048 records, 35 % VALU utilisation, 385 us per batch-4 encoder call.  Here the routing IS the arithmetic:
//
//   grad_value[pix, ch] = sum over (query, level) groups g of  A[pix, g] * G[g, ch]
//   A[pix, g] = sum over the group's 4 points of  attn * tent(x - pix_x) * tent(y - pix_y),   tent(d) = max(0, 1 - |d|)
//
// (the bilinear corner weights of .cuh:33-84 are exactly the tent products at the two neighbouring pixel centres).
// A wave owns a PATCH of 4 x 4 pixels of one (image, head, level).  It walks the exact list of groups that touch
// the patch (bit masks written by bin2_kernel), 32 groups per step: lane (g, half) loads the group's two points of
// that half, evaluates the separable tents for the patch's 4 columns and 4 rows (48 multiplies for 16 pixels), splits
// the 16 weights into bfloat16 hi + lo (relative error 2^-16 per weight, below the bfloat16 rounding of grad_out and of
// the bfloat16 result by 2^7) and writes them as a row of the K x 16 matrix `A^T` in LDS next to the group's
// grad_out row; ds_read_b64_tr_b16 turns both row-major images into MFMA operands and
// v_mfma_f32_16x16x32_bf16 accumulates D^T[ch, pix] += G^T[ch, g] A^T[g, pix] -- eight MFMAs per 32 groups.
// No sort, no atomics, no lists per pixel, no workgroup barrier inside the loop: waves are independent, five per
// SIMD hide each other's latencies.  Every row of grad_value has exactly one writer and a fixed summation order:
// bit-for-bit repeatable, no zero-fill.
//
// Candidate lists.  Queries are grouped into CELLS: the pyramid column over a 16 x 16 block of level-0 pixels, i.e.
// the queries (iy, ix) of level lq with (iy >> (4 - lq), ix >> (4 - lq)) == (cy, cx) -- at most 256 + 64 + 16 + 4 = 340
// queries, one bit each.  A patch looks at a fixed NEIGHBOURHOOD of cells around its own position (radius 1 / 2 / 3 /
// 6 cells at level 0 / 1 / 2 / 3: every sample within >= 12 pixels of its query's own pyramid position); for every
// (patch, neighbourhood slot) bin2_kernel stores the 340-bit mask of that cell's queries with a corner in the patch.
// A sample that lands outside its cell's reach ("far": uniform random locations, degenerate pyramids) raises a flag
// and the whole call falls back to the sorting pass of msda_dest.hip, which has no such assumption.
#include <algorithm>

#include "msda_device.h"
#include "msda_geometry.h"
#include "msda_internal.h"

namespace msda {

namespace {

constexpr int kL = 4, kP = 4, kD = 32;
constexpr int kCellQ = 340;                   // queries (bits) per cell: 256 + 64 + 16 + 4
constexpr int kSlotWords = 12;                // 11 mask words + 1 of padding: 48 bytes per (patch, slot)
constexpr int kStep = 32;                     // groups per MFMA step (K of v_mfma_f32_16x16x32_bf16)
constexpr int kListCap = 384;                 // entries: < 32 carried over + one cell's <= 340
constexpr int kWaves = 4;                     // waves per workgroup (each owns a patch, or a part of one)
constexpr int kThreads = kWaves * 64;
// per-wave LDS: candidate list | staged grad_out rows [32][32 ch] bf16 | A^T [half][hi/lo][32][16 px] bf16
constexpr int kOffList = 0, kOffG = kListCap * 2, kOffA = kOffG + kStep * 64, kWaveLds = kOffA + 4 * kStep * 32;
static_assert(kOffG % 16 == 0 && kOffA % 16 == 0 && kWaveLds % 16 == 0, "16-byte carve-up");
constexpr int kFarWord = 60;                  // int index of the "far sample seen" flag in the control block

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

struct PatchPlan {                            // by-value kernel argument, built from the host copy of spatial_shapes
    int H[kL], W[kL];
    int CY, CX;                               // cells
    int PY[kL], PX[kL];                       // patches per level
    int rad[kL];                              // neighbourhood radius (cells)
    int nby[kL], nbx[kL];                     // neighbourhood extent (cells), always inside the cell grid
    int invx[kL];                             // ceil(65536 / nbx): slot / nbx == (slot * invx) >> 16 for slot < 128
    int sbase[kL];                            // first mask slot of the level, per (image, head)
    int slots;                                // mask slots per (image, head)
    int parts[kL];                            // waves per patch (1, 2 or 4)
    int reps[kL];                             // patches per wave, one after the other (only where parts == 1)
    int ibase[kL], nitems[kL];                // workgroups of the level per (image, head); coarsest level first
    int items;
    int bin_lds;                              // bytes of bin2_kernel's LDS table (maximum over the cells)
};

// first cell row / column of the neighbourhood of patch row / column `pp` of level l
__host__ __device__ inline int nb_origin(int l, int pp, int rad, int nb, int cells)
{
    const int home = (pp * 4) >> (4 - l);
    const int o = home - rad;
    return o < 0 ? 0 : (o > cells - nb ? cells - nb : o);
}

// The patch rows (columns) of level l whose neighbourhood contains cell row (column) c, in closed form: nb_origin is
// non-decreasing in the patch index, so they are the interval [first t with origin(t) > c - nb, last t with origin(t) <= c].
// Returns {lo, hi}; lo > hi: none.  (Checked against the enumeration for every level, grid size, radius and patch count.)
__host__ __device__ inline void nb_range(int l, int c, int rad, int nb, int cells, int P, int &lo, int &hi)
{
    const int s = 4 - l, v = c - nb + 1;
    lo = v <= 0 ? 0 : v > cells - nb ? P : ((((v + rad) << s) + 3) >> 2);
    hi = c >= cells - nb ? P - 1 : ((((c + rad + 1) << s) + 3) >> 2) - 1;
    hi = hi < P - 1 ? hi : P - 1;
}

// ------------------------------------------------------------------------------------------------------------------
// bin2_kernel: one workgroup per (image, cell, head) -- which of the cell's queries touch which patch
// ------------------------------------------------------------------------------------------------------------------
#ifdef MSDA_ABLATION
__device__ unsigned long long cell_ts[16];      // cycle sums over all workgroups (thread 0): phases of cell_backward_kernel
#define CTS(k) do { if (threadIdx.x == 0) { const unsigned long long t_ = clock64(); atomicAdd(&cell_ts[k], t_ - ts_last); ts_last = t_; } } while (0)
#else
#define CTS(k) do { } while (0)
#endif
// The binning of one cell's queries, shared by bin2_kernel and cell_backward_kernel.  `tab` = the workgroup's LDS table
// (PatchPlan::bin_lds bytes), `rng` = [4][4] LDS ints.  Ends with a barrier; table_layout() then gives every thread the
// table's carve-up and write_masks() sends it to the workspace.
struct TableLayout { int base[kL], ylo[kL], xlo[kL], ph[kL], pw[kL], total; };

__device__ __forceinline__ TableLayout table_layout(const int (*rng)[4])
{
    TableLayout t;
    t.total = 0;
#pragma unroll
    for (int l = 0; l < kL; ++l) {
        const int r0 = __builtin_amdgcn_readfirstlane(rng[l][0]), r1 = __builtin_amdgcn_readfirstlane(rng[l][1]);
        const int r2 = __builtin_amdgcn_readfirstlane(rng[l][2]), r3 = __builtin_amdgcn_readfirstlane(rng[l][3]);
        t.ylo[l] = r0; t.xlo[l] = r2;
        t.ph[l] = max(r1 - r0 + 1, 0); t.pw[l] = max(r3 - r2 + 1, 0);
        if (t.ph[l] == 0 || t.pw[l] == 0) { t.ph[l] = 0; t.pw[l] = 0; }
        t.base[l] = t.total;
        t.total += t.ph[l] * t.pw[l] * kSlotWords;
    }
    return t;
}

// BBOX: also the bounding box of every in-level corner per sampled level: box[l] = {x0, y0, -x1, -y1} (LDS, via min)
// SCALAR_STARTS (experiment): level_start_index read as four scalars and selected per lane -- `starts[lq]` with a per-lane
// lq is a vector load whose result the NEXT load's address needs: the "loads up front" below were in fact chained by up to
// four full memory latencies (s_waitcnt vmcnt(0) before the dependent address, visible in the device assembly).
template <int THREADS, bool BBOX, bool SCALAR_STARTS = false>
__device__ __forceinline__ void bin_cell(const PatchPlan &pl, const int64_t *__restrict__ starts,
                                         const float *__restrict__ loc, const float *__restrict__ aw, int n, int m, int c,
                                         int M, int Lq, uint32_t *tab, int (*rng)[4], int (*box)[4],
                                         float *__restrict__ recs, int *__restrict__ ctl, int dbg = 0)
{
    const int tid = threadIdx.x;
    const int cells = pl.CY * pl.CX;
    const int cy = c / pl.CX, cx = c % pl.CX;
    unsigned long long ts_last = clock64();
    (void)ts_last;
    if (SCALAR_STARTS) {
        // (experiment, with the scalar level starts: the ranges below in closed form -- wave-uniform scalar arithmetic with
        //  compile-time level indices, no loops over the patch rows / columns, no LDS atomics, one barrier less; thread 0 writes)
#pragma unroll
        for (int l = 0; l < kL; ++l) {
            int ylo, yhi, xlo, xhi;
            nb_range(l, cy, pl.rad[l], pl.nby[l], pl.CY, pl.PY[l], ylo, yhi);
            nb_range(l, cx, pl.rad[l], pl.nbx[l], pl.CX, pl.PX[l], xlo, xhi);
            if (tid == 0) {
                rng[l][0] = ylo <= yhi ? ylo : (1 << 30); rng[l][1] = ylo <= yhi ? yhi : -1;
                rng[l][2] = xlo <= xhi ? xlo : (1 << 30); rng[l][3] = xlo <= xhi ? xhi : -1;
            }
        }
        if (BBOX && tid >= 32 && tid < 48) box[(tid - 32) >> 2][tid & 3] = 0x3fffffff;
    } else {
    if (tid < 16) rng[tid >> 2][tid & 3] = (tid & 1) ? -1 : (1 << 30);
    if (BBOX && tid >= 32 && tid < 48) box[(tid - 32) >> 2][tid & 3] = 0x3fffffff;
    __syncthreads();
    // the patches whose neighbourhood contains this cell: a contiguous range of rows and of columns per level
#pragma unroll
    for (int l = 0; l < kL; ++l) {
        for (int t = tid; t < pl.PY[l]; t += THREADS) {
            const int o = nb_origin(l, t, pl.rad[l], pl.nby[l], pl.CY);
            if (o <= cy && cy < o + pl.nby[l]) { atomicMin(&rng[l][0], t); atomicMax(&rng[l][1], t); }
        }
        for (int t = tid; t < pl.PX[l]; t += THREADS) {
            const int o = nb_origin(l, t, pl.rad[l], pl.nbx[l], pl.CX);
            if (o <= cx && cx < o + pl.nbx[l]) { atomicMin(&rng[l][2], t); atomicMax(&rng[l][3], t); }
        }
    }
    }
    __syncthreads();
    if (BBOX) CTS(8);
    const TableLayout tl = table_layout(rng);
    for (int i = tid; i < tl.total; i += THREADS) tab[i] = 0u;
    __syncthreads();
    if (BBOX) CTS(9);

    // 8 lanes read the 128 bytes of one (query, head): lane chunk holds points (2 chunk & 3, +1) of level chunk / 2
    const int chunk = tid & 7, l = chunk >> 1;
    auto sel = [l](const int (&v)[kL]) { return l == 0 ? v[0] : l == 1 ? v[1] : l == 2 ? v[2] : v[3]; };   // (no scratch)
    const int H = sel(pl.H), W = sel(pl.W);
    const int tb = sel(tl.base), y_lo = sel(tl.ylo), x_lo = sel(tl.xlo), hh = sel(tl.ph), ww = sel(tl.pw);
    bool far = false;
    int bx0 = 0x3fffffff, by0 = 0x3fffffff, bx1 = -0x3fffffff, by1 = -0x3fffffff;
    // All of the thread's loads first (every pass of a plain loop waited 3-4 us for its 24 bytes: cycle stamps in
    // tools/cell_timeline.py), then the arithmetic and the LDS ORs.
    constexpr int IT = (kCellQ * 8 + THREADS - 1) / THREADS;
    // (readfirstlane: opaque scalars -- the compiler otherwise folds the select chain back into the indexed vector load)
    const int sst0 = SCALAR_STARTS ? __builtin_amdgcn_readfirstlane((int)starts[0]) : 0;
    const int sst1 = SCALAR_STARTS ? __builtin_amdgcn_readfirstlane((int)starts[1]) : 0;
    const int sst2 = SCALAR_STARTS ? __builtin_amdgcn_readfirstlane((int)starts[2]) : 0;
    const int sst3 = SCALAR_STARTS ? __builtin_amdgcn_readfirstlane((int)starts[3]) : 0;
    float4 vv[IT];
    float2 av[IT];
    bool have[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int j = (tid >> 3) + it * (THREADS / 8);
        const int lq = j < 256 ? 0 : j < 320 ? 1 : j < 336 ? 2 : 3;
        const int r = j - (lq == 0 ? 0 : lq == 1 ? 256 : lq == 2 ? 320 : 336);
        const int sh = 4 - lq;
        const int iy = (cy << sh) + (r >> sh), ix = (cx << sh) + (r & ((1 << sh) - 1));
        const int Hq = lq == 0 ? pl.H[0] : lq == 1 ? pl.H[1] : lq == 2 ? pl.H[2] : pl.H[3];
        const int Wq = lq == 0 ? pl.W[0] : lq == 1 ? pl.W[1] : lq == 2 ? pl.W[2] : pl.W[3];
        have[it] = j < kCellQ && iy < Hq && ix < Wq;
        const int stq = !SCALAR_STARTS ? (int)starts[lq] : lq == 0 ? sst0 : lq == 1 ? sst1 : lq == 2 ? sst2 : sst3;
        const int q = have[it] ? stq + iy * Wq + ix : 0;
        const long qm = ((long)n * Lq + q) * M + m;
        vv[it] = reinterpret_cast<const float4 *>(loc)[qm * 8 + chunk];
        av[it] = reinterpret_cast<const float2 *>(aw)[qm * 8 + chunk];
    }
#ifdef MSDA_ABLATION
    if (BBOX && (MSDA_DBG(dbg) & 16)) { MSDA_ASM_WAIT_VM(); CTS(10); }
#endif
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        if (!have[it]) continue;
        const int j = (tid >> 3) + it * (THREADS / 8);
        const float4 v = vv[it];
        if (!(MSDA_DBG(dbg) & 32)) {   // the group's record for the patch pass: [x0 y0 x1 y1 | x2 y2 x3 y3 | a0 a1 a2 a3], cell-major
            float *rec = recs + ((((size_t)(n * M + m) * kL + l) * cells + c) * kCellQ + j) * 12;
            reinterpret_cast<float4 *>(rec)[chunk & 1] = v;
            reinterpret_cast<float2 *>(rec + 8)[chunk & 1] = av[it];
        }
        const uint32_t bit = 1u << (j & 31);
        const int word = j >> 5;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const float x = k ? v.z : v.x, y = k ? v.w : v.y;
            const float h_im = fmaf(y, (float)H, -0.5f), w_im = fmaf(x, (float)W, -0.5f);
            const bool inside = (h_im > -1.f) && (w_im > -1.f) && (h_im < (float)H) && (w_im < (float)W);   // .cuh:285
            if (!inside) continue;
            const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
            const int y0 = max(h_low, 0), y1 = min(h_low + 1, H - 1), x0 = max(w_low, 0), x1 = min(w_low + 1, W - 1);
            if (BBOX) { bx0 = min(bx0, x0); by0 = min(by0, y0); bx1 = max(bx1, x1); by1 = max(by1, y1); }
            const int ya = y0 >> 2, yb = y1 >> 2, xa = x0 >> 2, xb = x1 >> 2;
            auto mark = [&](int py, int px) {
                const int ry = py - y_lo, rx = px - x_lo;
                if ((unsigned)ry < (unsigned)hh && (unsigned)rx < (unsigned)ww)
                    atomicOr(&tab[tb + (ry * ww + rx) * kSlotWords + word], bit);
                else
                    far = true;
            };
            mark(ya, xa);
            if (xb != xa) mark(ya, xb);
            if (yb != ya) {
                mark(yb, xa);
                if (xb != xa) mark(yb, xb);
            }
        }
    }
    if (BBOX && bx0 <= bx1) {
        int *b = &box[0][0] + l * 4;
        atomicMin(b + 0, bx0); atomicMin(b + 1, by0); atomicMin(b + 2, -bx1); atomicMin(b + 3, -by1);
    }
    if (far) atomicOr(ctl + kFarWord, 1);
    if (BBOX) CTS(11);
    __syncthreads();
    if (BBOX) CTS(12);
}

// masks[(n, m)][slot of (patch, this cell)][12 words]: every slot of the buffer has exactly one writer
template <int THREADS>
__device__ __forceinline__ void write_masks(const PatchPlan &pl, const TableLayout &tl, const uint32_t *tab, int n, int m,
                                            int c, int M, uint32_t *__restrict__ masks)
{
    const int tid = threadIdx.x;
    const int cy = c / pl.CX, cx = c % pl.CX;
    const long nm = (long)n * M + m;
#pragma unroll
    for (int lv = 0; lv < kL; ++lv) {
        const int rows = tl.ph[lv] * tl.pw[lv], nb2 = pl.nby[lv] * pl.nbx[lv];
        for (int i = tid; i < rows * 3; i += THREADS) {
            const int row = i / 3, piece = i - row * 3;
            const int py = tl.ylo[lv] + row / tl.pw[lv], px = tl.xlo[lv] + row % tl.pw[lv];
            const int oy = nb_origin(lv, py, pl.rad[lv], pl.nby[lv], pl.CY);
            const int ox = nb_origin(lv, px, pl.rad[lv], pl.nbx[lv], pl.CX);
            const int k = (cy - oy) * pl.nbx[lv] + (cx - ox);
            const size_t slot = (size_t)nm * pl.slots + pl.sbase[lv] + (size_t)(py * pl.PX[lv] + px) * nb2 + k;
            reinterpret_cast<uint4 *>(masks + slot * kSlotWords)[piece] =
                reinterpret_cast<const uint4 *>(tab + tl.base[lv] + row * kSlotWords)[piece];
        }
    }
}

constexpr int kBinThreads = 1024;              // 128 (query, head) rows of 8 lanes per pass over a cell's 340 queries
__global__ __launch_bounds__(kBinThreads) void bin2_kernel(PatchPlan pl, const int64_t *__restrict__ starts,
                                                           const float *__restrict__ loc, const float *__restrict__ aw,
                                                           int M, int Lq, uint32_t *__restrict__ masks,
                                                           float *__restrict__ recs, int *__restrict__ ctl)
{
    MSDA_DYNAMIC_LDS(uint32_t, tab);
    __shared__ int rng[kL][4];                // per level: first / last patch row, first / last patch column in reach
    const int cells = pl.CY * pl.CX;
    const int m = blockIdx.x % M;
    const int c = (blockIdx.x / M) % cells;
    const int n = blockIdx.x / (M * cells);
    bin_cell<kBinThreads, false>(pl, starts, loc, aw, n, m, c, M, Lq, tab, rng, nullptr, recs, ctl);
    const TableLayout tl = table_layout(rng);
    write_masks<kBinThreads>(pl, tl, tab, n, m, c, M, masks);
}

// ------------------------------------------------------------------------------------------------------------------
// patch_dest_kernel
// ------------------------------------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK, int BANK_MASK> __device__ __forceinline__ int dpp_add(int v)
{
    return v + __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, BANK_MASK, false);
}
__device__ __forceinline__ int wave_inclusive_scan(int v)       // DPP row shifts / broadcasts, no LDS
{
    v = dpp_add<0x111, 0xf, 0xf>(v);
    v = dpp_add<0x112, 0xf, 0xf>(v);
    v = dpp_add<0x114, 0xf, 0xf>(v);
    v = dpp_add<0x118, 0xf, 0xf>(v);
    v = dpp_add<0x142, 0xa, 0xf>(v);
    v = dpp_add<0x143, 0xc, 0xf>(v);
    return v;
}

#ifndef MSDA_EMU
__device__ __forceinline__ s16x4 lds_tr_read(unsigned addr)
{
    s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
#endif

// two floats -> packed bfloat16 pair (round to nearest even: v_cvt_pk_bf16_f32)
__device__ __forceinline__ uint32_t cvt_pk_bf16(float a, float b)
{
    const f32x2 f = {a, b};
    const bf16x2 h = __builtin_convertvector(f, bf16x2);
    return __builtin_bit_cast(uint32_t, h);
}

// hi / lo bfloat16 split of two weights: hi = rne(e), lo = rne(e - hi)
__device__ __forceinline__ void split_pair(float e0, float e1, uint32_t &hi, uint32_t &lo)
{
    hi = cvt_pk_bf16(e0, e1);
    lo = cvt_pk_bf16(e0 - bf16_lo(hi), e1 - bf16_hi(hi));
}

template <typename OT> __device__ __forceinline__ void store4(OT *p, const f32x4 &v);
template <> __device__ __forceinline__ void store4<float>(float *p, const f32x4 &v)
{
    *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t *p, const f32x4 &v)
{
    *reinterpret_cast<uint2 *>(p) = make_uint2(cvt_pk_bf16(v[0], v[1]), cvt_pk_bf16(v[2], v[3]));
}

// The product kernel.  Its device code is the one that passed the GPU suite (tools/isa_audit.py compares it with that build
// instruction for instruction); experiments live in patch_dest_multi_kernel below, which only the ablation build compiles.
template <typename OT, int WPS>
__global__ __launch_bounds__(kThreads, WPS) void patch_dest_kernel(
    PatchPlan pl, const int64_t *__restrict__ starts, const float *__restrict__ recs,
    const bf16_t *__restrict__ grad_out, const uint32_t *__restrict__ masks, const int *__restrict__ ctl,
    OT *__restrict__ g_value, int N, int S, int M, int Lq, int dbg)
{
    MSDA_DYNAMIC_LDS(unsigned char, lds);
    if (ctl[kFarWord] != 0) return;                       // a far sample: the sorting pass of msda_dest.hip takes the call
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    unsigned char *wl = lds + wave * kWaveLds;
    uint16_t *list = reinterpret_cast<uint16_t *>(wl + kOffList);
    const unsigned lds0 = MSDA_LDS_BYTE_ADDR(wl);

    // ---- item: (image, head) group by XCD (hardware block b runs on XCD b % 8), coarsest level first -----------------
    const int NM = N * M;
    int nm, it;
    if ((NM & 7) == 0) {
        const int per = NM >> 3, idx = blockIdx.x >> 3;
        nm = (blockIdx.x & 7) * per + idx % per;
        it = idx / per;
    } else {
        nm = blockIdx.x % NM;
        it = blockIdx.x / NM;
    }
    int l = 0;
#pragma unroll
    for (int k = 0; k < kL; ++k) l = (it >= pl.ibase[k] && it < pl.ibase[k] + pl.nitems[k]) ? k : l;
    const int parts = pl.parts[l];
    const int pi = (it - pl.ibase[l]) * (kWaves / parts) + wave / parts, part = wave % parts;
    const int PX = pl.PX[l], npatch = pl.PY[l] * PX;
    const bool active = pi < npatch;
    const int n = nm / M, m = nm % M;
    const int H = pl.H[l], W = pl.W[l];
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};     // channels 0-15 | 16-31 of pixel lane & 15
    const int py = active ? pi / PX : 0, px = active ? pi % PX : 0;

    if (active) {
        const int nbx = pl.nbx[l], nb2 = pl.nby[l] * nbx, invx = pl.invx[l];
        const int oy = nb_origin(l, py, pl.rad[l], pl.nby[l], pl.CY), ox = nb_origin(l, px, pl.rad[l], nbx, pl.CX);
        const uint32_t *mrow = masks + ((size_t)nm * pl.slots + pl.sbase[l] + (size_t)pi * nb2) * kSlotWords;
        const float *rbase = recs + ((size_t)nm * kL + l) * (size_t)(pl.CY * pl.CX) * kCellQ * 12;
        const float Hf = (float)H, Wf = (float)W;
        const float y0f = (float)(py * 4), x0f = (float)(px * 4);
        const bool border = py * 4 + 3 >= H || px * 4 + 3 >= W;
        // per-level (start, width) of the query levels, selected per lane below
        const int st0 = (int)starts[0], st1 = (int)starts[1], st2 = (int)starts[2], st3 = (int)starts[3];
        const int W0 = pl.W[0], W1 = pl.W[1], W2 = pl.W[2], W3 = pl.W[3];
        const int kk = lane & 31, half = lane >> 5;
        const int nq = n * Lq;
        // transpose-read addresses: lane (p = lane & 15, kg = lane >> 4) supplies row 8 kg + 4 j + (p >> 2), piece p & 3
        const int p16 = lane & 15, kg = lane >> 4;
        const unsigned a_rd = lds0 + kOffA + (8 * kg + (p16 >> 2)) * 32 + (p16 & 3) * 8;     // + matrix * 1024 + j * 128
        const unsigned g_rd = lds0 + kOffG + (8 * kg + (p16 >> 2)) * 64 + (p16 & 3) * 8;     // + tile * 32 + j * 256

        // This wave's mask words: slots part, part + parts, ... of the patch's neighbourhood, 12 words each, taken 64 at
        // a time (lane i holds word wbase + i; the next 64 are already travelling).  Word order = (slot, word) order, so
        // a wave-wide prefix sum of the popcounts puts the candidates in a fixed order.
        const int nslots = (nb2 - part + parts - 1) / parts, nwords = nslots * kSlotWords;
        auto word_of = [&](int w) -> uint32_t {              // (captures scalars only; inlined)
            if (w >= nwords) return 0u;
            const int s = (w * 683) >> 13, wi = w - s * kSlotWords;      // w / 12 for w < 2048
            return mrow[(part + s * parts) * kSlotWords + wi];
        };
        int wbase = 0;
        uint32_t cur = word_of(lane), nxt = word_of(64 + lane);
        int head = 0, tail = 0;
        bool exhausted = nwords <= 0;
        int avail_c = 0;                                   // the step whose operands are in flight / in registers
        float4 xy_c = make_float4(0.f, 0.f, 0.f, 0.f);
        float2 a2_c = make_float2(0.f, 0.f);
        uint4 g0_c = make_uint4(0u, 0u, 0u, 0u), g1_c = g0_c;
        for (;;) {
            // ---- refill: expand mask bits into the candidate list until a full step is there ---------------------------
            while (tail - head < kStep && !exhausted) {
                if (__builtin_amdgcn_ballot_w64(cur != 0u) == 0ull) {
                    wbase += 64;
                    if (wbase >= nwords) { exhausted = true; break; }
                    cur = nxt;
                    nxt = word_of(wbase + 64 + lane);
                    continue;
                }
                if (head > 0) {                           // carry the < 32 left-over entries to the front
                    const int left = tail - head;
                    int v = 0;
                    if (lane < left) v = list[head + lane];
                    MSDA_WAVE_LDS_SYNC();
                    if (lane < left) list[lane] = (uint16_t)v;
                    MSDA_WAVE_LDS_SYNC();
                    head = 0; tail = left;
                }
                const int cnt = __popc(cur);
                const int incl = wave_inclusive_scan(cnt);
                const bool fits = incl <= kListCap - tail;             // a prefix of the lanes
                const unsigned long long fm = __builtin_amdgcn_ballot_w64(fits);
                const int nfit = __popcll(fm);                          // >= 1: 32 bits of one lane always fit
                const int total = __builtin_amdgcn_readlane(incl, nfit - 1);
                if (fits) {
                    const int w = wbase + lane;
                    const int s = (w * 683) >> 13, wi = w - s * kSlotWords;
                    const int code0 = ((part + s * parts) << 9) | (wi * 32);
                    int pos = tail + incl - cnt;
                    uint32_t f = cur;
                    while (f) {
                        const int b = __ffs(f) - 1;
                        list[pos++] = (uint16_t)(code0 | b);
                        f &= f - 1u;
                    }
                    cur = 0u;
                }
                MSDA_WAVE_LDS_SYNC();
                tail += total;
            }
            // ---- operands of the NEXT step start travelling (software pipeline: one step of loads in flight) ------------
            const int avail_n = min(tail - head, kStep);
            float4 xy_n = make_float4(0.f, 0.f, 0.f, 0.f);
            float2 a2_n = make_float2(0.f, 0.f);
            uint4 g0_n = make_uint4(0u, 0u, 0u, 0u), g1_n = g0_n;
            if (avail_n > 0) {
                const int h0 = head;
                head += avail_n;
                const int code = list[h0 + (kk < avail_n ? kk : 0)];
                const int slot = code >> 9, bit = code & 511;
                const int sy = (slot * invx) >> 16, sx = slot - sy * nbx;
                const int lq = bit < 256 ? 0 : bit < 320 ? 1 : bit < 336 ? 2 : 3;
                const int r = bit - (lq == 0 ? 0 : lq == 1 ? 256 : lq == 2 ? 320 : 336);
                const int sh = 4 - lq;
                const int iy = ((oy + sy) << sh) + (r >> sh), ix = ((ox + sx) << sh) + (r & ((1 << sh) - 1));
                const int stq = lq == 0 ? st0 : lq == 1 ? st1 : lq == 2 ? st2 : st3;
                const int Wq = lq == 0 ? W0 : lq == 1 ? W1 : lq == 2 ? W2 : W3;
                int qm = (nq + stq + iy * Wq + ix) * M + m;                            // < 2^25 (checked by the ABI)
                int ri = ((oy + sy) * pl.CX + ox + sx) * kCellQ + bit;                 // record of (cell, query)
                if (MSDA_DBG(dbg) & 2) { qm = (nq + (lane & 7)) * M + m; ri = lane & 7; }   // ablation: cache-resident operands
                const float *rec = rbase + (size_t)ri * 12;
                xy_n = reinterpret_cast<const float4 *>(rec)[half];
                a2_n = reinterpret_cast<const float2 *>(rec + 8)[half];
                const uint4 *gp = reinterpret_cast<const uint4 *>(grad_out + (size_t)qm * kD + half * 16);
                g0_n = gp[0]; g1_n = gp[1];
            }
            // ---- one MFMA step over the `avail` <= 32 candidates whose operands were requested one iteration ago -------
            const int avail = avail_c;
            const float4 xy = xy_c;
            const float2 a2 = a2_c;
            const uint4 g0 = g0_c, g1 = g1_c;
            avail_c = avail_n; xy_c = xy_n; a2_c = a2_n; g0_c = g0_n; g1_c = g1_n;
            if (avail <= 0) {
                if (avail_n <= 0) break;
                continue;
            }
            if (MSDA_DBG(dbg) & 1) { acc0[0] += (float)avail; continue; }            // ablation: enumeration only
            if (MSDA_DBG(dbg) & 4) { acc0[0] += xy.x + a2.x + __uint_as_float(g0.x ^ g1.y); continue; }   // ablation: loads only
            const bool valid = kk < avail;
            float e[16];
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
                const float x = pt ? xy.z : xy.x, y = pt ? xy.w : xy.y, a_in = pt ? a2.y : a2.x;
                const float h_im = fmaf(y, Hf, -0.5f), w_im = fmaf(x, Wf, -0.5f);
                const bool inside = (h_im > -1.f) && (w_im > -1.f) && (h_im < Hf) && (w_im < Wf);            // .cuh:285
                const bool use = inside && valid;
                const float a = use ? a_in : 0.f;
                const float dx = use ? w_im - x0f : -8.f, dy = use ? h_im - y0f : -8.f;
                float tx[4], ty[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    tx[c] = fmaxf(0.f, 1.f - fabsf(dx - (float)c));
                    ty[c] = a * fmaxf(0.f, 1.f - fabsf(dy - (float)c));
                }
                if (border) {                              // pixels of the patch beyond the level's edge receive nothing
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        tx[c] = px * 4 + c < W ? tx[c] : 0.f;
                        ty[c] = py * 4 + c < H ? ty[c] : 0.f;
                    }
                }
#pragma unroll
                for (int ry = 0; ry < 4; ++ry)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        e[ry * 4 + c] = pt ? fmaf(ty[ry], tx[c], e[ry * 4 + c]) : ty[ry] * tx[c];
            }
            uint32_t hi[8], lo[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) split_pair(e[2 * i], e[2 * i + 1], hi[i], lo[i]);
            // A^T matrices: [half][hi | lo][32 groups][16 pixels] bfloat16, row kk of the lane's half
            uint4 *arow = reinterpret_cast<uint4 *>(wl + kOffA + half * 2048 + kk * 32);
            arow[0] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
            arow[1] = make_uint4(hi[4], hi[5], hi[6], hi[7]);
            arow[64] = make_uint4(lo[0], lo[1], lo[2], lo[3]);            // + 1024 bytes
            arow[65] = make_uint4(lo[4], lo[5], lo[6], lo[7]);
            uint4 *grow = reinterpret_cast<uint4 *>(wl + kOffG + kk * 64 + half * 32);
            grow[0] = g0;
            grow[1] = g1;
            MSDA_WAVE_LDS_SYNC();
            // operands: G^T tiles (channels 0-15 | 16-31) and the four A^T matrices, 8 consecutive groups per lane
            union Frag { bf16x8 v; s16x4 h[2]; };
            Frag gt0, gt1, at0, at1, at2, at3;
            gt0.h[0] = lds_tr_read(g_rd); gt0.h[1] = lds_tr_read(g_rd + 256);
            gt1.h[0] = lds_tr_read(g_rd + 32); gt1.h[1] = lds_tr_read(g_rd + 32 + 256);
            at0.h[0] = lds_tr_read(a_rd); at0.h[1] = lds_tr_read(a_rd + 128);
            at1.h[0] = lds_tr_read(a_rd + 1024); at1.h[1] = lds_tr_read(a_rd + 1024 + 128);
            at2.h[0] = lds_tr_read(a_rd + 2048); at2.h[1] = lds_tr_read(a_rd + 2048 + 128);
            at3.h[0] = lds_tr_read(a_rd + 3072); at3.h[1] = lds_tr_read(a_rd + 3072 + 128);
            MSDA_WAVE_LDS_SYNC();
            // (the wait names the fragments so that the scheduler cannot lift an MFMA above it: the compiler does not
            //  know that the transpose-reads' results are still in flight)
#ifndef MSDA_EMU
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(gt0.h[0]), "+v"(gt0.h[1]), "+v"(gt1.h[0]), "+v"(gt1.h[1]), "+v"(at0.h[0]), "+v"(at0.h[1]),
                           "+v"(at1.h[0]), "+v"(at1.h[1]), "+v"(at2.h[0]), "+v"(at2.h[1]), "+v"(at3.h[0]), "+v"(at3.h[1])
                         :
                         : "memory");
#endif
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gt0.v, at0.v, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gt1.v, at0.v, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gt0.v, at1.v, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gt1.v, at1.v, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gt0.v, at2.v, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gt1.v, at2.v, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gt0.v, at3.v, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gt1.v, at3.v, acc1, 0, 0, 0);
        }
    }

    // ---- parts of one patch: summed in part order by the first wave of the patch, then the rows leave once ----------
    if (parts > 1) {
        float *red = reinterpret_cast<float *>(wl);           // this wave's own (now idle) LDS
        if (part > 0) {
            reinterpret_cast<f32x4 *>(red)[lane * 2] = acc0;
            reinterpret_cast<f32x4 *>(red)[lane * 2 + 1] = acc1;
        }
        __syncthreads();
        if (part == 0) {
            for (int j = 1; j < parts; ++j) {
                const f32x4 *o = reinterpret_cast<const f32x4 *>(lds + (wave + j) * kWaveLds);
                acc0 += o[lane * 2];
                acc1 += o[lane * 2 + 1];
            }
        }
    }
    if (active && part == 0) {
        const int pix = lane & 15, y = py * 4 + (pix >> 2), x = px * 4 + (pix & 3);
        if (y < H && x < W) {
            const long row = (long)n * S + (long)starts[l] + (long)y * W + x;
            OT *dst = g_value + (row * M + m) * kD + 4 * (lane >> 4);
            store4<OT>(dst, acc0);
            store4<OT>(dst + 16, acc1);
        }
    }
}

#ifdef MSDA_ABLATION
// Experiment arms of the patch pass, never in the product library (tools/r03_experiments.py): several patches per wave on the
// fine levels with the next patch's mask words prefetched, the mask-word prefetch outside a branch, tents through the clamp
// modifier, operand images staged with exchanged halves (bank-conflict-free transposing reads).  Must reproduce
// patch_dest_kernel bit for bit.
// CELLG (round 6, RLIPV2_PATCH_CELLG=1): grad_out rows are read from a copy in the group records' CELL-MAJOR order
// (`gcell`: [(image, head)][cell][query of the cell][32 channels], written by grad_out_cells_kernel below) -- the candidate's
// record index addresses its grad_out row too: no decode of the query's pyramid position per candidate (the level select chain,
// ~18 of the step's ~244 VALU instructions), and the rows of a cell's neighbouring queries share 128-byte lines (in grad_out a
// head's 64-byte row shares its line with another head's, i.e. with a workgroup on another XCD).
template <typename OT, int WPS, bool CELLG>
__global__ __launch_bounds__(kThreads, WPS) void patch_dest_multi_kernel(
    PatchPlan pl, const int64_t *__restrict__ starts, const float *__restrict__ recs,
    const bf16_t *__restrict__ grad_out, const uint32_t *__restrict__ masks, const int *__restrict__ ctl,
    OT *__restrict__ g_value, int N, int S, int M, int Lq, int dbg, const bf16_t *__restrict__ gcell)
{
    constexpr bool MULTI = true;
    MSDA_DYNAMIC_LDS(unsigned char, lds);
    if (ctl[kFarWord] != 0) return;                       // a far sample: the sorting pass of msda_dest.hip takes the call
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    unsigned char *wl = lds + wave * kWaveLds;
    uint16_t *list = reinterpret_cast<uint16_t *>(wl + kOffList);
    const unsigned lds0 = MSDA_LDS_BYTE_ADDR(wl);

    // ---- item: (image, head) group by XCD (hardware block b runs on XCD b % 8), coarsest level first -----------------
    const int NM = N * M;
    int nm, it;
    if ((NM & 7) == 0) {
        const int per = NM >> 3, idx = blockIdx.x >> 3;
        nm = (blockIdx.x & 7) * per + idx % per;
        it = idx / per;
    } else {
        nm = blockIdx.x % NM;
        it = blockIdx.x / NM;
    }
    int l = 0;
#pragma unroll
    for (int k = 0; k < kL; ++k) l = (it >= pl.ibase[k] && it < pl.ibase[k] + pl.nitems[k]) ? k : l;
    const int parts = pl.parts[l], reps = MULTI ? pl.reps[l] : 1;   // waves per patch | patches per wave (> 1 only with parts == 1)
    const int part = wave % parts;
    const int pi0 = ((it - pl.ibase[l]) * (kWaves / parts) + wave / parts) * reps;
    const int PX = pl.PX[l], npatch = pl.PY[l] * PX;
    const int n = nm / M, m = nm % M;
    const int H = pl.H[l], W = pl.W[l];
    const int nbx = pl.nbx[l], nb2 = pl.nby[l] * nbx, invx = pl.invx[l];
    const float *rbase = recs + ((size_t)nm * kL + l) * (size_t)(pl.CY * pl.CX) * kCellQ * 12;
    const bf16_t *gbase = CELLG ? gcell + (size_t)nm * (size_t)(pl.CY * pl.CX) * kCellQ * kD : nullptr;
    (void)gbase;
    const float Hf = (float)H, Wf = (float)W;
    // per-level (start, width) of the query levels, selected per lane below
    const int st0 = (int)starts[0], st1 = (int)starts[1], st2 = (int)starts[2], st3 = (int)starts[3];
    const int W0 = pl.W[0], W1 = pl.W[1], W2 = pl.W[2], W3 = pl.W[3];
    const int kk = lane & 31, half = lane >> 5;
    const int nq = n * Lq;
    // transpose-read addresses: lane (p = lane & 15, kg = lane >> 4) supplies row 8 kg + 4 j + (p >> 2), piece p & 3
    const int p16 = lane & 15, kg = lane >> 4;
    // (experimental instantiation: the odd 8-row blocks are staged with their two 4-row halves (A^T) / their two 32-byte
    //  channel halves (grad_out) exchanged, so that the two lane groups of a 32-lane half -- rows 8 kg .. and 8 (kg + 1) ..,
    //  256 / 512 bytes apart -- read from different banks; addresses only, the instruction stream is the same)
    const int rsw = MULTI ? (kg & 1) : 0;
    const int a_j = rsw ? -128 : 128, g_t = rsw ? -32 : 32;
    const unsigned a_rd = lds0 + kOffA + (8 * kg + (p16 >> 2) + 4 * rsw) * 32 + (p16 & 3) * 8;     // + matrix * 1024 + j * a_j
    const unsigned g_rd = lds0 + kOffG + (8 * kg + (p16 >> 2)) * 64 + (p16 & 3) * 8 + rsw * 32;    // + tile * g_t + j * 256
    const int wsw = MULTI ? ((kk >> 3) & 1) : 0;
    // This wave's mask words of a patch: slots part, part + parts, ... of the patch's neighbourhood, 12 words each, taken 64
    // at a time (lane i holds word wbase + i; the next 64 are already travelling).  Word order = (slot, word) order, so a
    // wave-wide prefix sum of the popcounts puts the candidates in a fixed order.
    const int nslots = (nb2 - part + parts - 1) / parts, nwords = nslots * kSlotWords;
    const uint32_t *mrow0 = masks + ((size_t)nm * pl.slots + pl.sbase[l]) * kSlotWords;      // + patch * nb2 * 12
    auto word_of = [part, parts, nwords](const uint32_t *mrow, int w) -> uint32_t {         // (value captures; inlined)
        if (w >= nwords) return 0u;
        const int s = (w * 683) >> 13, wi = w - s * kSlotWords;                              // w / 12 for w < 2048
        return mrow[(part + s * parts) * kSlotWords + wi];
    };
    // A wave that owns several patches in turn (MULTI; fine levels: a few MFMA steps per patch) already holds the first
    // 128 mask words of its NEXT patch: without them every patch starts with a load latency nothing else of the wave covers.
    uint32_t pcur = 0u, pnxt = 0u;
    if (pi0 < npatch) {
        const uint32_t *mr = mrow0 + (size_t)pi0 * nb2 * kSlotWords;
        pcur = word_of(mr, lane); pnxt = word_of(mr, 64 + lane);
    }
  for (int rep = 0; rep < reps; ++rep) {
    const int pi = pi0 + rep;
    const bool active = pi < npatch;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};     // channels 0-15 | 16-31 of pixel lane & 15
    const int py = active ? pi / PX : 0, px = active ? pi % PX : 0;

    if (active) {
        const int oy = nb_origin(l, py, pl.rad[l], pl.nby[l], pl.CY), ox = nb_origin(l, px, pl.rad[l], nbx, pl.CX);
        const uint32_t *mrow = mrow0 + (size_t)pi * nb2 * kSlotWords;
        const float y0f = (float)(py * 4), x0f = (float)(px * 4);
        const bool border = py * 4 + 3 >= H || px * 4 + 3 >= W;
        int wbase = 0;
        uint32_t cur = pcur, nxt = pnxt;
        if (rep + 1 < reps && pi + 1 < npatch) {
            const uint32_t *mr = mrow + (size_t)nb2 * kSlotWords;
            pcur = word_of(mr, lane); pnxt = word_of(mr, 64 + lane);
        }
        int head = 0, tail = 0;
        bool exhausted = nwords <= 0;
        int avail_c = 0;                                   // the step whose operands are in flight / in registers
        float4 xy_c = make_float4(0.f, 0.f, 0.f, 0.f);
        float2 a2_c = make_float2(0.f, 0.f);
        uint4 g0_c = make_uint4(0u, 0u, 0u, 0u), g1_c = g0_c;
        for (;;) {
            // ---- refill: expand mask bits into the candidate list until a full step is there ---------------------------
            while (tail - head < kStep && !exhausted) {
                if (__builtin_amdgcn_ballot_w64(cur != 0u) == 0ull) {
                    wbase += 64;
                    if (wbase >= nwords) { exhausted = true; break; }
                    if (MULTI) {
                        // (experiment) the load NOT in a branch: the plain form below merges `0` and the loaded word in a
                        // phi, and the compiler waits for the load right where it is issued (s_waitcnt vmcnt(0) + v_mov in
                        // the device assembly) -- one exposed memory latency per 64 mask words instead of a prefetch.
                        // Clamped address, the words beyond the end masked when they become `cur`.
                        cur = (wbase + lane < nwords) ? nxt : 0u;
                        const int wn = min(wbase + 64 + lane, nwords - 1);
                        const int sn = (wn * 683) >> 13;
                        nxt = mrow[(part + sn * parts) * kSlotWords + (wn - sn * kSlotWords)];
                    } else {
                        cur = nxt;
                        nxt = word_of(mrow, wbase + 64 + lane);
                    }
                    continue;
                }
                if (head > 0) {                           // carry the < 32 left-over entries to the front
                    const int left = tail - head;
                    int v = 0;
                    if (lane < left) v = list[head + lane];
                    MSDA_WAVE_LDS_SYNC();
                    if (lane < left) list[lane] = (uint16_t)v;
                    MSDA_WAVE_LDS_SYNC();
                    head = 0; tail = left;
                }
                const int cnt = __popc(cur);
                const int incl = wave_inclusive_scan(cnt);
                const bool fits = incl <= kListCap - tail;             // a prefix of the lanes
                const unsigned long long fm = __builtin_amdgcn_ballot_w64(fits);
                const int nfit = __popcll(fm);                          // >= 1: 32 bits of one lane always fit
                const int total = __builtin_amdgcn_readlane(incl, nfit - 1);
                if (fits) {
                    const int w = wbase + lane;
                    const int s = (w * 683) >> 13, wi = w - s * kSlotWords;
                    const int code0 = ((part + s * parts) << 9) | (wi * 32);
                    int pos = tail + incl - cnt;
                    uint32_t f = cur;
                    while (f) {
                        const int b = __ffs(f) - 1;
                        list[pos++] = (uint16_t)(code0 | b);
                        f &= f - 1u;
                    }
                    cur = 0u;
                }
                MSDA_WAVE_LDS_SYNC();
                tail += total;
            }
            // ---- operands of the NEXT step start travelling (software pipeline: one step of loads in flight) ------------
            const int avail_n = min(tail - head, kStep);
            float4 xy_n = make_float4(0.f, 0.f, 0.f, 0.f);
            float2 a2_n = make_float2(0.f, 0.f);
            uint4 g0_n = make_uint4(0u, 0u, 0u, 0u), g1_n = g0_n;
            if (avail_n > 0) {
                const int h0 = head;
                head += avail_n;
                const int code = list[h0 + (kk < avail_n ? kk : 0)];
                const int slot = code >> 9, bit = code & 511;
                const int sy = (slot * invx) >> 16, sx = slot - sy * nbx;
                int ri = ((oy + sy) * pl.CX + ox + sx) * kCellQ + bit;                 // record of (cell, query)
                if (MSDA_DBG(dbg) & 2) ri = lane & 7;                                   // ablation: cache-resident operands
                const uint4 *gp;
                if (CELLG) {                                                            // the record index is the row's index too
                    gp = reinterpret_cast<const uint4 *>(gbase + (size_t)ri * kD + half * 16);
                } else {
                    const int lq = bit < 256 ? 0 : bit < 320 ? 1 : bit < 336 ? 2 : 3;
                    const int r = bit - (lq == 0 ? 0 : lq == 1 ? 256 : lq == 2 ? 320 : 336);
                    const int sh = 4 - lq;
                    const int iy = ((oy + sy) << sh) + (r >> sh), ix = ((ox + sx) << sh) + (r & ((1 << sh) - 1));
                    const int stq = lq == 0 ? st0 : lq == 1 ? st1 : lq == 2 ? st2 : st3;
                    const int Wq = lq == 0 ? W0 : lq == 1 ? W1 : lq == 2 ? W2 : W3;
                    int qm = (nq + stq + iy * Wq + ix) * M + m;                        // < 2^25 (checked by the ABI)
                    if (MSDA_DBG(dbg) & 2) qm = (nq + (lane & 7)) * M + m;
                    gp = reinterpret_cast<const uint4 *>(grad_out + (size_t)qm * kD + half * 16);
                }
                const float *rec = rbase + (size_t)ri * 12;
                xy_n = reinterpret_cast<const float4 *>(rec)[half];
                a2_n = reinterpret_cast<const float2 *>(rec + 8)[half];
                g0_n = gp[0]; g1_n = gp[1];
            }
            // ---- one MFMA step over the `avail` <= 32 candidates whose operands were requested one iteration ago -------
            const int avail = avail_c;
            const float4 xy = xy_c;
            const float2 a2 = a2_c;
            const uint4 g0 = g0_c, g1 = g1_c;
            avail_c = avail_n; xy_c = xy_n; a2_c = a2_n; g0_c = g0_n; g1_c = g1_n;
            if (avail <= 0) {
                if (avail_n <= 0) break;
                continue;
            }
            if (MSDA_DBG(dbg) & 1) { acc0[0] += (float)avail; continue; }            // ablation: enumeration only
            if (MSDA_DBG(dbg) & 4) { acc0[0] += xy.x + a2.x + __uint_as_float(g0.x ^ g1.y); continue; }   // ablation: loads only
            const bool valid = kk < avail;
            float e[16];
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
                const float x = pt ? xy.z : xy.x, y = pt ? xy.w : xy.y, a_in = pt ? a2.y : a2.x;
                const float h_im = fmaf(y, Hf, -0.5f), w_im = fmaf(x, Wf, -0.5f);
                const bool inside = (h_im > -1.f) && (w_im > -1.f) && (h_im < Hf) && (w_im < Wf);            // .cuh:285
                const bool use = inside && valid;
                const float a = use ? a_in : 0.f;
                const float dx = use ? w_im - x0f : -8.f, dy = use ? h_im - y0f : -8.f;
                float tx[4], ty[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    // (experimental instantiation: the median with 0 and 1 compiles to the subtraction's `clamp` modifier, one
                    //  instruction less per tent, same bits -- 1 - |d| never exceeds 1, NaN -> 0 either way)
                    const float ux = 1.f - fabsf(dx - (float)c), uy = 1.f - fabsf(dy - (float)c);
                    tx[c] = MULTI ? __builtin_amdgcn_fmed3f(ux, 0.f, 1.f) : fmaxf(0.f, ux);
                    ty[c] = a * (MULTI ? __builtin_amdgcn_fmed3f(uy, 0.f, 1.f) : fmaxf(0.f, uy));
                }
                if (border) {                              // pixels of the patch beyond the level's edge receive nothing
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        tx[c] = px * 4 + c < W ? tx[c] : 0.f;
                        ty[c] = py * 4 + c < H ? ty[c] : 0.f;
                    }
                }
#pragma unroll
                for (int ry = 0; ry < 4; ++ry)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        e[ry * 4 + c] = pt ? fmaf(ty[ry], tx[c], e[ry * 4 + c]) : ty[ry] * tx[c];
            }
            uint32_t hi[8], lo[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) split_pair(e[2 * i], e[2 * i + 1], hi[i], lo[i]);
            // A^T matrices: [half][hi | lo][32 groups][16 pixels] bfloat16, row kk of the lane's half
            uint4 *arow = reinterpret_cast<uint4 *>(wl + kOffA + half * 2048 + (kk ^ (wsw << 2)) * 32);
            arow[0] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
            arow[1] = make_uint4(hi[4], hi[5], hi[6], hi[7]);
            arow[64] = make_uint4(lo[0], lo[1], lo[2], lo[3]);            // + 1024 bytes
            arow[65] = make_uint4(lo[4], lo[5], lo[6], lo[7]);
            uint4 *grow = reinterpret_cast<uint4 *>(wl + kOffG + kk * 64 + (half ^ wsw) * 32);
            grow[0] = g0;
            grow[1] = g1;
            MSDA_WAVE_LDS_SYNC();
            // operands: G^T tiles (channels 0-15 | 16-31) and the four A^T matrices, 8 consecutive groups per lane
            union Frag { bf16x8 v; s16x4 h[2]; };
            Frag gt0, gt1, at0, at1, at2, at3;
            gt0.h[0] = lds_tr_read(g_rd); gt0.h[1] = lds_tr_read(g_rd + 256);
            gt1.h[0] = lds_tr_read(g_rd + g_t); gt1.h[1] = lds_tr_read(g_rd + g_t + 256);
            at0.h[0] = lds_tr_read(a_rd); at0.h[1] = lds_tr_read(a_rd + a_j);
            at1.h[0] = lds_tr_read(a_rd + 1024); at1.h[1] = lds_tr_read(a_rd + 1024 + a_j);
            at2.h[0] = lds_tr_read(a_rd + 2048); at2.h[1] = lds_tr_read(a_rd + 2048 + a_j);
            at3.h[0] = lds_tr_read(a_rd + 3072); at3.h[1] = lds_tr_read(a_rd + 3072 + a_j);
            MSDA_WAVE_LDS_SYNC();
            // (the wait names the fragments so that the scheduler cannot lift an MFMA above it: the compiler does not
            //  know that the transpose-reads' results are still in flight)
#ifndef MSDA_EMU
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(gt0.h[0]), "+v"(gt0.h[1]), "+v"(gt1.h[0]), "+v"(gt1.h[1]), "+v"(at0.h[0]), "+v"(at0.h[1]),
                           "+v"(at1.h[0]), "+v"(at1.h[1]), "+v"(at2.h[0]), "+v"(at2.h[1]), "+v"(at3.h[0]), "+v"(at3.h[1])
                         :
                         : "memory");
#endif
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gt0.v, at0.v, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gt1.v, at0.v, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gt0.v, at1.v, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gt1.v, at1.v, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gt0.v, at2.v, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gt1.v, at2.v, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gt0.v, at3.v, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gt1.v, at3.v, acc1, 0, 0, 0);
        }
    }

    // ---- parts of one patch: summed in part order by the first wave of the patch, then the rows leave once ----------
    if (parts > 1) {
        float *red = reinterpret_cast<float *>(wl);           // this wave's own (now idle) LDS
        if (part > 0) {
            reinterpret_cast<f32x4 *>(red)[lane * 2] = acc0;
            reinterpret_cast<f32x4 *>(red)[lane * 2 + 1] = acc1;
        }
        __syncthreads();
        if (part == 0) {
            for (int j = 1; j < parts; ++j) {
                const f32x4 *o = reinterpret_cast<const f32x4 *>(lds + (wave + j) * kWaveLds);
                acc0 += o[lane * 2];
                acc1 += o[lane * 2 + 1];
            }
        }
    }
    if (active && part == 0) {
        const int pix = lane & 15, y = py * 4 + (pix >> 2), x = px * 4 + (pix & 3);
        if (y < H && x < W) {
            const long row = (long)n * S + (long)starts[l] + (long)y * W + x;
            OT *dst = g_value + (row * M + m) * kD + 4 * (lane >> 4);
            store4<OT>(dst, acc0);
            store4<OT>(dst + 16, acc1);
        }
    }
  }   // patches of this wave
}
#endif   // MSDA_ABLATION

// ------------------------------------------------------------------------------------------------------------------
// cell_backward_kernel: grad_sampling_loc / grad_attn_weight (or, fused, the gradient of the projection row) of one
// cell's queries with the value windows in LDS and the channel dot products on v_mfma_f32_4x4x4_16B_bf16 -- plus the
// binning of bin2_kernel, whose sample geometry it shares.  (Reference semantics: ms_deform_im2col_cuda.cuh:301-403,
// ms_deform_attn_col2im_bilinear :87-159 for the two gradients.)
//
// Round 2's K1 (quad_backward_shared_kernel) gathers every corner row from global memory -- 2.9 GB of 64-byte
// gathers per batch-4 encoder call at ~64 B/clk/CU of texture path -- and spends 16 VALU instructions per corner on
// the 8-channel piece of a dot product (8 bf16 unpacks + 8 FMAs) plus a DPP reduction.  Here:
//   * a workgroup owns one (image, head, cell): the <= 340 queries of a pyramid column sample a compact window of
//     every level; phase 1 (bin_cell) finds the windows' bounding boxes while it builds the patch masks, phase 2 copies
//     the windows into LDS once (48 KB budget; a level that does not fit is gathered directly, so any input works);
//   * a DPP quad owns one query and lane c of the quad owns bilinear CORNER c of the current sample: it reads that
//     corner's whole 64-byte row from LDS (4 x ds_read_b128, piece order rotated by the quad's index so that the 16
//     lanes of an LDS group hit 16 different bank quads; window pitch = 2 mod 4 keeps a quad's four rows apart);
//   * the 16 MFMA blocks of v_mfma_f32_4x4x4_16B_bf16 are exactly the 16 quads of a wave: block = quad, B column j =
//     corner j's 4 channels, A rows = the query's grad_out channels -> D[., j] = <grad_out, corner j> after 8
//     instructions per sample, exact products, float32 accumulation, no unpacking and no cross-lane reduction.
// ------------------------------------------------------------------------------------------------------------------
// What the product build runs (the other modes / instantiations are the arms of tools/experiments_r05.py; they become the
// default here, in one place, once a GPU run has ACCEPTED them and shown them faster.  Accepted = grad_value bit-identical and the
// float32-formula gradients within a rounding: "same arithmetic, bit-identical" below holds on the lane-level model; on the device
// hipcc contracts the formulas into FMAs per kernel -- in this very kernel differently for sample 3 of a level than for samples
// 0-2 -- see profiles/r05_records_route_static.txt):
constexpr int kCellMode = 0;                  // cell_backward_kernel<., MODE>
constexpr int kPatchMulti = 0;                // 1: patch_dest_multi_kernel (ablation build only)
constexpr int kPatchReps = 1;                 // patches per wave on the fine levels (MULTI only)
constexpr int kCellThreads = 512;
constexpr int kWinBytes = 48 * 1024;          // LDS window budget (all levels together)
constexpr int kZeroBytes = 128;               // zeros in front of the windows: where out-of-level corners read
constexpr int kMaxWinPx = kWinBytes / 64;
constexpr int kCellTableMax = 28 * 1024;      // bin table bytes the cell kernel can hold next to the windows
constexpr unsigned kOobOff = 0xFFFFFF00u;     // buffer offset beyond num_records: the hardware returns zeros
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma444(uint32_t a0, uint32_t a1, uint32_t b0, uint32_t b1, f32x4 c)
{
    union { uint32_t u[2]; s16x4 v; } a, b;
    a.u[0] = a0; a.u[1] = a1; b.u[0] = b0; b.u[1] = b1;
    return __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a.v, b.v, c, 0, 0, 0);
}

// one sample: this lane's corner row (4 pieces, rotated) -> <grad_out, corner> of the quad's four corners -> the three
// gradients of the sample (identical in all four lanes)
__device__ __forceinline__ void cell_sample(const unsigned char *lds, __amdgpu_buffer_rsrc_t vr, const uint4 (&g)[4], float x,
                                            float y, float w, int H, int W, int wx0, int wy0, int pitch, int wbase,
                                            unsigned lvl_byte, int row_bytes, int crn, int rot, float &g_a, float &g_x,
                                            float &g_y, int dbg = 0)
{
    const float Hf = (float)H, Wf = (float)W;
    const float h_im = fmaf(y, Hf, -0.5f), w_im = fmaf(x, Wf, -0.5f);
    const bool inside = (h_im > -1.f) && (w_im > -1.f) && (h_im < Hf) && (w_im < Wf);     // .cuh:285 (NaN -> false)
    const float hs = inside ? h_im : 0.f, ws = inside ? w_im : 0.f;
    const float hf = floorf(hs), wf = floorf(ws);
    const float lh = hs - hf, lw = ws - wf, hh = 1.f - lh, hw = 1.f - lw;
    const int iy = (int)hf + (crn >> 1), ix = (int)wf + (crn & 1);
    const bool ok = inside && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
    uint4 v[4];
    if (MSDA_DBG(dbg) & 4) {                                                   // ablation: no corner reads at all
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = make_uint4(g[t].x + ok, g[t].y, g[t].z, g[t].w);
    } else if (wbase >= 0) {                                                   // (wave-uniform)
        const int addr = ok ? wbase + (__mul24(iy - wy0, pitch) + (ix - wx0)) * 64 : 0;     // 0: the zero slot
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = *reinterpret_cast<const uint4 *>(lds + addr + ((t ^ rot) << 4));
    } else {
        const unsigned off = ok ? lvl_byte + (unsigned)__mul24(__mul24(iy, W) + ix, row_bytes) : kOobOff;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(vr, off + ((t ^ rot) << 4), 0, 0);
            v[t] = make_uint4(r.x, r.y, r.z, r.w);
        }
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#ifdef MSDA_CELL_NO_MFMA
    acc[0] = __uint_as_float(v[0].x ^ v[1].y ^ v[2].z ^ v[3].w);                // ablation build: no MFMAs
#else
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        acc = mfma444(g[t].x, g[t].y, v[t].x, v[t].y, acc);
        acc = mfma444(g[t].z, g[t].w, v[t].z, v[t].w, acc);
    }
#endif
    const float e = acc[0];
    const float e1 = quad_bcast<0>(e), e2 = quad_bcast<1>(e), e3 = quad_bcast<2>(e), e4 = quad_bcast<3>(e);
    const float wgt = inside ? w : 0.f;
    g_a = inside ? hh * (hw * e1 + lw * e2) + lh * (hw * e3 + lw * e4) : 0.f;
    g_x = Wf * wgt * (hh * (e2 - e1) + lh * (e4 - e3));
    g_y = Hf * wgt * (hw * (e3 - e1) + lw * (e4 - e2));
}

// <grad_out, corner row> of THIS lane's corner of a sample whose corner base / validity mask are given (the dot-product
// half of cell_sample)
// SWAP: the value row is the A operand and grad_out the B operand -- D[i][j] = <corner i, grad_out> in EVERY lane j, so
// the four dots arrive in acc[0..3] of each lane without the four DPP broadcasts (same products, same sums).
template <bool SWAP>
__device__ __forceinline__ f32x4 cell_corner_dot(const unsigned char *lds, __amdgpu_buffer_rsrc_t vr, const uint4 (&g)[4],
                                                 bool staged, int base, int delta, int okmask, int crn, int rot)
{
    const bool ok = (okmask >> crn) & 1;
    uint4 v[4];
    if (staged) {
        const int addr = ok ? base + delta : 0;                                  // 0: the zero slot
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = *reinterpret_cast<const uint4 *>(lds + addr + ((t ^ rot) << 4));
    } else {
        const unsigned off = ok ? (unsigned)base + (unsigned)delta : kOobOff;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(vr, off + ((t ^ rot) << 4), 0, 0);
            v[t] = make_uint4(r.x, r.y, r.z, r.w);
        }
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if (SWAP) {
            acc = mfma444(v[t].x, v[t].y, g[t].x, g[t].y, acc);
            acc = mfma444(v[t].z, v[t].w, g[t].z, g[t].w, acc);
        } else {
            acc = mfma444(g[t].x, g[t].y, v[t].x, v[t].y, acc);
            acc = mfma444(g[t].z, g[t].w, v[t].z, v[t].w, acc);
        }
    }
    return acc;
}

// One level of a task with the sample geometry computed ONCE per quad: quad lane p owns POINT p of the level (the four
// points in parallel instead of every lane repeating each sample's ~25 geometry instructions), the quad fetches a
// sample's corner base and validity mask with two DPP moves, lane p keeps the four dots of ITS sample and evaluates the
// reference's formulas once.  Same arithmetic on the same operands as cell_sample: bit-identical results.
// `la` / `lb` / `wa`: quad lane 0 holds the level (after the rotations of the level loop).
template <bool SWAP>
__device__ __forceinline__ void cell_level_shared(const unsigned char *lds, __amdgpu_buffer_rsrc_t vr, const uint4 (&g)[4],
                                                  const float4 &la, const float4 &lb, const float4 &wa, int H, int W, int wx0,
                                                  int wy0, int pitch, int wbase, unsigned lvl_byte, int row_bytes, int crn,
                                                  int rot, float4 &ra, float4 &rb, float4 &rw)
{
    // this lane's point of the level
    const float x0 = quad_bcast<0>(la.x), x1 = quad_bcast<0>(la.z), x2 = quad_bcast<0>(lb.x), x3 = quad_bcast<0>(lb.z);
    const float y0 = quad_bcast<0>(la.y), y1 = quad_bcast<0>(la.w), y2 = quad_bcast<0>(lb.y), y3 = quad_bcast<0>(lb.w);
    const float w0 = quad_bcast<0>(wa.x), w1 = quad_bcast<0>(wa.y), w2 = quad_bcast<0>(wa.z), w3 = quad_bcast<0>(wa.w);
    const float x = crn == 0 ? x0 : crn == 1 ? x1 : crn == 2 ? x2 : x3;
    const float y = crn == 0 ? y0 : crn == 1 ? y1 : crn == 2 ? y2 : y3;
    const float w = crn == 0 ? w0 : crn == 1 ? w1 : crn == 2 ? w2 : w3;
    const float Hf = (float)H, Wf = (float)W;
    const float h_im = fmaf(y, Hf, -0.5f), w_im = fmaf(x, Wf, -0.5f);
    const bool inside = (h_im > -1.f) && (w_im > -1.f) && (h_im < Hf) && (w_im < Wf);     // .cuh:285 (NaN -> false)
    const float hs = inside ? h_im : 0.f, ws = inside ? w_im : 0.f;
    const float hf = floorf(hs), wf = floorf(ws);
    const float lh = hs - hf, lw = ws - wf, hh = 1.f - lh, hw = 1.f - lw;
    const int ih = (int)hf, iw = (int)wf;
    const bool yok0 = inside && (unsigned)ih < (unsigned)H, yok1 = inside && (unsigned)(ih + 1) < (unsigned)H;
    const bool xok0 = (unsigned)iw < (unsigned)W, xok1 = (unsigned)(iw + 1) < (unsigned)W;
    const int okmask = (yok0 && xok0 ? 1 : 0) | (yok0 && xok1 ? 2 : 0) | (yok1 && xok0 ? 4 : 0) | (yok1 && xok1 ? 8 : 0);
    const bool staged = wbase >= 0;                                               // (wave-uniform)
    // base of the sample's top-left corner (may lie outside the window / level: only valid corners are read) and this
    // lane's corner offset from it
    const int base = staged ? wbase + (__mul24(ih - wy0, pitch) + (iw - wx0)) * 64
                            : (int)(lvl_byte + (unsigned)__mul24(__mul24(ih, W) + iw, row_bytes));
    const int delta = staged ? ((crn >> 1) * pitch + (crn & 1)) * 64 : ((crn >> 1) * W + (crn & 1)) * row_bytes;
    float e1 = 0.f, e2 = 0.f, e3 = 0.f, e4 = 0.f;                                 // the dots of THIS lane's sample
#define MSDA_CELL_SHARED_SAMPLE(S)                                                                                      \
    {                                                                                                                   \
        const int b_ = __builtin_amdgcn_update_dpp(0, base, MSDA_QUAD_PERM(S, S, S, S), 0xf, 0xf, true);              \
        const int m_ = __builtin_amdgcn_update_dpp(0, okmask, MSDA_QUAD_PERM(S, S, S, S), 0xf, 0xf, true);            \
        const f32x4 e = cell_corner_dot<SWAP>(lds, vr, g, staged, b_, delta, m_, crn, rot);                            \
        const float d1 = SWAP ? e[0] : quad_bcast<0>(e[0]), d2 = SWAP ? e[1] : quad_bcast<1>(e[0]);                    \
        const float d3 = SWAP ? e[2] : quad_bcast<2>(e[0]), d4 = SWAP ? e[3] : quad_bcast<3>(e[0]);                    \
        const bool mine = crn == S;                                                                                     \
        e1 = mine ? d1 : e1; e2 = mine ? d2 : e2; e3 = mine ? d3 : e3; e4 = mine ? d4 : e4;                             \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
    }
    MSDA_CELL_SHARED_SAMPLE(0)
    MSDA_CELL_SHARED_SAMPLE(1)
    MSDA_CELL_SHARED_SAMPLE(2)
    MSDA_CELL_SHARED_SAMPLE(3)
#undef MSDA_CELL_SHARED_SAMPLE
    const float wgt = inside ? w : 0.f;
    const float a_ = inside ? hh * (hw * e1 + lw * e2) + lh * (hw * e3 + lw * e4) : 0.f;
    const float gx = Wf * wgt * (hh * (e2 - e1) + lh * (e4 - e3));
    const float gy = Hf * wgt * (hw * (e3 - e1) + lw * (e4 - e2));
    // point p's results live in lane p: the level's vectors for its owner lane
    ra = make_float4(quad_bcast<0>(gx), quad_bcast<0>(gy), quad_bcast<1>(gx), quad_bcast<1>(gy));
    rb = make_float4(quad_bcast<2>(gx), quad_bcast<2>(gy), quad_bcast<3>(gx), quad_bcast<3>(gy));
    rw = make_float4(quad_bcast<0>(a_), quad_bcast<1>(a_), quad_bcast<2>(a_), quad_bcast<3>(a_));
}

// rotate the quad's per-lane data by one lane: lane j takes lane j + 1's registers, so that after l rotations quad lane 0
// holds level l (the level loop stays a loop: unrolled over levels and read routes the kernel was 36 KB of code and every
// phase ran 3-4x slower than its instruction count -- instruction-cache misses between workgroups in different phases)
__device__ __forceinline__ void quad_rotate4(float4 &v)
{
    constexpr int R = MSDA_QUAD_PERM(1, 2, 3, 0);
    v.x = dpp_quad<R>(v.x); v.y = dpp_quad<R>(v.y); v.z = dpp_quad<R>(v.z); v.w = dpp_quad<R>(v.w);
}

// MODE (experiments, ablation build only): 0 = the product kernel, 1 = sample geometry once per quad
// (cell_level_shared), 2 = that + the operand swap of cell_corner_dot, 3 = that + the window copies issued up front
// + level starts as scalars in the binning, 4 = mode 3 without the operand swap (every primitive then is one the product
// kernel already runs on the hardware)
template <int REFDIM, int MODE>
__global__ __launch_bounds__(kCellThreads, 4) void cell_backward_kernel(
    PatchPlan pl, const bf16_t *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ starts,
    const float *__restrict__ loc, const float *__restrict__ aw, const bf16_t *__restrict__ grad_out, int N, int S, int M,
    int Lq, unsigned value_bytes, float *__restrict__ g_loc, float *__restrict__ g_aw, const float *__restrict__ ref,
    bf16_t *__restrict__ g_qproj, uint32_t *__restrict__ masks, float *__restrict__ recs, int *__restrict__ ctl, int dbg)
{
    MSDA_DYNAMIC_LDS(unsigned char, clds);    // [zeros | windows | bin table]
    __shared__ int rng[kL][4];
    __shared__ int box[kL][4];
    __shared__ int winfo[kL][8];              // per level: H, W, window x0, y0, pitch, LDS base (< 0: gathered directly), start
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int cells = pl.CY * pl.CX, NM = N * M;
    // (image, head) pairs grouped by XCD (hardware block b runs on XCD b % 8): neighbouring cells share window rows
    int nm, c;
    if ((NM & 7) == 0) {
        const int per = NM >> 3, idx = blockIdx.x >> 3;
        nm = (blockIdx.x & 7) * per + idx % per;
        c = idx / per;
    } else {
        nm = blockIdx.x % NM;
        c = blockIdx.x / NM;
    }
    const int n = nm / M, m = nm % M;
    uint32_t *tab = reinterpret_cast<uint32_t *>(clds + kZeroBytes + kWinBytes);
    if (tid < kZeroBytes / 4) reinterpret_cast<int *>(clds)[tid] = 0;
    unsigned long long ts_last = clock64();
    (void)ts_last;

    // ---- phase 1: masks, records, bounding boxes -----------------------------------------------------------------------
    bin_cell<kCellThreads, true, (MODE >= 3)>(pl, starts, loc, aw, n, m, c, M, Lq, tab, rng, box, recs, ctl, dbg);
    CTS(0);
    {
        const TableLayout tl = table_layout(rng);
        write_masks<kCellThreads>(pl, tl, tab, n, m, c, M, masks);
    }
    CTS(1);

    if (MSDA_DBG(dbg) & 1) return;                                             // ablation: binning only
#ifdef MSDA_ABLATION
    // Arm RLIPV2_CELL_FAR_RETURN (round 6): a "far" sample anywhere in the call (as far as this workgroup can
    // see by now) means the patch pass will return without writing and the sorting pass takes grad_value -- and with this arm a
    // K1 launch gated on the same word writes EVERY location / weight gradient (launch_quad_backward_gated), so what this
    // workgroup would compute from here on is overwritten: stop.  One lane reads the word, the workgroup agrees through LDS (a
    // workgroup must return as a whole: barriers follow).  The final result is a pure function of the input: whether the word
    // is set at the end of the launch is, and then K1 has written everything.
    if (dbg & 16) {
        __shared__ int far_seen;
        if (tid == 0) far_seen = atomicOr(ctl + kFarWord, 0);
        __syncthreads();
        if (far_seen != 0) return;
    }
#endif
    // ---- phase 2: the windows of the four levels, smallest first, as long as they fit ----------------------------------
    int wx0[kL], wy0[kL], wpitch[kL], wbase[kL], wcols[kL], wrows[kL];
    {
        int npx[kL];
#pragma unroll
        for (int l = 0; l < kL; ++l) {
            // (wave-uniform: kept in scalar registers)
            const int x0 = __builtin_amdgcn_readfirstlane(box[l][0]), y0 = __builtin_amdgcn_readfirstlane(box[l][1]);
            const int x1 = -__builtin_amdgcn_readfirstlane(box[l][2]), y1 = -__builtin_amdgcn_readfirstlane(box[l][3]);
            const bool any = x0 <= x1 && y0 <= y1;
            wx0[l] = any ? x0 : 0; wy0[l] = any ? y0 : 0;
            wcols[l] = any ? x1 - x0 + 1 : 0; wrows[l] = any ? y1 - y0 + 1 : 0;
            wpitch[l] = wcols[l] + ((2 - wcols[l]) & 3);            // = 2 (mod 4): a quad's four corner rows on four bank quads
            npx[l] = wpitch[l] * wrows[l];
            wbase[l] = -1;
        }
        int used = 0;
#pragma unroll
        for (int round = 0; round < kL; ++round) {                    // smallest unplaced window first
            int best = -1, bestpx = 0x3fffffff;
#pragma unroll
            for (int l = 0; l < kL; ++l) {
                const bool cand = wbase[l] == -1 && npx[l] < bestpx;
                best = cand ? l : best; bestpx = cand ? npx[l] : bestpx;
            }
#pragma unroll
            for (int l = 0; l < kL; ++l) {
                if (l == best) {
                    if (used + npx[l] <= kMaxWinPx) { wbase[l] = kZeroBytes + used * 64; used += npx[l]; }
                    else wbase[l] = -2;                               // gathered directly
                }
            }
        }
    }
    const int row_bytes = M * 64;
    const unsigned img_byte = (unsigned)n * (unsigned)S * (unsigned)row_bytes + (unsigned)(m * 64);
    if (tid == 0) {
#pragma unroll
        for (int l = 0; l < kL; ++l) {
            winfo[l][0] = pl.H[l]; winfo[l][1] = pl.W[l]; winfo[l][2] = wx0[l]; winfo[l][3] = wy0[l];
            winfo[l][4] = wpitch[l]; winfo[l][5] = wbase[l]; winfo[l][6] = (int)starts[l]; winfo[l][7] = 0;
        }
    }
    if (MODE >= 3) {
        // (experiment) every copy of the thread, all levels, issued before the first LDS store: the loop below waits a
        // full memory latency per pass (6 passes for a full 48 KB window: the 13-15 k cycles of the timeline)
        int cnt[kL + 1];
        cnt[0] = 0;
#pragma unroll
        for (int l = 0; l < kL; ++l) cnt[l + 1] = cnt[l] + (wbase[l] >= 0 ? wrows[l] * wcols[l] * 4 : 0);
        constexpr int IT = kMaxWinPx * 4 / kCellThreads;               // pieces per thread when the budget is full
        static_assert(IT * kCellThreads == kMaxWinPx * 4, "staging passes");
        static_assert(IT == 6, "the copies below are written out six times");
        const int total = cnt[kL];
        // (index clamped instead of a branch around the load -- an empty cell reads pixel (0, 0) of its level for nothing --
        //  so that the six loads stay back to back; the stores are predicated)
        auto piece_of = [&](int k, int &dst) -> const uint4 * {
            const int i = min(tid + k * kCellThreads, max(total - 1, 0));
            const int l = (i >= cnt[1] ? 1 : 0) + (i >= cnt[2] ? 1 : 0) + (i >= cnt[3] ? 1 : 0);
            auto sel = [l](int a0, int a1, int a2, int a3) { return l == 0 ? a0 : l == 1 ? a1 : l == 2 ? a2 : a3; };
            const int local = i - sel(cnt[0], cnt[1], cnt[2], cnt[3]);
            const int cols = max(sel(wcols[0], wcols[1], wcols[2], wcols[3]), 1);
            const int pix = local >> 2, piece = local & 3;
            const int wy = (int)(((float)pix + 0.5f) * (1.f / (float)cols)), wx = pix - wy * cols;
            const int W = sel(pl.W[0], pl.W[1], pl.W[2], pl.W[3]);
            const int start = sel((int)starts[0], (int)starts[1], (int)starts[2], (int)starts[3]);
            const int gy = sel(wy0[0], wy0[1], wy0[2], wy0[3]) + wy, gx = sel(wx0[0], wx0[1], wx0[2], wx0[3]) + wx;
            const size_t src = (size_t)img_byte + (size_t)(start + gy * W + gx) * row_bytes + piece * 16;
            dst = max(sel(wbase[0], wbase[1], wbase[2], wbase[3]), 0) +
                  (wy * sel(wpitch[0], wpitch[1], wpitch[2], wpitch[3]) + wx) * 64 + piece * 16;
            return reinterpret_cast<const uint4 *>(reinterpret_cast<const unsigned char *>(value) + src);
        };
        int d0, d1, d2, d3, d4, d5;
        const uint4 *s0 = piece_of(0, d0), *s1 = piece_of(1, d1), *s2 = piece_of(2, d2);
        const uint4 *s3 = piece_of(3, d3), *s4 = piece_of(4, d4), *s5 = piece_of(5, d5);
        const uint4 v0 = *s0, v1 = *s1, v2 = *s2, v3 = *s3, v4 = *s4, v5 = *s5;
        if (tid < total) *reinterpret_cast<uint4 *>(clds + d0) = v0;
        if (tid + kCellThreads < total) *reinterpret_cast<uint4 *>(clds + d1) = v1;
        if (tid + 2 * kCellThreads < total) *reinterpret_cast<uint4 *>(clds + d2) = v2;
        if (tid + 3 * kCellThreads < total) *reinterpret_cast<uint4 *>(clds + d3) = v3;
        if (tid + 4 * kCellThreads < total) *reinterpret_cast<uint4 *>(clds + d4) = v4;
        if (tid + 5 * kCellThreads < total) *reinterpret_cast<uint4 *>(clds + d5) = v5;
    } else
#pragma unroll
    for (int l = 0; l < kL; ++l) {
        if (wbase[l] < 0 || wcols[l] == 0) continue;
        const int W = pl.W[l], start = (int)starts[l];
        const int pieces = wrows[l] * wcols[l] * 4;
        const float inv = 1.f / (float)wcols[l];
        for (int i = tid; i < pieces; i += kCellThreads) {
            const int pix = i >> 2, piece = i & 3;
            const int wy = (int)(((float)pix + 0.5f) * inv), wx = pix - wy * wcols[l];
            const size_t src = (size_t)img_byte + (size_t)(start + (wy0[l] + wy) * W + wx0[l] + wx) * row_bytes + piece * 16;
            *reinterpret_cast<uint4 *>(clds + wbase[l] + (wy * wpitch[l] + wx) * 64 + piece * 16) =
                *reinterpret_cast<const uint4 *>(reinterpret_cast<const unsigned char *>(value) + src);
        }
    }
    __syncthreads();
    CTS(2);
    if (MSDA_DBG(dbg) & 2) return;                                             // ablation: binning + staging

    // ---- phase 3: a quad per query, 16 queries per wave and task ---------------------------------------------------------
    const __amdgpu_buffer_rsrc_t vr = __builtin_amdgcn_make_buffer_rsrc((void *)value, 0, value_bytes, 0x00020000);
    const int crn = lane & 3, rot = (lane >> 2) & 3;
    const int cy = c / pl.CX, cx = c % pl.CX;
    for (int task = wave; task * 16 < kCellQ; task += kCellThreads / 64) {
        const int j = task * 16 + (lane >> 2);
        const int lq = j < 256 ? 0 : j < 320 ? 1 : j < 336 ? 2 : 3;
        const int r = j - (lq == 0 ? 0 : lq == 1 ? 256 : lq == 2 ? 320 : 336);
        const int sh = 4 - lq;
        const int iy = (cy << sh) + (r >> sh), ix = (cx << sh) + (r & ((1 << sh) - 1));
        const int Hq = lq == 0 ? pl.H[0] : lq == 1 ? pl.H[1] : lq == 2 ? pl.H[2] : pl.H[3];
        const int Wq = lq == 0 ? pl.W[0] : lq == 1 ? pl.W[1] : lq == 2 ? pl.W[2] : pl.W[3];
        // (MODE >= 3: from LDS -- the compiler folds the plain select chain into the vector load starts[lq], whose
        //  latency then sits IN FRONT of the task's operand loads: two memory round trips per task instead of one)
        const int stq = MODE >= 3 ? winfo[lq][6]      // (the level starts are in LDS for the level loop anyway)
                      : lq == 0 ? (int)starts[0] : lq == 1 ? (int)starts[1] : lq == 2 ? (int)starts[2] : (int)starts[3];
        const bool live = j < kCellQ && iy < Hq && ix < Wq;          // (a quad is live or dead as a whole)
        const int q = live ? stq + iy * Wq + ix : 0;
        const long qm = ((long)n * Lq + q) * M + m;
        // quad lane c loads level c's 4 points and weights (whole 16-byte vectors, 192 contiguous bytes per quad)
        const float4 *loc4 = reinterpret_cast<const float4 *>(loc) + qm * 8 + crn * 2;
        float4 la = loc4[0], lb = loc4[1];
        float4 wa = reinterpret_cast<const float4 *>(aw)[qm * 4 + crn];
        uint4 g[4];                                                  // the query's grad_out row, pieces in rotated order
#pragma unroll
        for (int t = 0; t < 4; ++t) g[t] = reinterpret_cast<const uint4 *>(grad_out + qm * kD)[t ^ rot];
#ifdef MSDA_ABLATION
        if (MSDA_DBG(dbg) & 16) { MSDA_ASM_WAIT_VM(); CTS(3); }     // task operands have arrived
#endif
        float4 gla = make_float4(0.f, 0.f, 0.f, 0.f), glb = gla, ga = gla;
#pragma unroll 1
        for (int l = 0; l < kL; ++l) {
            const int H = __builtin_amdgcn_readfirstlane(winfo[l][0]), W = __builtin_amdgcn_readfirstlane(winfo[l][1]);
            const int x0w = __builtin_amdgcn_readfirstlane(winfo[l][2]), y0w = __builtin_amdgcn_readfirstlane(winfo[l][3]);
            const int pitch = __builtin_amdgcn_readfirstlane(winfo[l][4]), base = __builtin_amdgcn_readfirstlane(winfo[l][5]);
            const unsigned lvl_byte = img_byte + (unsigned)__mul24(__builtin_amdgcn_readfirstlane(winfo[l][6]), row_bytes);
            const bool own = crn == l;
            if (MODE != 0) {                     // (experiment, off by default: geometry once per quad)
                float4 ra, rb, rw;
                cell_level_shared<MODE == 2 || MODE == 3>(clds, vr, g, la, lb, wa, H, W, x0w, y0w, pitch, base, lvl_byte, row_bytes, crn, rot, ra, rb, rw);
                if (own) { gla = ra; glb = rb; ga = rw; }
                quad_rotate4(la); quad_rotate4(lb); quad_rotate4(wa);
                continue;
            }
            float a_, x_, y_;
            // (a scheduling fence per sample: the compiler otherwise hoists every corner read of the level and spills; pairs
            //  of samples between fences were measured: no faster, scratch in the fused variants)
            cell_sample(clds, vr, g, quad_bcast<0>(la.x), quad_bcast<0>(la.y), quad_bcast<0>(wa.x), H, W, x0w, y0w, pitch, base,
                        lvl_byte, row_bytes, crn, rot, a_, x_, y_, dbg);
            ga.x = own ? a_ : ga.x; gla.x = own ? x_ : gla.x; gla.y = own ? y_ : gla.y;
            __builtin_amdgcn_sched_barrier(0);
            cell_sample(clds, vr, g, quad_bcast<0>(la.z), quad_bcast<0>(la.w), quad_bcast<0>(wa.y), H, W, x0w, y0w, pitch, base,
                        lvl_byte, row_bytes, crn, rot, a_, x_, y_, dbg);
            ga.y = own ? a_ : ga.y; gla.z = own ? x_ : gla.z; gla.w = own ? y_ : gla.w;
            __builtin_amdgcn_sched_barrier(0);
            cell_sample(clds, vr, g, quad_bcast<0>(lb.x), quad_bcast<0>(lb.y), quad_bcast<0>(wa.z), H, W, x0w, y0w, pitch, base,
                        lvl_byte, row_bytes, crn, rot, a_, x_, y_, dbg);
            ga.z = own ? a_ : ga.z; glb.x = own ? x_ : glb.x; glb.y = own ? y_ : glb.y;
            __builtin_amdgcn_sched_barrier(0);
            cell_sample(clds, vr, g, quad_bcast<0>(lb.z), quad_bcast<0>(lb.w), quad_bcast<0>(wa.w), H, W, x0w, y0w, pitch, base,
                        lvl_byte, row_bytes, crn, rot, a_, x_, y_, dbg);
            ga.w = own ? a_ : ga.w; glb.z = own ? x_ : glb.z; glb.w = own ? y_ : glb.w;
            __builtin_amdgcn_sched_barrier(0);
            quad_rotate4(la); quad_rotate4(lb); quad_rotate4(wa);        // (after 4 rotations: the lane's own level again)
        }
        if (REFDIM == 0) {
            if (live) {
                float4 *gl4 = reinterpret_cast<float4 *>(g_loc) + qm * 8 + crn * 2;
                gl4[0] = gla;
                gl4[1] = glb;
                reinterpret_cast<float4 *>(g_aw)[qm * 4 + crn] = ga;
            }
        } else if (live) {
            constexpr int RD = REFDIM == 0 ? 2 : REFDIM;
            const long row = qm / M;
            const float a[4] = {wa.x, wa.y, wa.z, wa.w};
            float gq[4] = {ga.x, ga.y, ga.z, ga.w};
            const float gl[8] = {gla.x, gla.y, gla.z, gla.w, glb.x, glb.y, glb.z, glb.w};
            geom::backward<bf16_t, RD>(g_qproj + row * (M * 48), ref + row * (kL * RD), shapes, m, M, crn, a, gq, gl);
        }
#ifdef MSDA_ABLATION
        if (MSDA_DBG(dbg) & 16) CTS(4);                                        // samples + epilogue of the task
#endif
    }
    CTS(5);
}

#ifdef MSDA_ABLATION
}  // namespace
}  // namespace msda
extern "C" int msda_debug_cell_timeline(void *host, int reset)
{
    if (reset) { unsigned long long z[16] = {0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(msda::cell_ts), z, sizeof(z)); }
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(msda::cell_ts), sizeof(msda::cell_ts));
}
namespace msda {
namespace {
#endif

// ---- cell_forward_kernel: source in msda_cell_forward.inc (shared with the host-side lane-level model) ----------------
#define MSDA_DEVFN __device__ __forceinline__
#define MSDA_KERNEL_BOUNDS(T, W) __global__ __launch_bounds__(T, W)
#define MSDA_LDS_DYNAMIC(name) MSDA_DYNAMIC_LDS_ALIGNED(unsigned char, name, 256)
#define MSDA_LDS_STATIC(type, name, dims) __shared__ type name dims
#define MSDA_TID threadIdx.x
#define MSDA_BID blockIdx.x
#define MSDA_READFIRSTLANE(x) __builtin_amdgcn_readfirstlane(x)
#define MSDA_UPDATE_DPP(old, v, ctrl, row_mask) __builtin_amdgcn_update_dpp(old, v, ctrl, row_mask, 0xf, false)
#define MSDA_SYNCTHREADS() __syncthreads()
#define MSDA_LDS_ATOMIC_MIN(p, v) atomicMin(p, v)
#define MSDA_LDS_ADDR(p) MSDA_LDS_BYTE_ADDR(p)
#define MSDA_WAVE_FENCE() do { __builtin_amdgcn_wave_barrier(); MSDA_ASM_FENCE(); } while (0)
#ifndef MSDA_EMU
#define MSDA_TR_READ_PAIR(b0, b1, addr0, addr1)                                                                      \
    do {                                                                                                              \
        b0 = lds_tr_read(addr0); b1 = lds_tr_read(addr1);                                                            \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b0), "+v"(b1) : : "memory");                                      \
    } while (0)
#else
#define MSDA_TR_READ_PAIR(b0, b1, addr0, addr1) do { b0 = lds_tr_read(addr0); b1 = lds_tr_read(addr1); } while (0)
#endif
#define MSDA_MFMA444(a, b, c) __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a, b, c, 0, 0, 0)
#include "msda_cell_forward.inc"
#include "msda_cell_records.inc"
#undef MSDA_DEVFN
#undef MSDA_KERNEL_BOUNDS
#undef MSDA_LDS_DYNAMIC
#undef MSDA_LDS_STATIC
#undef MSDA_TID
#undef MSDA_BID
#undef MSDA_READFIRSTLANE
#undef MSDA_UPDATE_DPP
#undef MSDA_SYNCTHREADS
#undef MSDA_LDS_ATOMIC_MIN
#undef MSDA_LDS_ADDR
#undef MSDA_WAVE_FENCE
#undef MSDA_TR_READ_PAIR
#undef MSDA_MFMA444

#ifdef MSDA_ABLATION
// grad_out in the group records' cell-major order (the CELLG arm of the patch pass): one workgroup per (image, head, cell)
// copies the 64-byte rows of the cell's <= 340 queries, four lanes per row.  Gated like the pass that reads the copy.
__global__ __launch_bounds__(256) void grad_out_cells_kernel(PatchPlan pl, const int64_t *__restrict__ starts,
                                                             const bf16_t *__restrict__ grad_out, int N, int M, int Lq,
                                                             bf16_t *__restrict__ gcell, const int *__restrict__ ctl)
{
    if (ctl[kFarWord] != 0) return;
    const int cells = pl.CY * pl.CX;
    const int c = blockIdx.x % cells, nm = blockIdx.x / cells;
    const int n = nm / M, m = nm % M;
    const int cy = c / pl.CX, cx = c % pl.CX;
    const int lsH0 = pl.H[0], lsH1 = pl.H[1], lsH2 = pl.H[2], lsH3 = pl.H[3];
    const int lsW0 = pl.W[0], lsW1 = pl.W[1], lsW2 = pl.W[2], lsW3 = pl.W[3];
    const int lsS0 = (int)starts[0], lsS1 = (int)starts[1], lsS2 = (int)starts[2], lsS3 = (int)starts[3];
    for (int i = threadIdx.x; i < kCellQ * 4; i += 256) {
        const int j = i >> 2, piece = i & 3;
        bool live;
        const int q = cell_query(MSDA_LS_ARGS, cy, cx, j, live);
        if (!live) continue;
        const uint4 *src = reinterpret_cast<const uint4 *>(grad_out + (((size_t)n * Lq + q) * M + m) * kD);
        uint4 *dst = reinterpret_cast<uint4 *>(gcell + (((size_t)nm * cells + c) * kCellQ + j) * kD);
        dst[piece] = src[piece];
    }
}
#endif

// ---- host side -------------------------------------------------------------------------------------------------------
bool make_patch_plan(const Problem &p, const int64_t *hs, PatchPlan &pl)
{
    if (!hs || p.L != kL || p.P != kP || p.D != kD || p.dtype != MSDA_BF16 || p.Lq != p.S) return false;
    long sum = 0;
    pl.CY = 1; pl.CX = 1;
    for (int l = 0; l < kL; ++l) {
        const int64_t H = hs[2 * l], W = hs[2 * l + 1];
        if (H < 1 || W < 1 || H > 4096 || W > 4096) return false;
        pl.H[l] = (int)H; pl.W[l] = (int)W;
        const int cs = 16 >> l;
        pl.CY = std::max(pl.CY, (int)((H + cs - 1) / cs));
        pl.CX = std::max(pl.CX, (int)((W + cs - 1) / cs));
        pl.PY[l] = (int)((H + 3) / 4); pl.PX[l] = (int)((W + 3) / 4);
        sum += H * W;
    }
    if (sum != p.S) return false;
    static const int kRad[kL] = {1, 2, 3, 6};
    long slots = 0;
    int items = 0;
    for (int l = 0; l < kL; ++l) {
        pl.rad[l] = kRad[l];
        const int nb = 2 * kRad[l] + 1 + (l == 3 ? 1 : 0);
        pl.nby[l] = std::min(nb, pl.CY); pl.nbx[l] = std::min(nb, pl.CX);
        if (pl.nby[l] * pl.nbx[l] > 128) return false;            // 7-bit slot in the 16-bit candidate code
        pl.invx[l] = (65536 + pl.nbx[l] - 1) / pl.nbx[l];
        pl.sbase[l] = (int)slots;
        slots += (long)pl.PY[l] * pl.PX[l] * pl.nby[l] * pl.nbx[l];
        if (slots >= (1L << 30)) return false;
    }
    pl.slots = (int)slots;
    for (int l = kL - 1; l >= 0; --l) {
        // expected MFMA steps per patch: a (query, level) group touches ~3.5 patches
        const double steps = (double)p.Lq * 3.5 / ((double)pl.PY[l] * pl.PX[l]) / kStep;
        pl.parts[l] = steps > 16.0 ? 4 : steps > 6.0 ? 2 : 1;
        pl.ibase[l] = items;
        // (experiment, off by default: on the fine levels -- a patch is 2-3 MFMA steps -- a wave takes several patches in
        //  turn with the mask words of the next one prefetched, instead of 4x as many waves that start with an exposed load)
        pl.reps[l] = (pl.parts[l] == 1 && steps < 4.0) ? ablation_env("RLIPV2_PATCH_REPS", kPatchReps) : 1;
        const int per = kWaves / pl.parts[l] * pl.reps[l];
        pl.nitems[l] = (pl.PY[l] * pl.PX[l] + per - 1) / per;
        items += pl.nitems[l];
    }
    pl.items = items;
    if ((long)items * p.N * p.M >= (1L << 30)) return false;
    // LDS of bin2_kernel: the largest table over all cells
    int worst = 0;
    for (int cy = 0; cy < pl.CY; ++cy)
        for (int cx = 0; cx < pl.CX; ++cx) {
            int words = 0;
            for (int l = 0; l < kL; ++l) {
                int ny = 0, nx = 0;
                for (int t = 0; t < pl.PY[l]; ++t) {
                    const int o = nb_origin(l, t, pl.rad[l], pl.nby[l], pl.CY);
                    ny += (o <= cy && cy < o + pl.nby[l]);
                }
                for (int t = 0; t < pl.PX[l]; ++t) {
                    const int o = nb_origin(l, t, pl.rad[l], pl.nbx[l], pl.CX);
                    nx += (o <= cx && cx < o + pl.nbx[l]);
                }
                words += ny * nx * kSlotWords;
            }
            worst = std::max(worst, words);
        }
    pl.bin_lds = worst * 4;
    if (pl.bin_lds > 60 * 1024) return false;
    return true;
}

}  // namespace

static size_t mask_bytes(const Problem &p, const PatchPlan &pl) { return (size_t)p.N * p.M * pl.slots * kSlotWords * 4; }
static size_t rec_bytes(const Problem &p, const PatchPlan &pl)
{
    return (size_t)p.N * p.M * kL * pl.CY * pl.CX * kCellQ * 48;
}

bool patch_supports(const Problem &p, const int64_t *shapes_host)
{
    PatchPlan pl;
    if (!make_patch_plan(p, shapes_host, pl)) return false;
    return mask_bytes(p, pl) + rec_bytes(p, pl) <= ((size_t)1 << 31);
}

// (ablation build: + room for the cell-major grad_out copy of the CELLG arm, behind the masks and group records)
static size_t gcell_bytes(const Problem &p, const PatchPlan &pl)
{
#ifdef MSDA_ABLATION
    return (size_t)p.N * p.M * pl.CY * pl.CX * kCellQ * kD * 2;
#else
    (void)p; (void)pl;
    return 0;
#endif
}

size_t patch_workspace_bytes(const Problem &p, const int64_t *shapes_host)
{
    PatchPlan pl;
    if (!patch_supports(p, shapes_host) || !make_patch_plan(p, shapes_host, pl)) return 0;
    return mask_bytes(p, pl) + rec_bytes(p, pl) + gcell_bytes(p, pl);
}

size_t patch_gcell_offset(const Problem &p, const int64_t *shapes_host)
{
    PatchPlan pl;
    if (!patch_supports(p, shapes_host) || !make_patch_plan(p, shapes_host, pl) || gcell_bytes(p, pl) == 0) return 0;
    return mask_bytes(p, pl) + rec_bytes(p, pl);
}

int patch_plan_info(const Problem &p, const int64_t *shapes_host, int32_t *out, int out_len)
{
    PatchPlan pl;
    if (!patch_supports(p, shapes_host) || !make_patch_plan(p, shapes_host, pl)) return 0;
    constexpr int kFields = 14, kTail = 5;
    if (!out || out_len < kL * kFields + kTail) return -1;
    for (int l = 0; l < kL; ++l) {
        const int v[kFields] = {pl.H[l], pl.W[l], pl.PY[l], pl.PX[l], pl.rad[l], pl.nby[l], pl.nbx[l], pl.invx[l],
                                pl.sbase[l], pl.parts[l], pl.reps[l], pl.ibase[l], pl.nitems[l], 16 >> l};
        for (int k = 0; k < kFields; ++k) out[l * kFields + k] = v[k];
    }
    int32_t *t = out + kL * kFields;
    t[0] = pl.CY; t[1] = pl.CX; t[2] = pl.slots; t[3] = pl.items; t[4] = pl.bin_lds;
    return kL * kFields + kTail;
}

bool cell_forward_supports(const Problem &p, const int64_t *shapes_host)
{
    PatchPlan pl;
    if (!p.value || !make_patch_plan(p, shapes_host, pl)) return false;       // bfloat16, D 32, L = P = 4, Lq == S, consistent pyramid
    if ((long)p.N * p.Lq * p.M * 16 >= (1L << 31)) return false;                // 32-bit sample indices
    if ((long)p.N * p.M * pl.CY * pl.CX >= (1L << 30)) return false;
    return quad_supports(p);                                                    // the window staging addresses value through a buffer resource
}

// ---- the records buffer: what cell_forward_kernel<., EMIT> leaves for the backward pass (one allocation, saved by the caller)
//   [control block 256 B | window table 128 B per (image, head, cell) | sample records: 2 B per sample, kRecQ query slots per
//    (.., cell, level) | patch masks | group records]  (the last two exactly as launch_patch_dest expects them: masks, then records)
struct RecordsLayout { size_t wtab, srec, masks, total; };
constexpr size_t kRecCtlBytes = 256;
static RecordsLayout records_layout(const Problem &p, const PatchPlan &pl)
{
    const size_t items = (size_t)p.N * p.M * pl.CY * pl.CX;
    RecordsLayout r;
    r.wtab = kRecCtlBytes;
    r.srec = r.wtab + items * kL * 8 * 4;
    r.masks = r.srec + items * kL * kRecQ * kP * sizeof(srec_t);       // (a multiple of 128 bytes per item)
    r.total = r.masks + mask_bytes(p, pl) + rec_bytes(p, pl);
    return r;
}

bool cell_records_supports(const Problem &p, const int64_t *shapes_host)
{
    PatchPlan pl;
    if (!cell_forward_supports(p, shapes_host) || !patch_supports(p, shapes_host) || !make_patch_plan(p, shapes_host, pl)) return false;
    if (pl.bin_lds > kFwdWinBytes || !quad_supports(p)) return false;         // the forward's mask table; buffer addressing of the direct route
    return (size_t)p.N * p.M * pl.CY * pl.CX * kL * kRecQ * kP < ((size_t)1 << 31);      // 32-bit record indices
}

size_t cell_records_bytes(const Problem &p, const int64_t *shapes_host)
{
    PatchPlan pl;
    if (!cell_records_supports(p, shapes_host) || !make_patch_plan(p, shapes_host, pl)) return 0;
    return records_layout(p, pl).total;
}

// out = forward pass of an encoder call, cell_forward_kernel (explicit variant MSDA_VARIANT_CELL); f != nullptr: the module's
// operands (projection rows + reference points), f->loc_save / f->aw_save receive the float32 locations / weights.
// records != nullptr (cell_records_bytes): the kernel also leaves the backward pass's records, masks and group records there
void launch_cell_forward(const Problem &p, const int64_t *shapes_host, const Fused *f, void *records)
{
    PatchPlan pl;
    make_patch_plan(p, shapes_host, pl);
    const dim3 grid(p.N * p.M * pl.CY * pl.CX), block(kCellThreads);
    unsigned char *rb = reinterpret_cast<unsigned char *>(records);
    const RecordsLayout rl = records_layout(p, pl);
    if (rb && hipMemsetAsync(rb, 0, kRecCtlBytes, p.stream) != hipSuccess) return;      // (the error stays recorded)
    int *rctl = reinterpret_cast<int *>(rb);
    int *wtab = rb ? reinterpret_cast<int *>(rb + rl.wtab) : nullptr;
    srec_t *srec = rb ? reinterpret_cast<srec_t *>(rb + rl.srec) : nullptr;
    uint32_t *masks = rb ? reinterpret_cast<uint32_t *>(rb + rl.masks) : nullptr;
    float *grecs = rb ? reinterpret_cast<float *>(rb + rl.masks + mask_bytes(p, pl)) : nullptr;
#define MSDA_CELL_FWD(RD, EMIT, LOC, AW)                                                                              \
    do {                                                                                                              \
        RLIPV2_ONCE_PER_DEVICE((void)hipFuncSetAttribute((const void *)cell_forward_kernel<RD, EMIT>,                 \
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, kFwdLds));       \
        hipLaunchKernelGGL((cell_forward_kernel<RD, EMIT>), grid, block, kFwdLds, p.stream, pl, (const bf16_t *)p.value, p.starts, \
                           (float *)(LOC), (float *)(AW), p.N, p.S, p.M, p.Lq, (bf16_t *)p.out,                       \
                           (const bf16_t *)(f ? f->qproj : nullptr), f ? f->ref : nullptr, p.shapes, wtab, srec, masks, grecs, rctl); \
    } while (0)
    if (!rb) {
        if (!f) MSDA_CELL_FWD(0, 0, const_cast<void *>(p.loc), const_cast<void *>(p.aw));      // (REFDIM 0 only reads them)
        else if (f->refdim == 2) MSDA_CELL_FWD(2, 0, f->loc_save, f->aw_save);
        else MSDA_CELL_FWD(4, 0, f->loc_save, f->aw_save);
    } else if (f && !f->loc_save) {                          // the records are the whole saved state
        if (f->refdim == 2) MSDA_CELL_FWD(2, 3, nullptr, nullptr);
        else MSDA_CELL_FWD(4, 3, nullptr, nullptr);
    } else {
        if (!f) MSDA_CELL_FWD(0, 2, const_cast<void *>(p.loc), const_cast<void *>(p.aw));
        else if (f->refdim == 2) MSDA_CELL_FWD(2, 2, f->loc_save, f->aw_save);
        else MSDA_CELL_FWD(4, 2, f->loc_save, f->aw_save);
    }
#undef MSDA_CELL_FWD
}

// The backward pass from the records: cell_records_backward_kernel (grad_sampling_loc / grad_attn_weight, or the projection
// row's gradient with f) + the matrix-core patch pass on the masks / group records the forward wrote.  `gate` (returned) is the
// "far sample" word of the records' control block: non-zero -> the patch pass has returned without writing and the caller's
// sorting pass must produce grad_value.
const int *launch_cell_records_backward(const Problem &p, const Fused *f, const int64_t *shapes_host, const void *records,
                                        bool out_bf16, bool swap, void *gcell_ws)
{
    PatchPlan pl;
    make_patch_plan(p, shapes_host, pl);
    const RecordsLayout rl = records_layout(p, pl);
    unsigned char *rb = reinterpret_cast<unsigned char *>(const_cast<void *>(records));
    int *rctl = reinterpret_cast<int *>(rb);
    const int *wtab = reinterpret_cast<const int *>(rb + rl.wtab);
    const srec_t *srec = reinterpret_cast<const srec_t *>(rb + rl.srec);
    const float *grecs = reinterpret_cast<const float *>(rb + rl.masks + mask_bytes(p, pl));
    const unsigned vbytes = (unsigned)((size_t)p.N * p.S * p.M * kD * 2);
    const dim3 grid(p.N * p.M * pl.CY * pl.CX), block(kCellThreads);
    // (ablation build, RLIPV2_PATCH_CELLG: the kernel also leaves the cell-major grad_out copy of the patch pass's CELLG arm)
    bf16_t *gcell = (gcell_ws && ablation_env("RLIPV2_PATCH_CELLG", 0) != 0) ? reinterpret_cast<bf16_t *>(gcell_ws) : nullptr;
#define MSDA_REC_K(RD, SWAP)                                                                                          \
    do {                                                                                                              \
        RLIPV2_ONCE_PER_DEVICE((void)hipFuncSetAttribute((const void *)cell_records_backward_kernel<RD, SWAP>,        \
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, kFwdWinBytes));  \
        hipLaunchKernelGGL((cell_records_backward_kernel<RD, SWAP>), grid, block, kFwdWinBytes, p.stream, pl,         \
                           (const bf16_t *)p.value, p.shapes, p.starts, srec, wtab, grecs,                            \
                           (const bf16_t *)p.grad_out, p.N, p.S, p.M, p.Lq, vbytes, (float *)p.g_loc, (float *)p.g_aw, \
                           f ? f->ref : nullptr, (bf16_t *)(f ? f->g_qproj : nullptr), gcell);                        \
    } while (0)
#define MSDA_REC(RD) do { if (swap) MSDA_REC_K(RD, true); else MSDA_REC_K(RD, false); } while (0)
    if (!f) MSDA_REC(0);
    else if (f->refdim == 2) MSDA_REC(2);
    else MSDA_REC(4);
#undef MSDA_REC
#undef MSDA_REC_K
    launch_patch_dest(p, shapes_host, rctl, rb + rl.masks, out_bf16, true, gcell_ws, gcell != nullptr);
    return rctl + kFarWord;
}

// float32 sampling_loc / attn_weight of the call rebuilt from the group records (only if the gate word is set): for the sorting
// fallback of a records call whose forward did not save them
void launch_records_unbin(const Problem &p, const int64_t *shapes_host, const void *records, float *loc, float *aw, const int *gate)
{
    PatchPlan pl;
    make_patch_plan(p, shapes_host, pl);
    const RecordsLayout rl = records_layout(p, pl);
    const float *grecs = reinterpret_cast<const float *>(reinterpret_cast<const unsigned char *>(records) + rl.masks + mask_bytes(p, pl));
    hipLaunchKernelGGL(records_unbin_kernel, dim3(p.N * p.M * pl.CY * pl.CX), dim3(256), 0, p.stream, pl, p.starts, grecs, p.N, p.M,
                       p.Lq, loc, aw, gate);
}

bool cell_backward_supports(const Problem &p, const int64_t *shapes_host)
{
    PatchPlan pl;
    if (!patch_supports(p, shapes_host) || !make_patch_plan(p, shapes_host, pl)) return false;
    if (pl.bin_lds > kCellTableMax || !quad_supports(p)) return false;      // (buffer addressing limits of the direct route)
    return true;
}

// grad_sampling_loc / grad_attn_weight (f == nullptr) or the projection row's gradient (fused geometry) + masks + records
void launch_cell_backward(const Problem &p, const Fused *f, const int64_t *shapes_host, int *ctl, void *mask_ws)
{
    PatchPlan pl;
    make_patch_plan(p, shapes_host, pl);
    uint32_t *masks = reinterpret_cast<uint32_t *>(mask_ws);
    float *recs = reinterpret_cast<float *>(reinterpret_cast<unsigned char *>(mask_ws) + mask_bytes(p, pl));
    const int lds_bytes = kZeroBytes + kWinBytes + pl.bin_lds;
    const unsigned vbytes = (unsigned)((size_t)p.N * p.S * p.M * kD * 2);
    const dim3 grid(p.N * p.M * pl.CY * pl.CX), block(kCellThreads);
    static const int mode = ablation_env("RLIPV2_CELL_SHARED", kCellMode);
#define MSDA_CELL_K(RD, MODE)                                                                                         \
    do {                                                                                                              \
        RLIPV2_ONCE_PER_DEVICE((void)hipFuncSetAttribute((const void *)cell_backward_kernel<RD, MODE>,                \
                                                         hipFuncAttributeMaxDynamicSharedMemorySize,                  \
                                                         kZeroBytes + kWinBytes + kCellTableMax));                    \
        hipLaunchKernelGGL((cell_backward_kernel<RD, MODE>), grid, block, lds_bytes, p.stream, pl, (const bf16_t *)p.value, \
                           p.shapes, p.starts, (const float *)p.loc, (const float *)p.aw, (const bf16_t *)p.grad_out, \
                           p.N, p.S, p.M, p.Lq, vbytes, (float *)p.g_loc, (float *)p.g_aw, f ? f->ref : nullptr,      \
                           (bf16_t *)(f ? f->g_qproj : nullptr), masks, recs, ctl,                                    \
                           ablation_env("RLIPV2_CELL_DBG", 0) | (ablation_env("RLIPV2_CELL_FAR_RETURN", 0) ? 16 : 0)); \
    } while (0)
#ifdef MSDA_ABLATION
#define MSDA_CELL(RD)                                                                                                 \
    do {                                                                                                              \
        if (mode == 1) MSDA_CELL_K(RD, 1); else if (mode == 2) MSDA_CELL_K(RD, 2);                                    \
        else if (mode == 3) MSDA_CELL_K(RD, 3); else if (mode == 4) MSDA_CELL_K(RD, 4); else MSDA_CELL_K(RD, 0);      \
    } while (0)
#else
#define MSDA_CELL(RD) MSDA_CELL_K(RD, kCellMode)
#endif
    (void)mode;
    if (!f) MSDA_CELL(0);
    else if (f->refdim == 2) MSDA_CELL(2);
    else MSDA_CELL(4);
#undef MSDA_CELL_K
#undef MSDA_CELL
}

// ctl: the control block of launch_backward_dest (zeroed by the caller on the stream), masks: patch_workspace_bytes;
// binned: cell_backward_kernel has already written the masks and records; gcell_ws (ablation build, may be null): room for the
// cell-major grad_out copy of the CELLG arm (patch_gcell_offset)
// gcell_filled: the copy is there already (cell_records_backward_kernel wrote it)
void launch_patch_dest(const Problem &p, const int64_t *shapes_host, int *ctl, void *mask_ws, bool out_bf16, bool binned, void *gcell_ws,
                       bool gcell_filled)
{
    PatchPlan pl;
    make_patch_plan(p, shapes_host, pl);
    uint32_t *masks = reinterpret_cast<uint32_t *>(mask_ws);
    float *recs = reinterpret_cast<float *>(reinterpret_cast<unsigned char *>(mask_ws) + mask_bytes(p, pl));
    RLIPV2_ONCE_PER_DEVICE((void)hipFuncSetAttribute((const void *)bin2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 60 * 1024));
    if (!binned)
        hipLaunchKernelGGL(bin2_kernel, dim3(p.N * pl.CY * pl.CX * p.M), dim3(kBinThreads), pl.bin_lds, p.stream, pl, p.starts,
                           (const float *)p.loc, (const float *)p.aw, p.M, p.Lq, masks, recs, ctl);
    const int grid = pl.items * p.N * p.M;
    static const int wps = ablation_env("RLIPV2_PATCH_WPS", 4);
#define MSDA_PATCH_ARGS(OT)                                                                                          \
    pl, p.starts, (const float *)recs, (const bf16_t *)p.grad_out, masks, (const int *)ctl, (OT *)p.g_value, p.N, p.S, p.M, p.Lq, \
        ablation_env("RLIPV2_PATCH_DBG", 0)
#define MSDA_PATCH(KERNEL, OT, WPS)                                                                                  \
    hipLaunchKernelGGL((KERNEL<OT, WPS>), dim3(grid), dim3(kThreads), kWaves * kWaveLds, p.stream, MSDA_PATCH_ARGS(OT))
#ifdef MSDA_ABLATION
#define MSDA_PATCH_MULTI(OT, CELLG)                                                                                  \
    hipLaunchKernelGGL((patch_dest_multi_kernel<OT, 4, CELLG>), dim3(grid), dim3(kThreads), kWaves * kWaveLds, p.stream, \
                       MSDA_PATCH_ARGS(OT), (const bf16_t *)gcell)
    bool multi = ablation_env("RLIPV2_PATCH_MULTI", kPatchMulti) != 0;      // (the experiment kernel, also with 1 patch per wave)
    for (int l = 0; l < kL; ++l) multi = multi || pl.reps[l] > 1;
    bf16_t *gcell = nullptr;
    if (gcell_ws && ablation_env("RLIPV2_PATCH_CELLG", 0) != 0) {           // the experiment kernel on a cell-major grad_out copy
        gcell = reinterpret_cast<bf16_t *>(gcell_ws);
        if (!gcell_filled)
        hipLaunchKernelGGL(grad_out_cells_kernel, dim3(p.N * p.M * pl.CY * pl.CX), dim3(256), 0, p.stream, pl, p.starts,
                           (const bf16_t *)p.grad_out, p.N, p.M, p.Lq, gcell, (const int *)ctl);
        multi = true;
    }
    if (multi) {
        if (gcell) { if (out_bf16) MSDA_PATCH_MULTI(bf16_t, true); else MSDA_PATCH_MULTI(float, true); }
        else { if (out_bf16) MSDA_PATCH_MULTI(bf16_t, false); else MSDA_PATCH_MULTI(float, false); }
        return;
    }
#undef MSDA_PATCH_MULTI
#else
    (void)gcell_ws; (void)gcell_filled;
#endif
    if (out_bf16) { if (wps == 5) MSDA_PATCH(patch_dest_kernel, bf16_t, 5); else MSDA_PATCH(patch_dest_kernel, bf16_t, 4); }
    else { if (wps == 5) MSDA_PATCH(patch_dest_kernel, float, 5); else MSDA_PATCH(patch_dest_kernel, float, 4); }
#undef MSDA_PATCH
#undef MSDA_PATCH_ARGS
}

}  // namespace msda
