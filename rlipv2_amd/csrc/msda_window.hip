// msda_window.hip -- the grad_value scatter of the MSDA backward pass for gfx950
// (D = 32, L = 4, P = 4), as a destination-sorted, atomics-free LDS pass per query tile.
//
// Background (all measured on MI355X, batch 4, 800x1333 pyramid, see profiles/ and
// tools/ubench/lds_atomics.hip):
//   * every sample adds 4 corners x 32 channels of float32 into grad_value: 45.5 M scattered
//     128-byte read-modify-writes per encoder call.  The L2 atomic units retire ~42 G 32-byte
//     sector-updates/s: 4.6 ms with 128-byte rows per instruction (msda_generic.hip), 34 ms with
//     the quad layout's strided 4-byte atomics -- against ~0.1 ms of HBM time.
//   * LDS float atomics do not help: ds_add_f32 retires 0.38 lane-updates/clk/CU (integer
//     ds_add_u32 13.8, ds_add_u64 5.7, ds_add_f64 3.4); an LDS window of float cells took 9.3 ms,
//     one of 64-bit fixed-point cells 2.2 ms.
// So this kernel uses NO floating-point atomics in LDS at all.  The backward pass is split:
//   K1 (msda_quad.hip, quad_backward_kernel<.., SCATTER=false>): value gathers, the channel
//      reductions and grad_sampling_loc / grad_attn_weight -- embarrassingly parallel, no LDS.
//   K2 (this file): grad_value.  Needs only sampling_loc, attn_weight and grad_out.
//
// K2 work decomposition
//   item   = (image n, head m, tile of 256 queries, ONE sampled level l): all scatter targets of
//            an item lie in one level and one head.  For encoder self-attention (Lq == S: the
//            queries are the pixels of the pyramid) a tile is a 16x16 patch of one level, whose
//            targets form a compact window of level l; otherwise it is 256 consecutive queries.
//   thread = (query, point): computes its point's 4 bilinear corners (target pixel, weight).
//   sort   = counting sort of the item's <= 4096 (target, query, weight) records into 1024 buckets
//            keyed by the low 5 bits of the target's (y, x) -- LDS integer atomics for the histogram
//            and the slot cursor, a block scan for the offsets.  Equal targets always share a
//            bucket, and any window of up to 32 x 32 pixels maps to buckets one-to-one, so for
//            compact windows every target's records end up adjacent.
//   walk   = 32 lanes (one channel each) per chunk of 32 consecutive records read the source
//            queries' grad_out rows from LDS and sum every run of equal targets in registers; a
//            finished run leaves as ONE global atomic instruction covering its 128-byte row, so a
//            touched row is updated about once per item instead of once per sample.  Balancing by
//            records (not by pixels) keeps a coarse level's hot pixels from serialising one wave.
//   spread = when the targets are spread out (uniformly random locations, decoder queries) buckets
//            mix several targets and runs get short: in the limit every record is its own
//            128-byte row -- the row count of the generic kernel with all 64 lanes busy.
//   The result is exact for ANY sampling locations: tile shape and buckets are only locality guesses.
//   The launch is persistent (the pyramid shape lives on the device, as in the reference, so the
//   host cannot size a grid from it) and XCD-aware: the (head, level) items of one query tile go
//   to blocks of one XCD so their reads of loc / grad_out meet in one L2.
#include <cstdlib>

#include "msda_device.h"
#include "msda_internal.h"

#ifdef MSDA_K2_TIMELINE      // cycle stamps of workgroups 0 / 100 / 301 (timeline builds only, tools/k2_timeline.py)
__device__ unsigned long long k2_ts[3 * 32 * 10];
#define K2TS(k)                                                                                                  \
    do {                                                                                                         \
        if (tid == 0 && (blockIdx.x == 0 || blockIdx.x == 100 || blockIdx.x == 301) && ts_item < 32)           \
            k2_ts[(((blockIdx.x != 0) + (blockIdx.x == 301)) * 32 + ts_item) * 10 + (k)] = clock64();          \
    } while (0)
extern "C" int msda_debug_k2_timeline(void *host)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(k2_ts), sizeof(k2_ts));
}
#else
#define K2TS(k) do { } while (0)
#endif

namespace msda {

namespace {

constexpr int kL = 4, kP = 4, kD = 32;
constexpr int kTile = 16;                       // 16 x 16 queries per tile
constexpr int kQ = kTile * kTile;               // 256 queries per item
constexpr int kThreads = kQ * kP;               // 1024: one thread per (query, point)
constexpr int kRec = kQ * kP * 4;               // 4096 corner records per item
constexpr int kMaxPix = 1024;                   // sortable window size (pixels)
constexpr int kGoStride = kD;                   // floats per staged grad_out row (32 lanes read one row: no padding needed)
constexpr int kXcds = 8;

// LDS carve-up (bytes, every offset a multiple of 16)
constexpr int kOffGo = 0;                                  // float [256][32]          32768
constexpr int kOffRec = kOffGo + kQ * kGoStride * 4;       // uint2 [4096] (key, w)    32768
constexpr int kOffCnt = kOffRec + kRec * 8;                // int   [1024] histogram    4096
constexpr int kOffCur = kOffCnt + kMaxPix * 4;             // int   [1024] offsets/cursor 4096
constexpr int kOffMisc = kOffCur + kMaxPix * 4;            // int   [32]  box, wave sums  128
constexpr int kLdsBytes = kOffMisc + 128;                  // 73856 -> two blocks per CU

__host__ __device__ inline int tiles_of(int H, int W) { return ((H + kTile - 1) / kTile) * ((W + kTile - 1) / kTile); }

// what one thread needs from global memory for one item: its point and 8 channels of its query's grad_out
template <typename VT> struct ItemLoad {
    float2 xy;
    float wgt;
    typename Vec8<VT>::raw g;
    bool live;
};

template <typename VT, bool TILED>
__device__ __forceinline__ ItemLoad<VT> load_item(int item, int tiles_per_image, const int64_t *__restrict__ shapes,
                                                  const int64_t *__restrict__ starts, const float *__restrict__ loc,
                                                  const float *__restrict__ aw, const VT *__restrict__ grad_out,
                                                  int M, int Lq, int ql, int pt)
{
    ItemLoad<VT> it;
    const int lvl = item & 3;
    const int m = (item >> 2) % M;
    int t = (item / (4 * M)) % tiles_per_image;
    const int n = item / (4 * M * tiles_per_image);
    int q;
    if (TILED) {
        int lq = 0, Hq = 1, Wq = 1;
#pragma unroll
        for (int l = 0; l < kL; ++l) {
            const int Hl = (int)shapes[2 * l], Wl = (int)shapes[2 * l + 1];
            const int nt = tiles_of(Hl, Wl);
            if (t >= 0 && t < nt) { lq = l; Hq = Hl; Wq = Wl; t -= 1 << 30; }   // found: park t below zero
            else if (t >= 0) t -= nt;
        }
        t += 1 << 30;
        const int tiles_x = (Wq + kTile - 1) / kTile;
        const int qy = (t / tiles_x) * kTile + (ql >> 4), qx = (t % tiles_x) * kTile + (ql & 15);
        it.live = qy < Hq && qx < Wq;
        q = it.live ? (int)starts[lq] + qy * Wq + qx : 0;
    } else {
        q = t * kQ + ql;
        it.live = q < Lq;
        q = it.live ? q : 0;
    }
    const long qm = ((long)n * Lq + q) * M + m;
    const long sidx = (qm * kL + lvl) * kP + pt;
    it.xy = reinterpret_cast<const float2 *>(loc)[sidx];
    it.wgt = aw[sidx];
    it.g = Vec8<VT>::load_raw(grad_out + qm * kD + pt * 8);
    return it;
}

template <typename VT, bool TILED>
// 8 waves per SIMD = 64 VGPRs = TWO of these 1024-thread workgroups per CU (the LDS carve-up is sized for two): at 70
// VGPRs only one fits and the phases of an item (sort barriers, walk tail) have nothing to overlap with -- 870 vs 776 us
__global__ __launch_bounds__(kThreads, 8) void scatter_kernel(
    const int64_t *__restrict__ shapes, const int64_t *__restrict__ starts, const float *__restrict__ loc,
    const float *__restrict__ aw, const VT *__restrict__ grad_out, int N, int S, int M, int Lq,
    float *__restrict__ g_value, int dbg)
{
    MSDA_DYNAMIC_LDS(unsigned char, lds);
    float *go = reinterpret_cast<float *>(lds + kOffGo);
    uint2 *rec = reinterpret_cast<uint2 *>(lds + kOffRec);
    int *cnt = reinterpret_cast<int *>(lds + kOffCnt);
    int *cur = reinterpret_cast<int *>(lds + kOffCur);
    int *misc = reinterpret_cast<int *>(lds + kOffMisc);   // [5] record count, [8..23] wave sums

    const int tid = threadIdx.x;
    const int pt = tid & 3;                 // point of the level
    const int ql = tid >> 2;                // local query 0..255

    int tiles_per_image;
    if (TILED) {
        tiles_per_image = 0;
#pragma unroll
        for (int l = 0; l < kL; ++l) tiles_per_image += tiles_of((int)shapes[2 * l], (int)shapes[2 * l + 1]);
    } else {
        tiles_per_image = (Lq + kQ - 1) / kQ;
    }
    const int total_items = N * M * kL * tiles_per_image;
    const int per_xcd = (total_items + kXcds - 1) / kXcds;
    const int xcd = blockIdx.x % kXcds, lane_blk = blockIdx.x / kXcds, blks = gridDim.x / kXcds;
    const int item_end = min(total_items, (xcd + 1) * per_xcd);
    const int row = M * kD;

    int item = xcd * per_xcd + lane_blk;
    if (item >= item_end) return;
    ItemLoad<VT> nxt = load_item<VT, TILED>(item, tiles_per_image, shapes, starts, loc, aw, grad_out, M, Lq, ql, pt);

    int ts_item = -1;
    (void)ts_item;
    for (; item < item_end; item += blks) {
        ++ts_item;
        K2TS(0);
        const ItemLoad<VT> me = nxt;
        // software prefetch: the next item's operands travel while this item is sorted and walked
        if (item + blks < item_end)
            nxt = load_item<VT, TILED>(item + blks, tiles_per_image, shapes, starts, loc, aw, grad_out, M, Lq, ql, pt);

        const int lvl = item & 3;
        const int m = (item >> 2) % M;
        const int n = item / (4 * M * tiles_per_image);
        const int H = (int)shapes[2 * lvl], W = (int)shapes[2 * lvl + 1], start = (int)starts[lvl];
        float *gimg = g_value + (long)n * S * row + (long)start * row + m * kD;

        // ---- stage grad_out rows of the tile (float) ------------------------------------------------
        {
            float g[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            Vec8<VT>::fma(me.live ? 1.f : 0.f, me.g, g);
            float4 *dst = reinterpret_cast<float4 *>(go + ql * kGoStride + pt * 8);
            dst[0] = make_float4(g[0], g[1], g[2], g[3]);
            dst[1] = make_float4(g[4], g[5], g[6], g[7]);
        }
        cnt[tid] = 0;
        if (tid == 0) misc[5] = 0;

        // ---- this thread's point: 4 corners (record key, weight, sort bucket) ---------------------------
        // key = (level-relative target pixel << 15) | (local query << 7): the low 15 bits ARE the byte offset of
        // the query's staged grad_out row (128 bytes per row), so the walk needs one v_and_or per record for
        // the LDS address instead of shift + mask + add; bucket = low 5 bits of (y, x): equal
        // targets always share a bucket, and a window of up to 32 x 32 pixels maps to buckets 1:1.
        unsigned key[4];
        int bkt[4];
        float cw[4];
        {
            const float h_im = fmaf(me.xy.y, (float)H, -0.5f), w_im = fmaf(me.xy.x, (float)W, -0.5f);
            const bool inside = me.live && (h_im > -1.f) && (w_im > -1.f) && (h_im < (float)H) && (w_im < (float)W);
            const float hs = inside ? h_im : 0.f, ws = inside ? w_im : 0.f;
            const float hf = floorf(hs), wf = floorf(ws);
            const int h_low = (int)hf, w_low = (int)wf;
            const float lh = hs - hf, lw = ws - wf, hh = 1.f - lh, hw = 1.f - lw;
            const bool hl = h_low >= 0, hh_ok = h_low + 1 <= H - 1, wl = w_low >= 0, wh_ok = w_low + 1 <= W - 1;
            const float wgt = inside ? me.wgt : 0.f;
            const int cy[2] = {max(h_low, 0), min(h_low + 1, H - 1)};
            const int cx[2] = {max(w_low, 0), min(w_low + 1, W - 1)};
            cw[0] = (hl && wl) ? hh * hw * wgt : 0.f;       cw[1] = (hl && wh_ok) ? hh * lw * wgt : 0.f;
            cw[2] = (hh_ok && wl) ? lh * hw * wgt : 0.f;    cw[3] = (hh_ok && wh_ok) ? lh * lw * wgt : 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                key[k] = ((unsigned)(cy[k >> 1] * W + cx[k & 1]) << 15) | ((unsigned)ql << 7);
                bkt[k] = ((cy[k >> 1] & 31) << 5) | (cx[k & 1] & 31);
            }
        }
        __syncthreads();
        K2TS(1);

        // ---- counting sort by bucket: equal targets become adjacent records ----------------------------
        if (!(MSDA_DBG(dbg) & 16)) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (cw[k] != 0.f) atomicAdd(&cnt[bkt[k]], 1);
            __syncthreads();
            K2TS(2);
            {   // exclusive scan of cnt[0..1024) -> cur[]; one element per thread
                const int v = cnt[tid];
                int inc = v;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const int u = __shfl_up(inc, off, 64);
                    if ((tid & 63) >= off) inc += u;
                }
                if ((tid & 63) == 63) misc[8 + (tid >> 6)] = inc;
                __syncthreads();
                K2TS(3);
                // prefix over the 16 wave totals inside every wave (one LDS read + 4 shuffles instead of 16 LDS reads
                // and 16 selects per thread: this step took 2 100 of an item's 30 000 cycles, tools/k2_timeline.py)
                int ws = ((tid & 63) < kThreads / 64) ? misc[8 + (tid & 63)] : 0;
#pragma unroll
                for (int off = 1; off < kThreads / 64; off <<= 1) {
                    const int u = __shfl_up(ws, off, 64);
                    if ((tid & 63) >= off) ws += u;
                }
                const int wv = tid >> 6;
                const int base = wv ? __shfl(ws, wv - 1, 64) : 0;
                cur[tid] = base + inc - v;
                if (tid == kThreads - 1) misc[5] = base + inc;       // number of records
            }
            __syncthreads();
            K2TS(4);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (cw[k] != 0.f) rec[atomicAdd(&cur[bkt[k]], 1)] = make_uint2(key[k], __float_as_uint(cw[k]));
        } else {   // ablation: records in arrival order (every record its own run)
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (cw[k] != 0.f) rec[atomicAdd(&misc[5], 1)] = make_uint2(key[k], __float_as_uint(cw[k]));
        }
        __syncthreads();
        K2TS(5);

        // ---- walk: 32 lanes (one channel each) per chunk of consecutive records ----------------------------
        // A run of records with the same target is summed in registers and leaves as ONE 32-lane
        // global atomic instruction = one contiguous 128-byte row.  Measured cost model of float
        // global atomics on MI355X (tools/ubench/global_atomics.hip): ~21 clk per touched 128-byte
        // line + ~8 clk per 32-byte sector, per CU, whatever the lane count -- so a row must leave
        // in one instruction, not four.  Work is balanced by records, not by pixels: a coarse
        // level's hot pixel with hundreds of records is shared by many half-waves.
        // The list is walked in groups of 16; the last group's missing records are zero-weight copies of the last
        // record made in registers at load time (they extend its run and add nothing) -- no padding pass, no barrier.
        // (float32 keeps the padding pass: its wider prefetch registers make the in-register variant spill into the
        // walk -- 907 vs 877 us; bf16: 771 -> 763 us)
        constexpr bool kPadInRegisters = sizeof(VT) == 2;
        const int nrec_real = misc[5];
        const int nrec = (nrec_real + 15) & ~15;
        if (!kPadInRegisters) {
            if (nrec_real > 0 && tid < nrec - nrec_real) rec[nrec_real + tid] = make_uint2(rec[nrec_real - 1].x, 0u);
            __syncthreads();
        }
        if (!(MSDA_DBG(dbg) & 4)) {
            const int ch = tid & 31;
            const unsigned ch_byte = (unsigned)ch * 4u;
            const unsigned row_bytes = (unsigned)row * 4u;
            // LDS address of this lane's channel in staged row 0; `go` sits at LDS offset 0 of a kernel without static
            // LDS, so (row offset from the record key) | go_ch is the whole address: one v_and_or_b32 per record instead
            // of and + add (the trap guards the assumption)
#ifndef MSDA_EMU
            typedef const __attribute__((address_space(3))) float lds_cfloat;
#define K2_LDS_FLOAT(addr) (*(lds_cfloat *)(uintptr_t)(addr))
#else
#define K2_LDS_FLOAT(addr) (*reinterpret_cast<const float *>(emu::lds_ptr(addr)))     /* (host model, tools/emu/) */
#endif
            const unsigned go_ch = MSDA_LDS_BYTE_ADDR(go) + ch_byte;
            if (go_ch & 0x7f80u) __builtin_trap();
            char *gbytes = reinterpret_cast<char *>(gimg);
            // records per half-wave visit: every visit ends with a flush, so 64 instead of 32 removed ~1/5 of the row
            // atomics (893 -> 772 us once the walk itself was no longer the limit).  Now ONE visit per half-wave, sized to
            // the item (a multiple of the 16-record group): full tiles get 128, partly filled ones stay balanced
            // (780 -> 771 us against a fixed 64; RLIPV2_MSDA_DEBUG bit 32 restores the fixed size)
            const int kChunk = (MSDA_DBG(dbg) >> 8) ? (MSDA_DBG(dbg) >> 8) : ((MSDA_DBG(dbg) & 32) ? 64 : max(16, (((nrec + 31) / 32) + 15) & ~15));        // (profiling: RLIPV2_MSDA_DEBUG = chunk << 8; multiple of 16)
            // Records reach the 32 lanes of a half-wave through the registers, not through 32-fold broadcast reads:
            // every row of 16 lanes loads 16 consecutive records with ONE ds_read_b64 (lane j: record j) and record i
            // is handed to the row by DPP row_newbcast:i -- the key with a v_mov_dpp, the weight as the DPP operand of
            // the v_fmac itself.  The LDS pipe was the walk's limit (2 of 3 LDS cycles per record were these
            // broadcasts: tools/k2_timeline.py + the K2_NOBOUNDARY experiment); it is now 1/8 + 1 cycles per record.
            const int l16 = tid & 15;
            for (int base = (tid >> 5) * kChunk; base < nrec; base += (kThreads / 32) * kChunk) {
                const int end = min(base + kChunk, nrec);           // (end - base) is a multiple of 16
                float acc = 0.f;
                unsigned cur_px = rec[base].x >> 15;
                for (int e = base; e < end; e += 16) {
                    uint2 mine = rec[kPadInRegisters ? min(e + l16, nrec_real - 1) : e + l16];
                    if (kPadInRegisters) mine.y = (e + l16 < nrec_real) ? mine.y : 0u;
                    unsigned key[16];
                    float g[16];
// acc += (record I of the row's 16: its weight, as the DPP operand) * g -- one v_fmac_f32_dpp
#ifndef MSDA_EMU
#define K2_FMAC_BCAST(acc, w_bits, gval, I)                                                                          \
    asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:" #I " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(w_bits), "v"(gval))
#else
#define K2_FMAC_BCAST(acc, w_bits, gval, I)                                                                          \
    acc = fmaf(__uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)(w_bits), 0x150 + I, 0xf, 0xf, false)), gval, acc)
#endif
#define K2_KEY(I) key[I] = __builtin_amdgcn_update_dpp(0u, mine.x, 0x150 + I, 0xf, 0xf, false);
                    K2_KEY(0) K2_KEY(1) K2_KEY(2) K2_KEY(3) K2_KEY(4) K2_KEY(5) K2_KEY(6) K2_KEY(7)
                    K2_KEY(8) K2_KEY(9) K2_KEY(10) K2_KEY(11) K2_KEY(12) K2_KEY(13) K2_KEY(14) K2_KEY(15)
#undef K2_KEY
#pragma unroll
                    for (int i = 0; i < 16; ++i) g[i] = K2_LDS_FLOAT((key[i] & 0x7f80u) | go_ch);
#ifdef K2_NOBOUNDARY       // experiment: no run detection at all (wrong results; lower bound of the walk's cost)
#define K2_BOUNDARY(px) false
#else
#define K2_BOUNDARY(px) ((px) != cur_px)
#endif
#define K2_STEP(I)                                                                                                   \
                    {                                                                                                \
                        const unsigned px = key[I] >> 15;                                                            \
                        if (K2_BOUNDARY(px)) {                  /* the run of cur_px is complete */                  \
                            if (!(MSDA_DBG(dbg) & 2))                                                                          \
                                atomic_add(reinterpret_cast<float *>(gbytes + (size_t)(__umul24(cur_px, row_bytes) + ch_byte)), acc); \
                            acc = 0.f;                                                                               \
                            MSDA_ASM_OPAQUE(acc);               /* keep the reset inside the branch */               \
                            cur_px = px;                                                                             \
                        }                                                                                            \
                        K2_FMAC_BCAST(acc, mine.y, g[I], I);                                                         \
                    }
                    K2_STEP(0) K2_STEP(1) K2_STEP(2) K2_STEP(3) K2_STEP(4) K2_STEP(5) K2_STEP(6) K2_STEP(7)
                    K2_STEP(8) K2_STEP(9) K2_STEP(10) K2_STEP(11) K2_STEP(12) K2_STEP(13) K2_STEP(14) K2_STEP(15)
#undef K2_STEP
#undef K2_BOUNDARY
                }
                if (!(MSDA_DBG(dbg) & 2)) atomic_add(reinterpret_cast<float *>(gbytes + (size_t)(__umul24(cur_px, row_bytes) + ch_byte)), acc);
            }
        }
        K2TS(8);
        __syncthreads();      // the next item reuses go / rec / cnt
        K2TS(9);
    }
}

}  // namespace

// The scatter kernel handles any Lq; `window` as a variant name means: backward = K1 (reduce)
// + K2 (this scatter).  Levels larger than 2^24 pixels do not fit the direct path's record key.
bool window_supports(const Problem &p, bool backward)
{
    // forward: the window-staged tile kernel of msda_quad.hip, for encoder self-attention (queries = pixels)
    if (!backward && p.Lq != p.S) return false;
    if (p.dtype != MSDA_F32 && p.dtype != MSDA_BF16) return false;
    if (p.D != kD || p.L != kL || p.P != kP) return false;
    if (p.S < 1 || p.S >= (1 << 17)) return false;          // record key: 17 bits of level-relative pixel index
    if ((long)p.S * p.M * kD * 4 >= (1L << 32)) return false;   // 32-bit byte offsets inside one image's grad_value
    if ((long)p.N * p.Lq * p.M * kL * kP >= (1L << 31)) return false;
    return quad_supports(p);
}

void launch_window_forward(const Problem &p) { launch_tile_forward(p); }

void launch_window_backward(const Problem &p)
{
    // K1: grad_sampling_loc / grad_attn_weight (no scatter)
    launch_quad_backward_reduce(p);
    // K2: grad_value.  Persistent grid: 2 blocks of 1024 threads per CU (LDS- and wave-limited).
    const int dbg = ablation_env("RLIPV2_MSDA_DEBUG", 0);       // ablation builds only
    const int g = ablation_env("RLIPV2_MSDA_GRID", 0);
    const int grid = g ? g : 256 * 2;
    const bool tiled = p.Lq == p.S;                    // encoder self-attention: queries are the pixels
#define MSDA_LAUNCH_SCATTER(VT, TILED)                                                                         \
    hipLaunchKernelGGL((scatter_kernel<VT, TILED>), dim3(grid), dim3(kThreads), kLdsBytes, p.stream, p.shapes, \
                       p.starts, (const float *)p.loc, (const float *)p.aw, (const VT *)p.grad_out, p.N, p.S,  \
                       p.M, p.Lq, (float *)p.g_value, dbg)
    if (p.dtype == MSDA_F32) {
        if (tiled) MSDA_LAUNCH_SCATTER(float, true); else MSDA_LAUNCH_SCATTER(float, false);
    } else {
        if (tiled) MSDA_LAUNCH_SCATTER(bf16_t, true); else MSDA_LAUNCH_SCATTER(bf16_t, false);
    }
#undef MSDA_LAUNCH_SCATTER
}

}  // namespace msda
