// msda_window.hip -- LDS-windowed MSDA kernels for the encoder (self-attention) call on gfx950:
// queries are the pixels of the pyramid themselves (Lq == S), D = 32, L = 4, P = 4.
//
// Why: in the backward pass every sample scatters 4 corners x 32 channels of float32 into
// grad_value.  Done with global atomics that is 45 M scattered 128-byte read-modify-writes per
// launch at batch 4 and is bound by the L2 atomic units (measured on MI355X: 34 ms for the
// direct-scatter kernel of msda_quad.hip, 4.6 ms for the wave-per-(q,m) generic kernel) against
// ~0.1 ms of HBM time for the algorithmic bytes.  The encoder's queries are spatially ordered
// and its sampling offsets are a few pixels per level (reference initialisation:
// models/ops/modules/ms_deform_attn.py:66-74), so the scatter targets of a 2-D tile of queries
// form a small window of the level: this kernel accumulates that window in LDS with LDS atomics
// and flushes it once with coalesced global atomics.
//
// Work decomposition
//   block  = (image n, head m, 16x16 query tile of one pyramid level, ONE sampled level l)
//            -> its scatter targets all lie in level l, one window, one LDS buffer (<= 80 KB,
//            two blocks per CU); grad_sampling_loc / grad_attn_weight of the 4 points of level
//            l are written by exactly this block.
//   thread = quad layout of msda_quad.hip: 4 lanes per (query, head), 8 channels each; the four
//            lanes own the four points of the level and share them with DPP broadcasts.
//   window = bounding box of the corners the block's samples actually touch (computed in a
//            first pass: wave reduction + LDS atomic min/max), clipped to the LDS capacity.
//            Corners outside the clipped box fall back to global atomics, so the result is
//            correct for ANY sampling locations -- the tile shape is only a locality guess.
//   XCD    = consecutive work items (all heads x levels of one query tile) are mapped to the
//            same XCD so that their loads of grad_out / loc and their partial-line stores of
//            grad_loc / grad_aw meet in one L2.
//
//   bands  = a box taller than the LDS capacity is processed in row bands (zero / accumulate /
//            flush per band) from per-point scatter records kept in registers, so no sample ever
//            falls back to scattered global atomics unless the box is wider than the capacity.
//
// LDS cells are 64-bit FIXED POINT, not float: measured on MI355X (tools/ubench/lds_atomics.hip)
// ds_add_f32 retires 0.38 lane-updates/clk/CU (it is ~36x slower than ds_add_u32 at 13.8;
// ds_add_f64 3.4, ds_add_u64 5.7-6.2), so float LDS atomics made this kernel SLOWER than global
// atomics (9.3 ms).  Each block scales its contributions by a power of two chosen from the
// largest |grad_out| it holds so that one contribution is a 31-bit integer (relative resolution
// 2^-30 of that maximum: finer than the float32 atomics of the reference) and adds it, sign
// extended, into a 64-bit cell (room for 2^33 contributions).  Integer addition is associative,
// so a window's sum does not depend on the order the lanes arrive in.
//
// LDS layout: pixel p of the window holds 32 cells; channel c sits at ((c + 4 * (p & 7)) & 31).
// A scatter instruction adds channel 4k + sub for every lane, so the lanes of a 16-lane group
// (4 quads = 4 neighbouring pixels x 4 lanes x 2 banks per cell) hit 32 different banks.
#include <cstdlib>

#include "msda_device.h"
#include "msda_internal.h"

namespace msda {

namespace {

constexpr int kL = 4, kP = 4, kD = 32;
constexpr int kTile = 16;                       // 16 x 16 queries per block
constexpr int kThreads = kTile * kTile * 4;     // 1024: one quad per query
constexpr int kWinCap = 318;                    // window capacity in pixels (x 256 B): 2 blocks fit in 160 KB
constexpr int kXcds = 8;

struct TileInfo {
    int lq, ty, tx;        // query level and tile coordinates
};

// number of 16x16 tiles of a level
__host__ __device__ inline int tiles_of(int H, int W) { return ((H + kTile - 1) / kTile) * ((W + kTile - 1) / kTile); }


template <typename VT>
__global__ __launch_bounds__(kThreads) void window_backward_kernel(
    const VT *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ starts,
    const float *__restrict__ loc, const float *__restrict__ aw, const VT *__restrict__ grad_out, int N, int S,
    int M, float *__restrict__ g_value, float *__restrict__ g_loc, float *__restrict__ g_aw, int dbg)
{
    extern __shared__ __attribute__((aligned(16))) long long win[];   // kWinCap * 32 cells + 8 ints
    int *box = reinterpret_cast<int *>(win + kWinCap * kD);          // {min_y, min_x, max_y, max_x, max|g| bits}

    // The pyramid shape lives on the device (as in the reference), so the grid cannot be sized
    // from it on the host: the launch is persistent -- a fixed number of blocks walks the items.
    int tiles_per_image = 0;
#pragma unroll
    for (int l = 0; l < kL; ++l) tiles_per_image += tiles_of((int)shapes[2 * l], (int)shapes[2 * l + 1]);
    const int total_items = N * M * kL * tiles_per_image;
    // XCD-aware walk: block b runs on XCD b % 8; each XCD owns a contiguous range of items and its
    // blocks take consecutive items, so the (head, level) items of one query tile run on one XCD
    // at about the same time.
    const int per_xcd = (total_items + kXcds - 1) / kXcds;
    const int xcd = blockIdx.x % kXcds, lane_blk = blockIdx.x / kXcds, blks = gridDim.x / kXcds;
    const int item_end = min(total_items, (xcd + 1) * per_xcd);
  for (int item = xcd * per_xcd + lane_blk; item < item_end; item += blks) {
    __syncthreads();                                   // previous item's flush has left the window
    const int lvl = item & 3;
    const int m = (item >> 2) % M;
    int t = (item / (4 * M)) % tiles_per_image;
    const int n = item / (4 * M * tiles_per_image);

    int lq = 0, Hq = 0, Wq = 0;
#pragma unroll
    for (int l = 0; l < kL; ++l) {
        const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
        const int nt = tiles_of(H, W);
        if (t >= 0 && t < nt) { lq = l; Hq = H; Wq = W; t -= 1 << 30; }   // found: park t below zero
        else if (t >= 0) t -= nt;
    }
    t += 1 << 30;
    const int tiles_x = (Wq + kTile - 1) / kTile;
    const int ty = t / tiles_x, tx = t % tiles_x;
    const int startq = (int)starts[lq];

    const int H = (int)shapes[2 * lvl], W = (int)shapes[2 * lvl + 1], start = (int)starts[lvl];

    // ---- which query / channels: quad layout -----------------------------------------------------
    const int tid = threadIdx.x;
    const int sub = tid & 3;
    const int quad = tid >> 2;
    const int qy = ty * kTile + (quad >> 4), qx = tx * kTile + (quad & 15);
    const bool live = qy < Hq && qx < Wq;
    const int q = live ? startq + qy * Wq + qx : startq;          // dead quads shadow a real query
    const long qm = ((long)n * S + q) * M + m;
    const long img = (long)n * S * M * kD;
    const VT *vimg = value + img;
    float *gimg = g_value + img;
    const int head_chan = m * kD + sub * 8;

    // lane `sub` loads point `sub` of level lvl
    const long sidx = (qm * kL + lvl) * kP + sub;
    const float2 xy = reinterpret_cast<const float2 *>(loc)[sidx];
    const float wgt_in = aw[sidx];
    float tg[8];
    Vec8<VT>::load(grad_out + qm * kD + sub * 8, tg);

    // grad_out a second time, channel-interleaved (channel 4k + sub): the scatter instructions then
    // cover 16 contiguous bytes per quad (LDS: conflict-free banks; global fallback: 4x fewer
    // 32-byte sectors per atomic instruction than an 8-channels-per-lane stride)
    float tgi[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) tgi[k] = Elem<float, VT>::ld(grad_out + qm * kD + 4 * k + sub);
    float gmax = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) gmax = fmaxf(gmax, live ? fabsf(tgi[k]) : 0.f);

    // ---- pass 1: geometry, value gathers, channel reductions; remember what to scatter --------------
    const int row = M * kD;
    int ys0[kP], xs0[kP];          // top-left corner (clamped) of each point; other corner = +1 (clamped)
    int ys1[kP], xs1[kP];
    float cw[kP][4];               // scatter weight per corner, 0 when the corner does not exist
    float my_ga = 0.f, my_gx = 0.f, my_gy = 0.f;
    int y_lo = 0x7fffffff, x_lo = 0x7fffffff, y_hi = -1, x_hi = -1;

    // (compile-time point index: runtime-indexed register arrays would be demoted to scratch)
#define MSDA_WIN_POINT(PT)                                                                                     \
    {                                                                                                          \
        const float x = quad_bcast<PT>(xy.x), y = quad_bcast<PT>(xy.y), w = quad_bcast<PT>(wgt_in);            \
        const float h_im = fmaf(y, (float)H, -0.5f), w_im = fmaf(x, (float)W, -0.5f);                          \
        const bool inside = live && (h_im > -1.f) && (w_im > -1.f) && (h_im < (float)H) && (w_im < (float)W);  \
        const float hs = inside ? h_im : 0.f, ws = inside ? w_im : 0.f;                                        \
        const float hf = floorf(hs), wf = floorf(ws);                                                          \
        const int h_low = (int)hf, w_low = (int)wf;                                                            \
        const float lh = hs - hf, lw = ws - wf, hh = 1.f - lh, hw = 1.f - lw;                                  \
        const bool hl = h_low >= 0, hh_ok = h_low + 1 <= H - 1, wl = w_low >= 0, wh_ok = w_low + 1 <= W - 1;   \
        const bool ok[4] = {inside && hl && wl, inside && hl && wh_ok, inside && hh_ok && wl,                  \
                            inside && hh_ok && wh_ok};                                                         \
        const float wgt = inside ? w : 0.f;                                                                    \
        ys0[PT] = max(h_low, 0); ys1[PT] = min(h_low + 1, H - 1);                                              \
        xs0[PT] = max(w_low, 0); xs1[PT] = min(w_low + 1, W - 1);                                              \
        const float bw[4] = {hh * hw, hh * lw, lh * hw, lh * lw};                                              \
        float e[4];                                                                                            \
        _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                                        \
            const int yy = (k >> 1) ? ys1[PT] : ys0[PT], xx = (k & 1) ? xs1[PT] : xs0[PT];                     \
            float v[8];                                                                                        \
            Vec8<VT>::load(vimg + ((dbg & 4) ? 0 : (start + yy * W + xx) * row) + head_chan, v);               \
            float d = v[0] * tg[0];                                                                            \
            _Pragma("unroll") for (int c = 1; c < 8; ++c) d = fmaf(v[c], tg[c], d);                            \
            e[k] = quad_sum(ok[k] ? d : 0.f);                                                                  \
            cw[PT][k] = ok[k] ? bw[k] * wgt : 0.f;                                                             \
        }                                                                                                      \
        if (inside) {                                                                                          \
            y_lo = min(y_lo, ys0[PT]); y_hi = max(y_hi, ys1[PT]);                                              \
            x_lo = min(x_lo, xs0[PT]); x_hi = max(x_hi, xs1[PT]);                                              \
        }                                                                                                      \
        const float g_a = hh * (hw * e[0] + lw * e[1]) + lh * (hw * e[2] + lw * e[3]);                         \
        const float g_w = (float)W * wgt * (hh * (e[1] - e[0]) + lh * (e[3] - e[2]));                          \
        const float g_h = (float)H * wgt * (hw * (e[2] - e[0]) + lw * (e[3] - e[1]));                          \
        if (sub == PT) { my_ga = g_a; my_gx = g_w; my_gy = g_h; }                                              \
        __builtin_amdgcn_sched_barrier(0); /* keep the next point's gathers from being hoisted (spills) */     \
    }
    MSDA_WIN_POINT(0)
    MSDA_WIN_POINT(1)
    MSDA_WIN_POINT(2)
    MSDA_WIN_POINT(3)
#undef MSDA_WIN_POINT
    if (live) {
        reinterpret_cast<float2 *>(g_loc)[sidx] = make_float2(my_gx, my_gy);
        g_aw[sidx] = my_ga;
    }

    // ---- bounding box of the block's scatter targets -----------------------------------------------
    if (tid < 5) box[tid] = tid < 2 ? 0x7fffffff : (tid < 4 ? -1 : 0);
    __syncthreads();
    int gbits = __float_as_int(gmax);       // non-negative floats order like their bit patterns
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        y_lo = min(y_lo, __shfl_xor(y_lo, off, 64)); x_lo = min(x_lo, __shfl_xor(x_lo, off, 64));
        y_hi = max(y_hi, __shfl_xor(y_hi, off, 64)); x_hi = max(x_hi, __shfl_xor(x_hi, off, 64));
        gbits = max(gbits, __shfl_xor(gbits, off, 64));
    }
    if ((tid & 63) == 0) {
        if (y_hi >= 0) {
            atomicMin(&box[0], y_lo); atomicMin(&box[1], x_lo);
            atomicMax(&box[2], y_hi); atomicMax(&box[3], x_hi);
        }
        atomicMax(&box[4], gbits);
    }
    __syncthreads();
    // fixed-point scale 2^fx: the block's largest |grad_out| (times a weight <= 1) stays below 2^30.
    // NaN / Inf gradients cannot be represented: such a block scatters with float global atomics.
    const int gexp = (box[4] >> 23) & 0xff;                   // biased exponent of the maximum
    const bool fixed_ok = gexp != 0xff;
    const int fx = 29 - (gexp - 127);
    float tgs[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) tgs[k] = ldexpf(tgi[k], fx);
    const int by0 = box[0], wx0 = box[1], by1 = box[2];
    const int ww = min(box[3] - wx0 + 1, kWinCap);          // columns beyond the cap: global atomics
    const int band = by1 >= 0 ? kWinCap / ww : 0;            // rows of the box one LDS window holds

    // ---- pass 2: the box in row bands -- zero, accumulate with LDS atomics, flush ---------------------
    if (!(dbg & 8))
    for (int wy0 = by0; wy0 <= by1; wy0 += band) {
        const int wh = min(band, by1 - wy0 + 1);
        const int npix = wh * ww;
        __syncthreads();                                   // previous band's flush is done
        for (int i = tid; i < npix * (kD / 2); i += kThreads)
            reinterpret_cast<uint4 *>(win)[i] = make_uint4(0u, 0u, 0u, 0u);
        __syncthreads();
#pragma unroll
        for (int pt = 0; pt < kP; ++pt) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float wk = cw[pt][k];
                // opaque to the optimiser: otherwise the 128 products wk * tgi[c] are hoisted out of
                // the band loop as loop invariants and spill to scratch
                asm volatile("" : "+v"(wk));
                const int yy = (k >> 1) ? ys1[pt] : ys0[pt], xx = (k & 1) ? xs1[pt] : xs0[pt];
                const int py = yy - wy0, px = xx - wx0;
                if (wk != 0.f && (unsigned)py < (unsigned)wh && !(dbg & 1)) {
                    if (px < ww && fixed_ok) {
                        const int p = py * ww + px;
                        unsigned long long *dst = reinterpret_cast<unsigned long long *>(win) + p * kD;
                        const int rot = 4 * (p & 7) + sub;
#pragma unroll
                        for (int c = 0; c < 8; ++c) {
                            const int q31 = __float2int_rn(wk * tgs[c]);          // |.| < 2^30
                            atomicAdd(dst + ((4 * c + rot) & 31), (unsigned long long)(long long)q31);
                        }
                    } else {
                        float *dst = gimg + (long)(start + yy * W + xx) * row + m * kD + sub;
#pragma unroll
                        for (int c = 0; c < 8; ++c) atomic_add(dst + 4 * c, wk * tgi[c]);
                    }
                }
            }
        }
        __syncthreads();
        // flush: one lane per (pixel, channel), 128 contiguous bytes per pixel, untouched lanes skipped
        const int c = tid & 31;
        for (int p = tid >> 5; p < npix; p += kThreads / 32) {
            const long long cell = win[p * kD + ((c + 4 * (p & 7)) & 31)];
            const float v = ldexpf((float)cell, -fx);
            if (cell != 0 && !(dbg & 2)) {
                const int py = p / ww, px = p - py * ww;
                atomic_add(gimg + (long)(start + (wy0 + py) * W + wx0 + px) * row + m * kD + c, v);
            }
        }
    }
  }   // item loop
}

}  // namespace

// Used when the call looks like encoder self-attention: Lq == S (the queries are the pixels of
// the pyramid), model head shape, and enough work to fill the chip.  The tile <-> query mapping
// is only a locality guess: results are correct for any sampling locations.
bool window_supports(const Problem &p, bool backward)
{
    if (!backward) return false;                       // forward: msda_quad.hip
    if (p.dtype != MSDA_F32 && p.dtype != MSDA_BF16) return false;
    if (p.D != kD || p.L != kL || p.P != kP) return false;
    if (p.Lq != p.S || p.S < 1) return false;
    if ((long)p.S * p.M * kD >= (1L << 31)) return false;
    if ((long)p.N * p.S * p.M * kL * kP >= (1L << 31)) return false;
    return true;
}

void launch_window_forward(const Problem &) {}

void launch_window_backward(const Problem &p)
{
    // persistent grid: 2 blocks of 1024 threads per CU (LDS- and wave-limited), 256 CUs
    const char *e = getenv("RLIPV2_MSDA_DEBUG");       // ablation switches for profiling only
    const int dbg = e ? atoi(e) : 0;
    const char *g = getenv("RLIPV2_MSDA_GRID");
    const int grid = g ? atoi(g) : 256 * 2;
    const size_t lds = (size_t)kWinCap * kD * sizeof(long long) + 32;
    if (p.dtype == MSDA_F32) {
        hipLaunchKernelGGL((window_backward_kernel<float>), dim3(grid), dim3(kThreads), lds, p.stream,
                           (const float *)p.value, p.shapes, p.starts, (const float *)p.loc, (const float *)p.aw,
                           (const float *)p.grad_out, p.N, p.S, p.M, (float *)p.g_value, (float *)p.g_loc,
                           (float *)p.g_aw, dbg);
    } else {
        hipLaunchKernelGGL((window_backward_kernel<bf16_t>), dim3(grid), dim3(kThreads), lds, p.stream,
                           (const bf16_t *)p.value, p.shapes, p.starts, (const float *)p.loc, (const float *)p.aw,
                           (const bf16_t *)p.grad_out, p.N, p.S, p.M, (float *)p.g_value, (float *)p.g_loc,
                           (float *)p.g_aw, dbg);
    }
}

}  // namespace msda
