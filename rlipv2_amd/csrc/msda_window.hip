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
// LDS layout: pixel p of the window holds 32 floats; channel c sits at ((c + (p & 7)) & 31) so
// that the 32 lanes of a half-wave (8 quads = 8 neighbouring pixels, 4 lanes x stride-8 channels)
// hit 32 different banks in one ds_add_f32.
#include "msda_device.h"
#include "msda_internal.h"

namespace msda {

namespace {

constexpr int kL = 4, kP = 4, kD = 32;
constexpr int kTile = 16;                       // 16 x 16 queries per block
constexpr int kThreads = kTile * kTile * 4;     // 1024: one quad per query
constexpr int kWinCap = 640;                    // window capacity in pixels (x 128 B = 80 KB)
constexpr int kXcds = 8;

struct TileInfo {
    int lq, ty, tx;        // query level and tile coordinates
};

// number of 16x16 tiles of a level
__host__ __device__ inline int tiles_of(int H, int W) { return ((H + kTile - 1) / kTile) * ((W + kTile - 1) / kTile); }

__device__ __forceinline__ int lds_slot(int p, int c) { return p * kD + ((c + (p & 7)) & 31); }

template <typename VT>
__global__ __launch_bounds__(kThreads) void window_backward_kernel(
    const VT *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ starts,
    const float *__restrict__ loc, const float *__restrict__ aw, const VT *__restrict__ grad_out, int N, int S,
    int M, float *__restrict__ g_value, float *__restrict__ g_loc, float *__restrict__ g_aw)
{
    extern __shared__ __attribute__((aligned(16))) float win[];   // kWinCap * 32 floats + 4 ints
    int *box = reinterpret_cast<int *>(win + kWinCap * kD);      // {min_y, min_x, max_y, max_x}

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

    // ---- pass 1: bounding box of the touched corners ----------------------------------------------
    if (tid < 4) box[tid] = tid < 2 ? 0x7fffffff : -1;
    __syncthreads();
    {
        const float h_im = fmaf(xy.y, (float)H, -0.5f), w_im = fmaf(xy.x, (float)W, -0.5f);
        const bool inside = live && (h_im > -1.f) && (w_im > -1.f) && (h_im < (float)H) && (w_im < (float)W);
        const int h0 = (int)floorf(inside ? h_im : 0.f), w0 = (int)floorf(inside ? w_im : 0.f);
        int y_lo = inside ? max(h0, 0) : 0x7fffffff, x_lo = inside ? max(w0, 0) : 0x7fffffff;
        int y_hi = inside ? min(h0 + 1, H - 1) : -1, x_hi = inside ? min(w0 + 1, W - 1) : -1;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            y_lo = min(y_lo, __shfl_xor(y_lo, off, 64)); x_lo = min(x_lo, __shfl_xor(x_lo, off, 64));
            y_hi = max(y_hi, __shfl_xor(y_hi, off, 64)); x_hi = max(x_hi, __shfl_xor(x_hi, off, 64));
        }
        if ((tid & 63) == 0) {
            atomicMin(&box[0], y_lo); atomicMin(&box[1], x_lo);
            atomicMax(&box[2], y_hi); atomicMax(&box[3], x_hi);
        }
    }
    __syncthreads();
    const int wy0 = box[0], wx0 = box[1];
    int wh = box[2] - wy0 + 1, ww = box[3] - wx0 + 1;
    if (box[2] < 0) { wh = 0; ww = 0; }                      // no sample of this block is inside the level
    if (ww > kWinCap) ww = kWinCap;
    if (wh * ww > kWinCap) wh = kWinCap / ww;                // clip rows; the rest uses global atomics
    const int npix = wh * ww;

    // ---- zero the window ---------------------------------------------------------------------------
    for (int i = tid; i < npix * (kD / 4); i += kThreads)
        reinterpret_cast<float4 *>(win)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();

    // ---- pass 2: the four points of this level --------------------------------------------------------
    const int row = M * kD;
    float my_ga = 0.f, my_gx = 0.f, my_gy = 0.f;

    auto one_point = [&](float x, float y, float w, float &g_a, float &g_w, float &g_h) {
        const float h_im = fmaf(y, (float)H, -0.5f), w_im = fmaf(x, (float)W, -0.5f);
        const bool inside = (h_im > -1.f) && (w_im > -1.f) && (h_im < (float)H) && (w_im < (float)W);
        const float hs = inside ? h_im : 0.f, ws = inside ? w_im : 0.f;
        const float hf = floorf(hs), wf = floorf(ws);
        const int h_low = (int)hf, w_low = (int)wf;
        const float lh = hs - hf, lw = ws - wf, hh = 1.f - lh, hw = 1.f - lw;
        const bool hl = h_low >= 0, hh_ok = h_low + 1 <= H - 1, wl = w_low >= 0, wh_ok = w_low + 1 <= W - 1;
        const bool ok[4] = {inside && hl && wl, inside && hl && wh_ok, inside && hh_ok && wl, inside && hh_ok && wh_ok};
        const float wgt = inside ? w : 0.f;
        const int ys[2] = {max(h_low, 0), min(h_low + 1, H - 1)};
        const int xs[2] = {max(w_low, 0), min(w_low + 1, W - 1)};
        const float cw[4] = {hh * hw * wgt, hh * lw * wgt, lh * hw * wgt, lh * lw * wgt};
        float e[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int yy = ys[k >> 1], xx = xs[k & 1];
            const int off = (start + yy * W + xx) * row + head_chan;
            float v[8];
            Vec8<VT>::load(vimg + off, v);
            float d = v[0] * tg[0];
#pragma unroll
            for (int c = 1; c < 8; ++c) d = fmaf(v[c], tg[c], d);
            e[k] = quad_sum(ok[k] ? d : 0.f);
            if (ok[k] && live) {
                const int py = yy - wy0, px = xx - wx0;
                if ((unsigned)py < (unsigned)wh && (unsigned)px < (unsigned)ww) {
                    const int p = py * ww + px;
#pragma unroll
                    for (int c = 0; c < 8; ++c) atomicAdd(&win[lds_slot(p, sub * 8 + c)], cw[k] * tg[c]);
                } else {
#pragma unroll
                    for (int c = 0; c < 8; ++c) atomic_add(gimg + off + c, cw[k] * tg[c]);
                }
            }
        }
        g_a = hh * (hw * e[0] + lw * e[1]) + lh * (hw * e[2] + lw * e[3]);
        g_w = (float)W * wgt * (hh * (e[1] - e[0]) + lh * (e[3] - e[2]));
        g_h = (float)H * wgt * (hw * (e[2] - e[0]) + lw * (e[3] - e[1]));
    };

    {
        float a, gw, gh;
        one_point(quad_bcast<0>(xy.x), quad_bcast<0>(xy.y), quad_bcast<0>(wgt_in), a, gw, gh);
        if (sub == 0) { my_ga = a; my_gx = gw; my_gy = gh; }
        one_point(quad_bcast<1>(xy.x), quad_bcast<1>(xy.y), quad_bcast<1>(wgt_in), a, gw, gh);
        if (sub == 1) { my_ga = a; my_gx = gw; my_gy = gh; }
        one_point(quad_bcast<2>(xy.x), quad_bcast<2>(xy.y), quad_bcast<2>(wgt_in), a, gw, gh);
        if (sub == 2) { my_ga = a; my_gx = gw; my_gy = gh; }
        one_point(quad_bcast<3>(xy.x), quad_bcast<3>(xy.y), quad_bcast<3>(wgt_in), a, gw, gh);
        if (sub == 3) { my_ga = a; my_gx = gw; my_gy = gh; }
    }
    if (live) {
        reinterpret_cast<float2 *>(g_loc)[sidx] = make_float2(my_gx, my_gy);
        g_aw[sidx] = my_ga;
    }
    __syncthreads();

    // ---- flush the window: one lane per (pixel, channel), 128 contiguous bytes per pixel ----------------
    const int c = tid & 31;
    for (int p = tid >> 5; p < npix; p += kThreads / 32) {
        const float v = win[lds_slot(p, c)];
        if (v != 0.f) {
            const int py = p / ww, px = p - py * ww;
            atomic_add(gimg + (long)(start + (wy0 + py) * W + wx0 + px) * row + m * kD + c, v);
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
    const int grid = 256 * 2;
    const size_t lds = (size_t)kWinCap * kD * sizeof(float) + 16;
    if (p.dtype == MSDA_F32) {
        hipLaunchKernelGGL((window_backward_kernel<float>), dim3(grid), dim3(kThreads), lds, p.stream,
                           (const float *)p.value, p.shapes, p.starts, (const float *)p.loc, (const float *)p.aw,
                           (const float *)p.grad_out, p.N, p.S, p.M, (float *)p.g_value, (float *)p.g_loc,
                           (float *)p.g_aw);
    } else {
        hipLaunchKernelGGL((window_backward_kernel<bf16_t>), dim3(grid), dim3(kThreads), lds, p.stream,
                           (const bf16_t *)p.value, p.shapes, p.starts, (const float *)p.loc, (const float *)p.aw,
                           (const bf16_t *)p.grad_out, p.N, p.S, p.M, (float *)p.g_value, (float *)p.g_loc,
                           (float *)p.g_aw);
    }
}

}  // namespace msda
