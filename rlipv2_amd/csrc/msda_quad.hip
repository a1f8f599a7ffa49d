// msda_quad.hip -- direct-gather MSDA kernels for the model's head shape on gfx950:
// D = 32 channels per head, L = 4 levels, P = 4 points (16 samples per (query, head)).
//
// Mapping (wave64-native, not a 32-thread-warp tiling):
//   * four adjacent lanes (a DPP "quad") own one (n, q, m); each lane owns 8 of the 32
//     channels, i.e. ONE 16-byte vector of bf16 or TWO of f32 per bilinear corner, so a quad
//     reads a head's 64 B / 128 B corner row as one contiguous segment;
//   * a wavefront therefore covers 16 consecutive (q, m) = 2 queries x 8 heads, and its loads of
//     sampling locations (128 B per (q, m)), attention weights (64 B) and its output stores are
//     contiguous across the whole wave;
//   * each lane loads only its quarter of the 16 samples' (x, y, weight) and the quad shares
//     them with DPP quad_perm broadcasts (full-rate VALU modifiers, no LDS traffic);
//   * backward: the per-sample channel reductions (grad of attention weight / location) are
//     8 in-register FMAs per lane followed by a 2-step DPP quad reduction -- replacing the
//     reference's shared-memory staging + single-thread serial sum + 2 barriers per sample
//     (reference: ms_deform_im2col_cuda.cuh:356-394); grad_value uses hardware f32 atomics.
//
// Quad lane j loads the 4 points of level j (P = 4), the kernel loops over the levels and
// rotates those registers through the quad, so level l is always broadcast from quad lane 0.
// Per-level (H, W, start) are wave-uniform scalar loads from the device-resident int64
// metadata, exactly the operands the reference passes (ms_deform_attn_cuda.cu:67-68).
#include <cstdlib>

#include "msda_device.h"
#include "msda_internal.h"

namespace msda {

namespace {

constexpr int kBlock = 256;
constexpr int kL = 4, kP = 4, kLP = 16, kD = 32;

// Bilinear set-up of one sample for one lane.  Offsets are in elements relative to the image
// base and already include the head and this lane's channel group.
struct Corner4 {
    int o1, o2, o3, o4;
    float hh, hw, lh, lw;      // raw fractional weights
    bool ok1, ok2, ok3, ok4;   // corner inside the level (and sample included)
    float wgt;                 // attention weight, 0 when the sample is excluded
};

__device__ __forceinline__ Corner4 setup_sample(float x, float y, float w, int H, int W, int start, int M,
                                                int head_chan /* m*D + sub*8 */)
{
    Corner4 c;
    const float h_im = fmaf(y, (float)H, -0.5f);
    const float w_im = fmaf(x, (float)W, -0.5f);
    const bool inside = (h_im > -1.f) && (w_im > -1.f) && (h_im < (float)H) && (w_im < (float)W);
    const float hs = inside ? h_im : 0.f, ws = inside ? w_im : 0.f;
    const float hf = floorf(hs), wf = floorf(ws);
    const int h_low = (int)hf, w_low = (int)wf;
    c.lh = hs - hf; c.lw = ws - wf;
    c.hh = 1.f - c.lh; c.hw = 1.f - c.lw;
    const bool hl = h_low >= 0, hh_ok = h_low + 1 <= H - 1, wl = w_low >= 0, wh = w_low + 1 <= W - 1;
    c.ok1 = inside && hl && wl;    c.ok2 = inside && hl && wh;
    c.ok3 = inside && hh_ok && wl; c.ok4 = inside && hh_ok && wh;
    c.wgt = inside ? w : 0.f;
    const int rl = start + max(h_low, 0) * W, rh = start + min(h_low + 1, H - 1) * W;
    const int cl = max(w_low, 0), ch = min(w_low + 1, W - 1);
    const int row = M * kD;
    c.o1 = (rl + cl) * row + head_chan; c.o2 = (rl + ch) * row + head_chan;
    c.o3 = (rh + cl) * row + head_chan; c.o4 = (rh + ch) * row + head_chan;
    return c;
}

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
// One sample: 4 corner rows of this lane's 8 channels, folded into acc.
template <typename VT>
__device__ __forceinline__ void fwd_sample(const VT *__restrict__ vimg, float x, float y, float w, int H, int W,
                                           int start, int M, int head_chan, float (&acc)[8])
{
    const Corner4 c = setup_sample(x, y, w, H, W, start, M, head_chan);
    const float a_h = c.hh * c.wgt, b_h = c.lh * c.wgt;
    const float w1 = c.ok1 ? a_h * c.hw : 0.f, w2 = c.ok2 ? a_h * c.lw : 0.f;
    const float w3 = c.ok3 ? b_h * c.hw : 0.f, w4 = c.ok4 ? b_h * c.lw : 0.f;
    // issue the four corner loads back to back, then fold them in arrival order
    typename Vec8<VT>::raw r1 = Vec8<VT>::load_raw(vimg + c.o1), r2 = Vec8<VT>::load_raw(vimg + c.o2);
    typename Vec8<VT>::raw r3 = Vec8<VT>::load_raw(vimg + c.o3), r4 = Vec8<VT>::load_raw(vimg + c.o4);
    Vec8<VT>::fma(w1, r1, acc);
    Vec8<VT>::fma(w2, r2, acc);
    Vec8<VT>::fma(w3, r3, acc);
    Vec8<VT>::fma(w4, r4, acc);
}

// rotate the quad's per-lane sample data by one lane: lane j takes lane j+1's registers, so that
// after l rotations quad lane 0 holds the samples of level l
__device__ __forceinline__ void quad_rotate(float4 &v)
{
    constexpr int R = MSDA_QUAD_PERM(1, 2, 3, 0);
    v.x = dpp_quad<R>(v.x); v.y = dpp_quad<R>(v.y); v.z = dpp_quad<R>(v.z); v.w = dpp_quad<R>(v.w);
}

// WAVES = occupancy target (waves per SIMD) handed to the register allocator; FENCE = number of
// samples the instruction scheduler may interleave (it otherwise hoists every gather of a level
// to the top and spills): a scheduling barrier closes each group of FENCE samples.
template <typename VT, int WAVES, int FENCE>
__global__ __launch_bounds__(kBlock, WAVES) void quad_forward_kernel(
    const VT *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ starts,
    const float *__restrict__ loc, const float *__restrict__ aw, int total_qm, int S, int M, int Lq,
    VT *__restrict__ out, int dbg)
{
    const int t = blockIdx.x * kBlock + threadIdx.x;
    int qm = t >> 2;
    const int sub = t & 3;
    const bool live = qm < total_qm;
    qm = live ? qm : total_qm - 1;   // keep whole quads converged for the DPP broadcasts
    const int m = qm % M;
    const int n = (qm / M) / Lq;
    const VT *vimg = value + (long)n * S * M * kD;
    const int head_chan = m * kD + sub * 8;
    if (dbg & 1) M = 0;          // ablation: every gather hits the head's first rows (cache-resident)
    // quad lane j loads the 4 points of level j: (x,y) x 4 and 4 weights
    const float4 *loc4 = reinterpret_cast<const float4 *>(loc) + (long)qm * 8 + sub * 2;
    float4 la = loc4[0], lb = loc4[1];
    float4 wa = reinterpret_cast<const float4 *>(aw)[(long)qm * 4 + sub];
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int l = 0; l < kL; ++l) {
        const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1], start = (int)starts[l];
        fwd_sample<VT>(vimg, quad_bcast<0>(la.x), quad_bcast<0>(la.y), quad_bcast<0>(wa.x), H, W, start, M, head_chan, acc);
        if (FENCE == 1) __builtin_amdgcn_sched_barrier(0);
        fwd_sample<VT>(vimg, quad_bcast<0>(la.z), quad_bcast<0>(la.w), quad_bcast<0>(wa.y), H, W, start, M, head_chan, acc);
        if (FENCE <= 2) __builtin_amdgcn_sched_barrier(0);
        fwd_sample<VT>(vimg, quad_bcast<0>(lb.x), quad_bcast<0>(lb.y), quad_bcast<0>(wa.z), H, W, start, M, head_chan, acc);
        if (FENCE == 1) __builtin_amdgcn_sched_barrier(0);
        fwd_sample<VT>(vimg, quad_bcast<0>(lb.z), quad_bcast<0>(lb.w), quad_bcast<0>(wa.w), H, W, start, M, head_chan, acc);
        quad_rotate(la); quad_rotate(lb); quad_rotate(wa);
    }
    if (live) Vec8<VT>::store(out + (long)qm * kD + sub * 8, acc);
}

// ------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float dot8(const float (&a)[8], const float (&b)[8])
{
    float s = a[0] * b[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) s = fmaf(a[k], b[k], s);
    return s;
}

__device__ __forceinline__ void scatter8(float *__restrict__ g, float w, const float (&tg)[8])
{
#pragma unroll
    for (int k = 0; k < 8; ++k) atomic_add(g + k, w * tg[k]);
}

// One sample of the backward pass.  Returns (d out / d attn_weight, d out / d x, d out / d y)
// contracted with grad_out, identical in all four lanes of the quad.
template <typename VT, bool SCATTER>
__device__ __forceinline__ void bwd_sample(const VT *__restrict__ vimg, float *__restrict__ gimg, float x, float y,
                                           float w, int H, int W, int start, int M, int head_chan, bool live,
                                           const float (&tg)[8], float &g_a, float &g_w, float &g_h)
{
    const Corner4 c = setup_sample(x, y, w, H, W, start, M, head_chan);
    float v1[8], v2[8], v3[8], v4[8];
    Vec8<VT>::load(vimg + c.o1, v1);
    Vec8<VT>::load(vimg + c.o2, v2);
    Vec8<VT>::load(vimg + c.o3, v3);
    Vec8<VT>::load(vimg + c.o4, v4);
    // grad_value: w_k * attn * grad_out, 8 channels per corner per lane
    const float a_h = c.hh * c.wgt, b_h = c.lh * c.wgt;
    if (SCATTER && live) {
        if (c.ok1) scatter8(gimg + c.o1, a_h * c.hw, tg);
        if (c.ok2) scatter8(gimg + c.o2, a_h * c.lw, tg);
        if (c.ok3) scatter8(gimg + c.o3, b_h * c.hw, tg);
        if (c.ok4) scatter8(gimg + c.o4, b_h * c.lw, tg);
    }
    // channel reductions: 8 channels in-lane, then across the quad with DPP
    const float e1 = quad_sum(c.ok1 ? dot8(tg, v1) : 0.f);
    const float e2 = quad_sum(c.ok2 ? dot8(tg, v2) : 0.f);
    const float e3 = quad_sum(c.ok3 ? dot8(tg, v3) : 0.f);
    const float e4 = quad_sum(c.ok4 ? dot8(tg, v4) : 0.f);
    g_a = c.hh * (c.hw * e1 + c.lw * e2) + c.lh * (c.hw * e3 + c.lw * e4);
    g_w = (float)W * c.wgt * (c.hh * (e2 - e1) + c.lh * (e4 - e3));
    g_h = (float)H * c.wgt * (c.hw * (e3 - e1) + c.lw * (e4 - e2));
}

// SCATTER = false: only grad_sampling_loc / grad_attn_weight ("K1"; grad_value is then produced by
// the sorted scatter of msda_window.hip).
template <typename VT, int WAVES, bool SCATTER>
__global__ __launch_bounds__(kBlock, WAVES) void quad_backward_kernel(
    const VT *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ starts,
    const float *__restrict__ loc, const float *__restrict__ aw, const VT *__restrict__ grad_out, int total_qm,
    int S, int M, int Lq, float *__restrict__ g_value, float *__restrict__ g_loc, float *__restrict__ g_aw)
{
    const int t = blockIdx.x * kBlock + threadIdx.x;
    int qm = t >> 2;
    const int sub = t & 3;
    const bool live = qm < total_qm;
    qm = live ? qm : total_qm - 1;
    const int m = qm % M;
    const int n = (qm / M) / Lq;
    const long img = (long)n * S * M * kD;
    const VT *vimg = value + img;
    float *gimg = g_value + img;
    const int head_chan = m * kD + sub * 8;
    const float4 *loc4 = reinterpret_cast<const float4 *>(loc) + (long)qm * 8 + sub * 2;
    float4 la = loc4[0], lb = loc4[1];
    float4 wa = reinterpret_cast<const float4 *>(aw)[(long)qm * 4 + sub];
    float tg[8];
    Vec8<VT>::load(grad_out + (long)qm * kD + sub * 8, tg);
    float4 gla = make_float4(0.f, 0.f, 0.f, 0.f), glb = gla, ga = gla;
#pragma unroll 1
    for (int l = 0; l < kL; ++l) {
        const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1], start = (int)starts[l];
        float4 ra, rb, rw;   // this level's results: (gx,gy) x 4 points, g_aw x 4 points
        bwd_sample<VT, SCATTER>(vimg, gimg, quad_bcast<0>(la.x), quad_bcast<0>(la.y), quad_bcast<0>(wa.x), H, W, start, M,
                       head_chan, live, tg, rw.x, ra.x, ra.y);
        __builtin_amdgcn_sched_barrier(0);
        bwd_sample<VT, SCATTER>(vimg, gimg, quad_bcast<0>(la.z), quad_bcast<0>(la.w), quad_bcast<0>(wa.y), H, W, start, M,
                       head_chan, live, tg, rw.y, ra.z, ra.w);
        __builtin_amdgcn_sched_barrier(0);
        bwd_sample<VT, SCATTER>(vimg, gimg, quad_bcast<0>(lb.x), quad_bcast<0>(lb.y), quad_bcast<0>(wa.z), H, W, start, M,
                       head_chan, live, tg, rw.z, rb.x, rb.y);
        __builtin_amdgcn_sched_barrier(0);
        bwd_sample<VT, SCATTER>(vimg, gimg, quad_bcast<0>(lb.z), quad_bcast<0>(lb.w), quad_bcast<0>(wa.w), H, W, start, M,
                       head_chan, live, tg, rw.w, rb.z, rb.w);
        if (sub == l) { gla = ra; glb = rb; ga = rw; }   // quad lane l owns level l's outputs
        quad_rotate(la); quad_rotate(lb); quad_rotate(wa);
    }
    if (live) {
        float4 *gl4 = reinterpret_cast<float4 *>(g_loc) + (long)qm * 8 + sub * 2;
        gl4[0] = gla;
        gl4[1] = glb;
        reinterpret_cast<float4 *>(g_aw)[(long)qm * 4 + sub] = ga;
    }
}

}  // namespace

bool quad_supports(const Problem &p)
{
    if (p.dtype != MSDA_F32 && p.dtype != MSDA_BF16) return false;
    if (p.D != kD || p.L != kL || p.P != kP) return false;
    const long total_qm = (long)p.N * p.Lq * p.M;
    if (total_qm * 4 >= (1L << 31)) return false;
    if ((long)p.S * p.M * kD >= (1L << 31)) return false;   // per-image element offsets are 32-bit
    return true;
}

void launch_quad_forward(const Problem &p)
{
    const char *e = getenv("RLIPV2_MSDA_DEBUG");       // ablation switch, profiling only
    const int dbg = e ? atoi(e) : 0;
    const int total_qm = p.N * p.Lq * p.M;
    const int grid = (int)(((long)total_qm * 4 + kBlock - 1) / kBlock);
    if (p.dtype == MSDA_F32)
        hipLaunchKernelGGL((quad_forward_kernel<float, 6, 2>), dim3(grid), dim3(kBlock), 0, p.stream,
                           (const float *)p.value, p.shapes, p.starts, (const float *)p.loc, (const float *)p.aw,
                           total_qm, p.S, p.M, p.Lq, (float *)p.out, dbg);
    else
        hipLaunchKernelGGL((quad_forward_kernel<bf16_t, 6, 2>), dim3(grid), dim3(kBlock), 0, p.stream,
                           (const bf16_t *)p.value, p.shapes, p.starts, (const float *)p.loc, (const float *)p.aw,
                           total_qm, p.S, p.M, p.Lq, (bf16_t *)p.out, dbg);
}

void launch_quad_backward(const Problem &p)
{
    const int total_qm = p.N * p.Lq * p.M;
    const int grid = (int)(((long)total_qm * 4 + kBlock - 1) / kBlock);
    if (p.dtype == MSDA_F32)
        hipLaunchKernelGGL((quad_backward_kernel<float, 3, true>), dim3(grid), dim3(kBlock), 0, p.stream,
                           (const float *)p.value, p.shapes, p.starts, (const float *)p.loc, (const float *)p.aw,
                           (const float *)p.grad_out, total_qm, p.S, p.M, p.Lq, (float *)p.g_value,
                           (float *)p.g_loc, (float *)p.g_aw);
    else
        hipLaunchKernelGGL((quad_backward_kernel<bf16_t, 4, true>), dim3(grid), dim3(kBlock), 0, p.stream,
                           (const bf16_t *)p.value, p.shapes, p.starts, (const float *)p.loc, (const float *)p.aw,
                           (const bf16_t *)p.grad_out, total_qm, p.S, p.M, p.Lq, (float *)p.g_value,
                           (float *)p.g_loc, (float *)p.g_aw);
}

void launch_quad_backward_reduce(const Problem &p)
{
    const int total_qm = p.N * p.Lq * p.M;
    const int grid = (int)(((long)total_qm * 4 + kBlock - 1) / kBlock);
    if (p.dtype == MSDA_F32)
        hipLaunchKernelGGL((quad_backward_kernel<float, 4, false>), dim3(grid), dim3(kBlock), 0, p.stream,
                           (const float *)p.value, p.shapes, p.starts, (const float *)p.loc, (const float *)p.aw,
                           (const float *)p.grad_out, total_qm, p.S, p.M, p.Lq, (float *)p.g_value,
                           (float *)p.g_loc, (float *)p.g_aw);
    else
        hipLaunchKernelGGL((quad_backward_kernel<bf16_t, 4, false>), dim3(grid), dim3(kBlock), 0, p.stream,
                           (const bf16_t *)p.value, p.shapes, p.starts, (const float *)p.loc, (const float *)p.aw,
                           (const bf16_t *)p.grad_out, total_qm, p.S, p.M, p.Lq, (float *)p.g_value,
                           (float *)p.g_loc, (float *)p.g_aw);
}

}  // namespace msda
