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
#include "msda_geometry.h"
#include "msda_internal.h"

namespace msda {

namespace {

constexpr int kBlock = 256;
constexpr int kL = 4, kP = 4, kD = 32;

// ---- sample geometry --------------------------------------------------------------------------------
// Addressing is done with buffer loads: a 32-bit byte offset from the tensor base in a wave-uniform
// resource descriptor (no 64-bit VALU address arithmetic), 24-bit integer multiplies, and the
// hardware's range check -- a corner outside the level gets the offset kOob, which is beyond
// num_records and therefore loads zeros.  So there is no clamping and no per-corner weight
// masking: weight * 0 is the zero padding of the reference (ms_deform_im2col_cuda.cuh:58-78).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kOob = 0xFFFFFF00u;

struct Corners {
    unsigned o00, o01, o10, o11;    // byte offsets (or kOob) of (y0,x0) (y0,x1) (y1,x0) (y1,x1)
    float hh, hw, lh, lw;           // bilinear fractions
    bool v00, v01, v10, v11;        // corner inside the level
};

// (x, y) normalised -> corners of level (H, W, start).  `lane_byte` = image base + head/channel offset.
// Pixel coordinates are clamped to [-2, size + 1] first: every out-of-range (or NaN / Inf) location
// then has all its corners out of the level, with finite fractions.
template <int ELEM>
__device__ __forceinline__ Corners corners_of(float x, float y, int H, int W, int start, int row_bytes,
                                               unsigned lane_byte)
{
    Corners c;
    const float Hf = (float)H, Wf = (float)W;
    float h = fmaf(y, Hf, -0.5f), w = fmaf(x, Wf, -0.5f);
    h = fminf(fmaxf(h, -2.f), Hf + 1.f);
    w = fminf(fmaxf(w, -2.f), Wf + 1.f);
    const float hf = floorf(h), wf = floorf(w);
    c.lh = h - hf; c.lw = w - wf; c.hh = 1.f - c.lh; c.hw = 1.f - c.lw;
    const int ih = (int)hf, iw = (int)wf;
    const bool y0 = (unsigned)ih < (unsigned)H, y1 = (unsigned)(ih + 1) < (unsigned)H;
    const bool x0 = (unsigned)iw < (unsigned)W, x1 = (unsigned)(iw + 1) < (unsigned)W;
    c.v00 = y0 && x0; c.v01 = y0 && x1; c.v10 = y1 && x0; c.v11 = y1 && x1;
    const int pix = start + __mul24(ih, W) + iw;                       // may be off-level: masked below
    const unsigned o = (unsigned)__mul24(pix, row_bytes) + lane_byte;
    const unsigned down = (unsigned)__mul24(W, row_bytes);
    c.o00 = c.v00 ? o : kOob;
    c.o01 = c.v01 ? o + row_bytes : kOob;
    c.o10 = c.v10 ? o + down : kOob;
    c.o11 = c.v11 ? o + down + row_bytes : kOob;
    return c;
}

// XCD-aware block order: hardware block b runs on XCD b % 8 (observed, MI355X_MICROARCH.md); give every
// XCD one contiguous eighth of the (image, query) range so that its private L2 fetches only the value
// rows that eighth samples.  Measured with the round-robin order: FETCH_SIZE 379 MB per batch-4 encoder
// forward against 182 MB of algorithmic reads -- every XCD pulled the whole value tensor through its own L2.
// Bijective for any grid size; placement only affects speed.
__device__ __forceinline__ int xcd_block_id()
{
    const int nb = gridDim.x, b = blockIdx.x;
    const int q = nb >> 3, r = nb & 7, xcd = b & 7, idx = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// 8 channels of one corner through the buffer descriptor
template <typename VT> struct Corner8;
template <> struct Corner8<float> {
    struct raw { u32x4 a, b; };
    static __device__ __forceinline__ raw load(__amdgpu_buffer_rsrc_t r, unsigned off)
    {
        raw v;
        v.a = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
        v.b = __builtin_amdgcn_raw_buffer_load_b128(r, off + 16, 0, 0);   // kOob + 16 is still out of range
        return v;
    }
    static __device__ __forceinline__ void unpack(const raw &v, float (&f)[8])
    {
        f[0] = __uint_as_float(v.a.x); f[1] = __uint_as_float(v.a.y); f[2] = __uint_as_float(v.a.z);
        f[3] = __uint_as_float(v.a.w); f[4] = __uint_as_float(v.b.x); f[5] = __uint_as_float(v.b.y);
        f[6] = __uint_as_float(v.b.z); f[7] = __uint_as_float(v.b.w);
    }
};
template <> struct Corner8<bf16_t> {
    typedef u32x4 raw;
    static __device__ __forceinline__ raw load(__amdgpu_buffer_rsrc_t r, unsigned off)
    {
        return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    }
    static __device__ __forceinline__ void unpack(const raw &v, float (&f)[8])
    {
        f[0] = bf16_lo(v.x); f[1] = bf16_hi(v.x); f[2] = bf16_lo(v.y); f[3] = bf16_hi(v.y);
        f[4] = bf16_lo(v.z); f[5] = bf16_hi(v.z); f[6] = bf16_lo(v.w); f[7] = bf16_hi(v.w);
    }
};

template <typename VT>
__device__ __forceinline__ void fma8(float w, const typename Corner8<VT>::raw &r, float (&acc)[8])
{
    float f[8];
    Corner8<VT>::unpack(r, f);
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = fmaf(w, f[k], acc[k]);
}

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
// One sample: 4 corner rows of this lane's 8 channels, folded into acc.  The reference's sample-level
// inclusion test (.cuh:285) is implied: a location outside (-1, size) has no corner inside the level.
template <typename VT>
__device__ __forceinline__ void fwd_sample(__amdgpu_buffer_rsrc_t vr, float x, float y, float w, int H, int W,
                                           int start, int row_bytes, unsigned lane_byte, float (&acc)[8])
{
    const Corners c = corners_of<sizeof(VT)>(x, y, H, W, start, row_bytes, lane_byte);
    // issue the four corner loads back to back, then fold them in arrival order
    const typename Corner8<VT>::raw r00 = Corner8<VT>::load(vr, c.o00), r01 = Corner8<VT>::load(vr, c.o01);
    const typename Corner8<VT>::raw r10 = Corner8<VT>::load(vr, c.o10), r11 = Corner8<VT>::load(vr, c.o11);
    const float a = c.hh * w, b = c.lh * w;
    fma8<VT>(a * c.hw, r00, acc);
    fma8<VT>(a * c.lw, r01, acc);
    fma8<VT>(b * c.hw, r10, acc);
    fma8<VT>(b * c.lw, r11, acc);
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
    unsigned value_bytes, VT *__restrict__ out, int dbg)
{
    const int t = xcd_block_id() * kBlock + threadIdx.x;
    int qm = t >> 2;
    const int sub = t & 3;
    const bool live = qm < total_qm;
    qm = live ? qm : total_qm - 1;   // keep whole quads converged for the DPP broadcasts
    const int m = qm % M;
    const int n = (qm / M) / Lq;
    int row_bytes = M * kD * (int)sizeof(VT);
    const unsigned lane_byte = (unsigned)n * (unsigned)S * (unsigned)row_bytes
                               + (unsigned)(m * kD + sub * 8) * (unsigned)sizeof(VT);
    if (MSDA_DBG(dbg) & 1) row_bytes = 0;      // ablation: every gather hits the head's first rows (cache-resident)
    const __amdgpu_buffer_rsrc_t vr = __builtin_amdgcn_make_buffer_rsrc((void *)value, 0, value_bytes, 0x00020000);
    // quad lane j loads the 4 points of level j: (x,y) x 4 and 4 weights
    const float4 *loc4 = reinterpret_cast<const float4 *>(loc) + (long)qm * 8 + sub * 2;
    float4 la = loc4[0], lb = loc4[1];
    float4 wa = reinterpret_cast<const float4 *>(aw)[(long)qm * 4 + sub];
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int l = 0; l < kL; ++l) {
        const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1], start = (int)starts[l];
        fwd_sample<VT>(vr, quad_bcast<0>(la.x), quad_bcast<0>(la.y), quad_bcast<0>(wa.x), H, W, start, row_bytes, lane_byte, acc);
        if (FENCE == 1) __builtin_amdgcn_sched_barrier(0);
        fwd_sample<VT>(vr, quad_bcast<0>(la.z), quad_bcast<0>(la.w), quad_bcast<0>(wa.y), H, W, start, row_bytes, lane_byte, acc);
        if (FENCE <= 2) __builtin_amdgcn_sched_barrier(0);
        fwd_sample<VT>(vr, quad_bcast<0>(lb.x), quad_bcast<0>(lb.y), quad_bcast<0>(wa.z), H, W, start, row_bytes, lane_byte, acc);
        if (FENCE == 1) __builtin_amdgcn_sched_barrier(0);
        fwd_sample<VT>(vr, quad_bcast<0>(lb.z), quad_bcast<0>(lb.w), quad_bcast<0>(wa.w), H, W, start, row_bytes, lane_byte, acc);
        quad_rotate(la); quad_rotate(lb); quad_rotate(wa);
    }
    if (live) Vec8<VT>::store(out + (long)qm * kD + sub * 8, acc);
}

// The direct-gather forward with the module's sampling geometry as its prologue (ms_deform_attn.py:101-112 +
// :116-117 in one launch): reads the raw projection row [offsets | logits] (48 values per head, in value's
// dtype) and the reference points instead of float32 sampling_loc / attn_weight -- 96 B instead of 192 B per
// (query, head) for bfloat16 -- and, when a backward pass will follow (SAVE), writes the float32 locations /
// weights it computed for that pass.  Quad lane j = level j in both the geometry and the gather loop.
template <typename VT, int REFDIM, bool SAVE, int WAVES>
__global__ __launch_bounds__(kBlock, WAVES) void quad_forward_fused_kernel(
    const VT *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ starts,
    const VT *__restrict__ qproj, const float *__restrict__ ref, int total_qm, int S, int M, int Lq,
    unsigned value_bytes, VT *__restrict__ out, float *__restrict__ loc_save, float *__restrict__ aw_save)
{
    const int t = xcd_block_id() * kBlock + threadIdx.x;
    int qm = t >> 2;
    const int sub = t & 3;
    const bool live = qm < total_qm;
    qm = live ? qm : total_qm - 1;   // keep whole quads converged for the DPP broadcasts
    const int m = qm % M;
    const int row = qm / M;          // n * Lq + q
    const int n = row / Lq;
    const int row_bytes = M * kD * (int)sizeof(VT);
    const unsigned lane_byte = (unsigned)n * (unsigned)S * (unsigned)row_bytes
                               + (unsigned)(m * kD + sub * 8) * (unsigned)sizeof(VT);
    const __amdgpu_buffer_rsrc_t vr = __builtin_amdgcn_make_buffer_rsrc((void *)value, 0, value_bytes, 0x00020000);
    float o[8], w[4];
    geom::forward<VT, REFDIM>(qproj + (long)row * (M * 48), ref + (long)row * (kL * REFDIM), shapes, m, M, sub, o, w);
    float4 la = make_float4(o[0], o[1], o[2], o[3]), lb = make_float4(o[4], o[5], o[6], o[7]);
    float4 wa = make_float4(w[0], w[1], w[2], w[3]);
    if (SAVE && live) {
        float4 *loc4 = reinterpret_cast<float4 *>(loc_save) + (long)qm * 8 + sub * 2;
        loc4[0] = la;
        loc4[1] = lb;
        reinterpret_cast<float4 *>(aw_save)[(long)qm * 4 + sub] = wa;
    }
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int l = 0; l < kL; ++l) {
        const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1], start = (int)starts[l];
        fwd_sample<VT>(vr, quad_bcast<0>(la.x), quad_bcast<0>(la.y), quad_bcast<0>(wa.x), H, W, start, row_bytes, lane_byte, acc);
        fwd_sample<VT>(vr, quad_bcast<0>(la.z), quad_bcast<0>(la.w), quad_bcast<0>(wa.y), H, W, start, row_bytes, lane_byte, acc);
        __builtin_amdgcn_sched_barrier(0);
        fwd_sample<VT>(vr, quad_bcast<0>(lb.x), quad_bcast<0>(lb.y), quad_bcast<0>(wa.z), H, W, start, row_bytes, lane_byte, acc);
        fwd_sample<VT>(vr, quad_bcast<0>(lb.z), quad_bcast<0>(lb.w), quad_bcast<0>(wa.w), H, W, start, row_bytes, lane_byte, acc);
        quad_rotate(la); quad_rotate(lb); quad_rotate(wa);
    }
    if (live) Vec8<VT>::store(out + (long)qm * kD + sub * 8, acc);
}

// ------------------------------------------------------------------------------------------
// forward with the coarse levels resident in LDS (bfloat16, many queries: the encoder's self-attention)
// ------------------------------------------------------------------------------------------
// The direct-gather kernels are bound by the texture path: 64 corner rows of 64 B per (query, head) = 2.9 GB per
// batch-4 encoder call at ~64 B/clk/CU.  Half of those gathers hit the two coarsest levels, which are tiny: 1 050 + 273
// pixels x 64 B = 85 KB per (image, head) at 800x1333.  Here a workgroup of 1 024 threads is bound to ONE (image, head)
// for its whole life (256 / (N*M) workgroups per pair, each a contiguous range of queries), copies that pair's rows of
// the trailing levels into LDS once, and samples them with ds_read_b128 (4x the texture path's bytes per clock);
// levels 0 / 1 still go through the buffer loads.  Which levels are staged is decided at run time from the
// device-resident shapes: the longest suffix of levels whose rows fit kCoarseBytes (none: everything is gathered).
// A quad = one query (same head for the whole workgroup), quad lane j = level j of the geometry, as everywhere.
constexpr int kCoarseBytes = 96 * 1024;
constexpr int kCoarseBlock = 1024;
constexpr int kCoarseZero = 128;             // bytes of zeros in front of the staged rows: where out-of-level corners read

// one sample from the LDS-resident rows; `stage_pix0` = first staged pixel of the image, rows of 64 B (bfloat16)
__device__ __forceinline__ void lds_sample_bf16(const unsigned char *rows, int stage_pix0, float x, float y, float w, int H,
                                                int W, int start, int sub, float (&acc)[8])
{
    const float Hf = (float)H, Wf = (float)W;
    float h = fmaf(y, Hf, -0.5f), v = fmaf(x, Wf, -0.5f);
    h = fminf(fmaxf(h, -2.f), Hf + 1.f);
    v = fminf(fmaxf(v, -2.f), Wf + 1.f);
    const float hf = floorf(h), wf = floorf(v);
    const float lh = h - hf, lw = v - wf, hh = 1.f - lh, hw = 1.f - lw;
    const int ih = (int)hf, iw = (int)wf;
    const bool y0 = (unsigned)ih < (unsigned)H, y1 = (unsigned)(ih + 1) < (unsigned)H;
    const bool x0 = (unsigned)iw < (unsigned)W, x1 = (unsigned)(iw + 1) < (unsigned)W;
    const int lane = sub * 16;
    const int base = kCoarseZero + (start - stage_pix0 + __mul24(ih, W) + iw) * 64 + lane;
    const int a00 = (y0 && x0) ? base : lane;                       // invalid corner -> the zero slot
    const int a01 = (y0 && x1) ? base + 64 : lane;
    const int a10 = (y1 && x0) ? base + W * 64 : lane;
    const int a11 = (y1 && x1) ? base + W * 64 + 64 : lane;
    const uint4 r00 = *reinterpret_cast<const uint4 *>(rows + a00), r01 = *reinterpret_cast<const uint4 *>(rows + a01);
    const uint4 r10 = *reinterpret_cast<const uint4 *>(rows + a10), r11 = *reinterpret_cast<const uint4 *>(rows + a11);
    const float a = hh * w, b = lh * w;
    Vec8<bf16_t>::fma(a * hw, r00, acc);
    Vec8<bf16_t>::fma(a * lw, r01, acc);
    Vec8<bf16_t>::fma(b * hw, r10, acc);
    Vec8<bf16_t>::fma(b * lw, r11, acc);
}

// number of trailing levels whose rows fit the LDS budget (0..kL) and the first staged pixel
__device__ __forceinline__ int coarse_levels(const int64_t *__restrict__ shapes, const int64_t *__restrict__ starts, int S,
                                             int &pix0)
{
    int n = 0;
    pix0 = S;
#pragma unroll
    for (int l = kL - 1; l >= 0; --l) {
        const int st = (int)starts[l];
        if (n == kL - 1 - l && (S - st) * 64 + kCoarseZero <= kCoarseBytes) { n = kL - l; pix0 = st; }
    }
    return n;
}

// REFDIM = 0: float32 sampling_loc / attn_weight operands (the B0 signature); 2 / 4: the module's geometry as the
// prologue (raw projection rows + reference points), SAVE: write the float32 locations / weights for the backward
template <int REFDIM, bool SAVE, int BLOCK = kCoarseBlock>
__global__ __launch_bounds__(BLOCK, 4) void quad_forward_coarse_kernel(
    const bf16_t *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ starts,
    const float *__restrict__ loc, const float *__restrict__ aw, const bf16_t *__restrict__ qproj,
    const float *__restrict__ ref, int N, int S, int M, int Lq, int wgs_per_pair, unsigned value_bytes,
    bf16_t *__restrict__ out, float *__restrict__ loc_save, float *__restrict__ aw_save, int max_staged)
{
    MSDA_DYNAMIC_LDS(unsigned char, coarse_lds);
    const int tid = threadIdx.x;
    // XCD-aware placement (hardware block b runs on XCD b % 8): all workgroups of one (image, head) pair on ONE XCD,
    // so that an XCD's L2 holds the fine-level rows of pairs / 8 pairs only.  Bijective for any grid; speed only.
    int pair, part;
    {
        const int pairs = N * M, b = blockIdx.x;
        if ((pairs & 15) == 0 && (M & 1) == 0) {
            // ... and the two heads sharing a 128-byte line of a value row ([S, M, 32] bfloat16: 64 B per head) on the
            // SAME XCD, so that every line an L2 fetches is used whole
            const int idx = b >> 3, k = idx / wgs_per_pair;
            const int g = (b & 7) + 8 * (k >> 1);                 // (image, head pair) group
            pair = (g / (M >> 1)) * M + 2 * (g % (M >> 1)) + (k & 1);
            part = idx % wgs_per_pair;
        } else if ((pairs & 7) == 0) {
            const int idx = b >> 3;
            pair = (b & 7) + 8 * (idx / wgs_per_pair);
            part = idx % wgs_per_pair;
        } else {
            pair = b / wgs_per_pair;
            part = b % wgs_per_pair;
        }
    }
    const int n = pair / M, m = pair % M;
    const int row_bytes = M * kD * 2;
    int pix0;
    int staged = coarse_levels(shapes, starts, S, pix0);             // levels kL - staged .. kL - 1 are in LDS
    if (MSDA_DBG(1) && staged > max_staged) {                        // ablation builds: fewer (or no) staged levels
        staged = max_staged;
        pix0 = staged > 0 ? (int)starts[kL - staged] : S;
    }
    if (tid < kCoarseZero / 4) reinterpret_cast<int *>(coarse_lds)[tid] = 0;
    {
        const int pieces = (S - pix0) * 4;                            // 16-byte pieces of this head's 64-byte rows
        const unsigned char *src = reinterpret_cast<const unsigned char *>(value)
                                   + ((size_t)n * S + pix0) * row_bytes + (size_t)m * 64;
        for (int i = tid; i < pieces; i += BLOCK)
            *reinterpret_cast<uint4 *>(coarse_lds + kCoarseZero + i * 16) =
                *reinterpret_cast<const uint4 *>(src + (size_t)(i >> 2) * row_bytes + (i & 3) * 16);
    }
    __syncthreads();
    const unsigned lane_base = (unsigned)n * (unsigned)S * (unsigned)row_bytes + (unsigned)(m * kD) * 2u;
    const __amdgpu_buffer_rsrc_t vr = __builtin_amdgcn_make_buffer_rsrc((void *)value, 0, value_bytes, 0x00020000);
    const int per = (Lq + wgs_per_pair - 1) / wgs_per_pair;
    const int q_lo = part * per, q_hi = min(Lq, q_lo + per);
    const int sub = tid & 3;
    const unsigned lane_byte = lane_base + (unsigned)(sub * 16);
    // (B0-signature operands: the next iteration's locations / weights travel while this one samples -- the 16 waves of
    //  the workgroup start together and stay roughly in phase, nothing else would hide that latency)
    float4 nla = make_float4(0.f, 0.f, 0.f, 0.f), nlb = nla, nwa = nla;
    auto fetch = [&](int q0_) {
        const int q_ = min(q0_ + (tid >> 2), q_hi - 1);
        const long qm_ = ((long)n * Lq + q_) * M + m;
        const float4 *loc4 = reinterpret_cast<const float4 *>(loc) + qm_ * 8 + sub * 2;
        nla = loc4[0]; nlb = loc4[1];
        nwa = reinterpret_cast<const float4 *>(aw)[qm_ * 4 + sub];
    };
    if (REFDIM == 0 && q_lo < q_hi) fetch(q_lo);
    for (int q0 = q_lo; q0 < q_hi; q0 += BLOCK / 4) {
        int q = q0 + (tid >> 2);
        const bool live = q < q_hi;
        q = live ? q : q_hi - 1;                                      // keep whole quads converged for the DPP broadcasts
        const long rowi = (long)n * Lq + q;                           // (image, query)
        const long qm = rowi * M + m;
        float4 la, lb, wa;
        if (REFDIM == 0) {
            la = nla; lb = nlb; wa = nwa;
            if (q0 + BLOCK / 4 < q_hi) fetch(q0 + BLOCK / 4);
        } else {
            constexpr int RD = REFDIM == 0 ? 2 : REFDIM;
            float o[8], w[4];
            geom::forward<bf16_t, RD>(qproj + rowi * (M * 48), ref + rowi * (kL * RD), shapes, m, M, sub, o, w);
            la = make_float4(o[0], o[1], o[2], o[3]); lb = make_float4(o[4], o[5], o[6], o[7]);
            wa = make_float4(w[0], w[1], w[2], w[3]);
            if (SAVE && live) {
                float4 *loc4 = reinterpret_cast<float4 *>(loc_save) + qm * 8 + sub * 2;
                loc4[0] = la;
                loc4[1] = lb;
                reinterpret_cast<float4 *>(aw_save)[qm * 4 + sub] = wa;
            }
        }
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        // (two loops, one code path each: a single loop with a branch keeps both paths' registers alive and spills)
#pragma unroll 1
        for (int l = 0; l < kL - staged; ++l) {
            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1], start = (int)starts[l];
            fwd_sample<bf16_t>(vr, quad_bcast<0>(la.x), quad_bcast<0>(la.y), quad_bcast<0>(wa.x), H, W, start, row_bytes, lane_byte, acc);
            fwd_sample<bf16_t>(vr, quad_bcast<0>(la.z), quad_bcast<0>(la.w), quad_bcast<0>(wa.y), H, W, start, row_bytes, lane_byte, acc);
            __builtin_amdgcn_sched_barrier(0);
            fwd_sample<bf16_t>(vr, quad_bcast<0>(lb.x), quad_bcast<0>(lb.y), quad_bcast<0>(wa.z), H, W, start, row_bytes, lane_byte, acc);
            fwd_sample<bf16_t>(vr, quad_bcast<0>(lb.z), quad_bcast<0>(lb.w), quad_bcast<0>(wa.w), H, W, start, row_bytes, lane_byte, acc);
            quad_rotate(la); quad_rotate(lb); quad_rotate(wa);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
        for (int l = kL - staged; l < kL; ++l) {
            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1], start = (int)starts[l];
            lds_sample_bf16(coarse_lds, pix0, quad_bcast<0>(la.x), quad_bcast<0>(la.y), quad_bcast<0>(wa.x), H, W, start, sub, acc);
            lds_sample_bf16(coarse_lds, pix0, quad_bcast<0>(la.z), quad_bcast<0>(la.w), quad_bcast<0>(wa.y), H, W, start, sub, acc);
            __builtin_amdgcn_sched_barrier(0);
            lds_sample_bf16(coarse_lds, pix0, quad_bcast<0>(lb.x), quad_bcast<0>(lb.y), quad_bcast<0>(wa.z), H, W, start, sub, acc);
            lds_sample_bf16(coarse_lds, pix0, quad_bcast<0>(lb.z), quad_bcast<0>(lb.w), quad_bcast<0>(wa.w), H, W, start, sub, acc);
            quad_rotate(la); quad_rotate(lb); quad_rotate(wa);
        }
        if (live) Vec8<bf16_t>::store(out + qm * kD + sub * 8, acc);
    }
}

// Geometry-sharing variant of the direct-gather forward.  In the kernel above every lane of a quad repeats
// the same ~45 instructions of sample geometry (pixel coordinates, bounds, corner offsets, bilinear x
// attention weights) for each of the 16 samples, although only the 8 channels differ between the lanes.
// Here quad lane p owns POINT p of every level: it computes that sample's geometry once, and the quad
// fetches the 4 corner offsets + 4 corner weights of sample s from lane s with DPP broadcasts (8 full-rate
// moves instead of 45 VALU instructions per lane and sample).  Same arithmetic, bit-identical results.
__device__ __forceinline__ unsigned quad_bcast_u(int s, unsigned v)
{
    const int i = (int)v;
    switch (s) {
        case 0: return (unsigned)__builtin_amdgcn_update_dpp(0, i, MSDA_QUAD_PERM(0, 0, 0, 0), 0xf, 0xf, true);
        case 1: return (unsigned)__builtin_amdgcn_update_dpp(0, i, MSDA_QUAD_PERM(1, 1, 1, 1), 0xf, 0xf, true);
        case 2: return (unsigned)__builtin_amdgcn_update_dpp(0, i, MSDA_QUAD_PERM(2, 2, 2, 2), 0xf, 0xf, true);
        default: return (unsigned)__builtin_amdgcn_update_dpp(0, i, MSDA_QUAD_PERM(3, 3, 3, 3), 0xf, 0xf, true);
    }
}

template <typename VT, int WAVES>
__global__ __launch_bounds__(kBlock, WAVES) void quad_forward_shared_kernel(
    const VT *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ starts,
    const float *__restrict__ loc, const float *__restrict__ aw, int total_qm, int S, int M, int Lq,
    unsigned value_bytes, VT *__restrict__ out)
{
    const int t = xcd_block_id() * kBlock + threadIdx.x;
    int qm = t >> 2;
    const int sub = t & 3;
    const bool live = qm < total_qm;
    qm = live ? qm : total_qm - 1;   // keep whole quads converged for the DPP broadcasts
    const int m = qm % M;
    const int n = (qm / M) / Lq;
    const int row_bytes = M * kD * (int)sizeof(VT);
    const unsigned head_byte = (unsigned)n * (unsigned)S * (unsigned)row_bytes + (unsigned)(m * kD) * (unsigned)sizeof(VT);
    const unsigned sub_byte = (unsigned)(sub * 8) * (unsigned)sizeof(VT);
    const __amdgpu_buffer_rsrc_t vr = __builtin_amdgcn_make_buffer_rsrc((void *)value, 0, value_bytes, 0x00020000);
    // quad lane p loads point p of the 4 levels: (x, y) and the attention weight
    const float2 *loc2 = reinterpret_cast<const float2 *>(loc) + (long)qm * (kL * kP) + sub;
    const float *aw1 = aw + (long)qm * (kL * kP) + sub;
    float2 xy0 = loc2[0], xy1 = loc2[kP], xy2 = loc2[2 * kP], xy3 = loc2[3 * kP];
    float w0 = aw1[0], w1 = aw1[kP], w2 = aw1[2 * kP], w3 = aw1[3 * kP];
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int l = 0; l < kL; ++l) {
        const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1], start = (int)starts[l];
        // this lane's own sample of the level
        const Corners c = corners_of<sizeof(VT)>(xy0.x, xy0.y, H, W, start, row_bytes, head_byte);
        const float a = c.hh * w0, b = c.lh * w0;
        const float k00 = a * c.hw, k01 = a * c.lw, k10 = b * c.hw, k11 = b * c.lw;
#pragma unroll
        for (int s = 0; s < kP; ++s) {
            const unsigned o00 = quad_bcast_u(s, c.o00) + sub_byte, o01 = quad_bcast_u(s, c.o01) + sub_byte;
            const unsigned o10 = quad_bcast_u(s, c.o10) + sub_byte, o11 = quad_bcast_u(s, c.o11) + sub_byte;
            const typename Corner8<VT>::raw r00 = Corner8<VT>::load(vr, o00), r01 = Corner8<VT>::load(vr, o01);
            const typename Corner8<VT>::raw r10 = Corner8<VT>::load(vr, o10), r11 = Corner8<VT>::load(vr, o11);
            fma8<VT>(__uint_as_float(quad_bcast_u(s, __float_as_uint(k00))), r00, acc);
            fma8<VT>(__uint_as_float(quad_bcast_u(s, __float_as_uint(k01))), r01, acc);
            fma8<VT>(__uint_as_float(quad_bcast_u(s, __float_as_uint(k10))), r10, acc);
            fma8<VT>(__uint_as_float(quad_bcast_u(s, __float_as_uint(k11))), r11, acc);
            if (s & 1) __builtin_amdgcn_sched_barrier(0);
        }
        xy0 = xy1; xy1 = xy2; xy2 = xy3;
        w0 = w1; w1 = w2; w2 = w3;
    }
    if (live) Vec8<VT>::store(out + (long)qm * kD + sub * 8, acc);
}

// ------------------------------------------------------------------------------------------
// forward, window-staged ("tile" kernel; encoder self-attention, where the queries are the pixels)
// ------------------------------------------------------------------------------------------
// The direct-gather kernel above is bound by the texture-address path: every wave load touches 16 distinct
// 64-byte segments and the 16 samples x 4 corners of a (query, head) re-read the same neighbourhood -- 2.9 GB
// of gathers per batch-4 encoder call against 45 MB of value.  Here a workgroup owns one head and a 16 x TH
// tile of spatially adjacent queries, and for each sampled level
//   1. the bounding box of every in-level corner its queries touch is found with LDS integer min/max,
//   2. that window of the head's value rows is copied into LDS ONCE (LDS-DMA, lane-linear image:
//      pixel-major, 64/128 bytes per pixel),
//   3. the quads read their corners from LDS (ds_read_b128; 4x the bandwidth of the texture path).
// A window that does not fit the buffer (far offsets, or a coarse-level tile sampling a fine level) makes the
// whole workgroup take the direct-gather route for that level, so any input is handled.  Out-of-level
// corners read a zeroed slot with weight 0 (exact zero padding even if the value tensor holds Inf/NaN).
constexpr int kTileW = 16;
constexpr int kWinBytes = 64 * 1024;            // LDS window buffer (incl. the zero slot)
constexpr int kZeroSlot = 128;                  // bytes
constexpr int kWinSlack = 1024;                 // the last DMA wave may overrun by up to 63 pieces

__host__ __device__ inline int tile_count(int H, int W, int TH)
{
    return ((H + TH - 1) / TH) * ((W + kTileW - 1) / kTileW);
}

// `win` = the window buffer (its first kZeroSlot bytes are zero), `wbase` = byte offset of this level's stretch
template <typename VT>
__device__ __forceinline__ void tile_sample(const unsigned char *win, int wbase, float x, float y, float w, int H,
                                            int W, int bx0, int by0, int ww, int sub, float (&acc)[8])
{
    constexpr int ROWB = kD * (int)sizeof(VT);
    const float Hf = (float)H, Wf = (float)W;
    float h = fmaf(y, Hf, -0.5f), v = fmaf(x, Wf, -0.5f);
    h = fminf(fmaxf(h, -2.f), Hf + 1.f);
    v = fminf(fmaxf(v, -2.f), Wf + 1.f);
    const float hf = floorf(h), wf = floorf(v);
    const float lh = h - hf, lw = v - wf, hh = 1.f - lh, hw = 1.f - lw;
    const int ih = (int)hf, iw = (int)wf;
    const bool y0 = (unsigned)ih < (unsigned)H, y1 = (unsigned)(ih + 1) < (unsigned)H;
    const bool x0 = (unsigned)iw < (unsigned)W, x1 = (unsigned)(iw + 1) < (unsigned)W;
    const int base = wbase + (__mul24(ih - by0, ww) + (iw - bx0)) * ROWB;
    const int lane = sub * 8 * (int)sizeof(VT);
    const int down = ww * ROWB;
    const int a00 = (y0 && x0) ? base + lane : lane;              // invalid corner -> the zero slot
    const int a01 = (y0 && x1) ? base + ROWB + lane : lane;
    const int a10 = (y1 && x0) ? base + down + lane : lane;
    const int a11 = (y1 && x1) ? base + down + ROWB + lane : lane;
    const typename Vec8<VT>::raw r00 = Vec8<VT>::load_raw(reinterpret_cast<const VT *>(win + a00));
    const typename Vec8<VT>::raw r01 = Vec8<VT>::load_raw(reinterpret_cast<const VT *>(win + a01));
    const typename Vec8<VT>::raw r10 = Vec8<VT>::load_raw(reinterpret_cast<const VT *>(win + a10));
    const typename Vec8<VT>::raw r11 = Vec8<VT>::load_raw(reinterpret_cast<const VT *>(win + a11));
    const float a = hh * w, b = lh * w;
    Vec8<VT>::fma(a * hw, r00, acc);
    Vec8<VT>::fma(a * lw, r01, acc);
    Vec8<VT>::fma(b * hw, r10, acc);
    Vec8<VT>::fma(b * lw, r11, acc);
}

template <typename VT, int TH, int WAVES>
__global__ __launch_bounds__(kTileW *TH * 4, WAVES) void tile_forward_kernel(
    const VT *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ starts,
    const float *__restrict__ loc, const float *__restrict__ aw, int N, int S, int M, int Lq,
    unsigned value_bytes, VT *__restrict__ out, int dbg)
{
    constexpr int THREADS = kTileW * TH * 4;
    constexpr int ROWB = kD * (int)sizeof(VT);
    constexpr int PPR = ROWB / 16;                                   // 16-byte pieces per pixel row
    constexpr int MAXPX = (kWinBytes - kZeroSlot - kWinSlack) / ROWB;
    MSDA_DYNAMIC_LDS_ALIGNED(unsigned char, tile_lds, 1024);
    unsigned char *win = tile_lds;
    int *box = reinterpret_cast<int *>(tile_lds + kWinBytes);        // [4 levels][x0, y0, -x1, -y1] (all via min)

    int tiles_per_image = 0;
#pragma unroll
    for (int l = 0; l < kL; ++l) tiles_per_image += tile_count((int)shapes[2 * l], (int)shapes[2 * l + 1], TH);
    const int total_items = N * tiles_per_image * M;
    constexpr int kXcds = 8;
    const int per_xcd = (total_items + kXcds - 1) / kXcds;
    const int xcd = blockIdx.x % kXcds, lane_blk = blockIdx.x / kXcds, blks = gridDim.x / kXcds;
    const int item_end = min(total_items, (xcd + 1) * per_xcd);
    const int row_bytes = M * ROWB;
    const __amdgpu_buffer_rsrc_t vr = __builtin_amdgcn_make_buffer_rsrc((void *)value, 0, value_bytes, 0x00020000);
    if (threadIdx.x < kZeroSlot / 4) reinterpret_cast<int *>(win)[threadIdx.x] = 0;

    for (int item = xcd * per_xcd + lane_blk; item < item_end; item += blks) {
        // Everything below is derived from an opaque copy of the thread id: otherwise the compiler hoists
        // the per-thread address arithmetic of all four levels out of the item loop and spills it.
        int tid = threadIdx.x;
        MSDA_ASM_OPAQUE(tid);
        const int sub = tid & 3, ql = tid >> 2;
        // own level (= quad lane): dimensions for the bounding box
        const int Hs = (int)shapes[2 * sub], Ws = (int)shapes[2 * sub + 1];
        const int m = item % M;
        int t = (item / M) % tiles_per_image;
        const int n = item / (M * tiles_per_image);
        int lq = 0, Hq = 1, Wq = 1;
#pragma unroll
        for (int l = 0; l < kL; ++l) {
            const int Hl = (int)shapes[2 * l], Wl = (int)shapes[2 * l + 1];
            const int nt = tile_count(Hl, Wl, TH);
            if (t >= 0 && t < nt) { lq = l; Hq = Hl; Wq = Wl; t -= 1 << 30; }
            else if (t >= 0) t -= nt;
        }
        t += 1 << 30;
        const int tiles_x = (Wq + kTileW - 1) / kTileW;
        const int qy = (t / tiles_x) * TH + (ql >> 4), qx = (t % tiles_x) * kTileW + (ql & 15);
        const bool live = qy < Hq && qx < Wq;
        const int q = live ? (int)starts[lq] + qy * Wq + qx : (int)starts[lq];
        const long qm = ((long)n * Lq + q) * M + m;
        const unsigned img_byte = (unsigned)n * (unsigned)S * (unsigned)row_bytes + (unsigned)(m * ROWB);
        const unsigned lane_byte = img_byte + (unsigned)(sub * 8 * (int)sizeof(VT));

        const float4 *loc4 = reinterpret_cast<const float4 *>(loc) + qm * 8 + sub * 2;
        float4 la = loc4[0], lb = loc4[1];
        float4 wa = reinterpret_cast<const float4 *>(aw)[qm * 4 + sub];
        if (!live) wa = make_float4(0.f, 0.f, 0.f, 0.f);

        // ---- bounding boxes: quad lane j covers level j's 4 points --------------------------------------
        if (tid < 16) box[tid] = 0x3fffffff;
        __syncthreads();
        if (live && !(MSDA_DBG(dbg) & 64)) {
            int mnx = 0x3fffffff, mny = 0x3fffffff, mxx = -0x3fffffff, mxy = -0x3fffffff;
            const float px[4] = {la.x, la.z, lb.x, lb.z}, py[4] = {la.y, la.w, lb.y, lb.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float h = fmaf(py[k], (float)Hs, -0.5f), v = fmaf(px[k], (float)Ws, -0.5f);
                h = fminf(fmaxf(h, -2.f), (float)Hs + 1.f);
                v = fminf(fmaxf(v, -2.f), (float)Ws + 1.f);
                const int ih = (int)floorf(h), iw = (int)floorf(v);
                const int ylo = max(ih, 0), yhi = min(ih + 1, Hs - 1), xlo = max(iw, 0), xhi = min(iw + 1, Ws - 1);
                if (ylo <= yhi && xlo <= xhi) {
                    mnx = min(mnx, xlo); mxx = max(mxx, xhi); mny = min(mny, ylo); mxy = max(mxy, yhi);
                }
            }
            if (mnx <= mxx) {
                atomicMin(&box[sub * 4 + 0], mnx);
                atomicMin(&box[sub * 4 + 1], mny);
                atomicMin(&box[sub * 4 + 2], -mxx);
                atomicMin(&box[sub * 4 + 3], -mxy);
            }
        }
        __syncthreads();

        // ---- windows of all four levels, staged back to back (one DMA latency, one barrier) ----------------
        // Level l gets the next free stretch of the buffer if its window fits, otherwise it is gathered
        // directly.  The boxes are read into scalars now: the next item re-initialises box[] early.
        int bx0[kL], by0[kL], ww[kL], wbase[kL];
        bool staged[kL], any[kL];
        int used = 0;
#pragma unroll
        for (int l = 0; l < kL; ++l) {
            bx0[l] = __builtin_amdgcn_readfirstlane(box[l * 4 + 0]);
            by0[l] = __builtin_amdgcn_readfirstlane(box[l * 4 + 1]);
            const int bx1 = -__builtin_amdgcn_readfirstlane(box[l * 4 + 2]);
            const int by1 = -__builtin_amdgcn_readfirstlane(box[l * 4 + 3]);
            ww[l] = bx1 - bx0[l] + 1;
            any[l] = bx0[l] <= bx1 && by0[l] <= by1;
            const int npx = any[l] ? ww[l] * (by1 - by0[l] + 1) : 0;
            staged[l] = npx > 0 && used + npx <= MAXPX;
            wbase[l] = kZeroSlot + used * ROWB;
            if (staged[l] && !(MSDA_DBG(dbg) & 16)) {
                // window image: pixel-major, piece-minor -> piece i lands at byte 16 i of the stretch: exactly
                // the lane-linear layout an LDS-DMA instruction writes
                const int W = (int)shapes[2 * l + 1], start = (int)starts[l];
                const int pieces = npx * PPR;
                const float inv_ww = 1.f / (float)ww[l];
                const unsigned lvl_byte = img_byte + (unsigned)__mul24(start + by0[l] * W + bx0[l], row_bytes);
                for (int i0 = (tid & ~63); i0 < pieces; i0 += THREADS) {
                    const int i = min(i0 + (tid & 63), pieces - 1);          // surplus lanes repeat the last piece
                    const int pxi = i / PPR, part = i % PPR;
                    const int wy = (int)(((float)pxi + 0.5f) * inv_ww);
                    const int wx = pxi - wy * ww[l];
                    const unsigned off = lvl_byte + (unsigned)__mul24(wy * W + wx, row_bytes) + (unsigned)(part * 16);
                    const unsigned char *src = reinterpret_cast<const unsigned char *>(value) + off;
                    MSDA_GLOBAL_LOAD_LDS16(src, win + wbase[l] + i0 * 16);
                }
                // a stretch is rounded up to whole 64-piece DMA waves so that the next one cannot be overrun
                used += (npx * PPR + 63) / 64 * 64 / PPR;
            }
        }
        __syncthreads();

        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int l = 0; l < kL; ++l) {
            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1], start = (int)starts[l];
            // (scheduling fences: the compiler otherwise hoists all 16 corner reads of a level and spills)
            if (MSDA_DBG(dbg) & 32) {
            } else if (staged[l]) {
                tile_sample<VT>(win, wbase[l], quad_bcast<0>(la.x), quad_bcast<0>(la.y), quad_bcast<0>(wa.x), H, W, bx0[l], by0[l], ww[l], sub, acc);
                __builtin_amdgcn_sched_barrier(0);
                tile_sample<VT>(win, wbase[l], quad_bcast<0>(la.z), quad_bcast<0>(la.w), quad_bcast<0>(wa.y), H, W, bx0[l], by0[l], ww[l], sub, acc);
                __builtin_amdgcn_sched_barrier(0);
                tile_sample<VT>(win, wbase[l], quad_bcast<0>(lb.x), quad_bcast<0>(lb.y), quad_bcast<0>(wa.z), H, W, bx0[l], by0[l], ww[l], sub, acc);
                __builtin_amdgcn_sched_barrier(0);
                tile_sample<VT>(win, wbase[l], quad_bcast<0>(lb.z), quad_bcast<0>(lb.w), quad_bcast<0>(wa.w), H, W, bx0[l], by0[l], ww[l], sub, acc);
                __builtin_amdgcn_sched_barrier(0);
            } else if (any[l]) {
                // rare route (window too large for the buffer): direct gathers, one sample at a time
                float4 ta = la, tb = lb, tw = wa;
#pragma unroll 1
                for (int k = 0; k < 4; ++k) {
                    fwd_sample<VT>(vr, quad_bcast<0>(ta.x), quad_bcast<0>(ta.y), quad_bcast<0>(tw.x), H, W, start, row_bytes, lane_byte, acc);
                    ta = make_float4(ta.z, ta.w, tb.x, tb.y); tb = make_float4(tb.z, tb.w, 0.f, 0.f);
                    tw = make_float4(tw.y, tw.z, tw.w, 0.f);
                }
            }
            quad_rotate(la); quad_rotate(lb); quad_rotate(wa);
        }
        if (live) Vec8<VT>::store(out + qm * kD + sub * 8, acc);
    }
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
__device__ __forceinline__ void bwd_sample(__amdgpu_buffer_rsrc_t vr, float *__restrict__ gbase, float x, float y,
                                           float w, int H, int W, int start, int row_bytes, unsigned lane_byte,
                                           bool live, const float (&tg)[8], float &g_a, float &g_w, float &g_h)
{
    const Corners c = corners_of<sizeof(VT)>(x, y, H, W, start, row_bytes, lane_byte);
    const typename Corner8<VT>::raw r00 = Corner8<VT>::load(vr, c.o00), r01 = Corner8<VT>::load(vr, c.o01);
    const typename Corner8<VT>::raw r10 = Corner8<VT>::load(vr, c.o10), r11 = Corner8<VT>::load(vr, c.o11);
    // the reference drops a sample whose pixel coordinate is not inside (-1, size) (.cuh:285); for the
    // value that is implied by the corner range checks, for the location gradient at the exact
    // boundary it is not, so the backward pass keeps the explicit test
    const float h_im = fmaf(y, (float)H, -0.5f), w_im = fmaf(x, (float)W, -0.5f);
    const bool inside = (h_im > -1.f) && (w_im > -1.f) && (h_im < (float)H) && (w_im < (float)W);
    const float wgt = inside ? w : 0.f;
    if (SCATTER && live) {
        // grad_value: w_k * attn * grad_out, 8 channels per corner per lane (reference-style direct scatter)
        const float a = c.hh * wgt, b = c.lh * wgt;
        const int esz = (int)sizeof(VT);
        if (c.v00) scatter8(gbase + c.o00 / esz, a * c.hw, tg);
        if (c.v01) scatter8(gbase + c.o01 / esz, a * c.lw, tg);
        if (c.v10) scatter8(gbase + c.o10 / esz, b * c.hw, tg);
        if (c.v11) scatter8(gbase + c.o11 / esz, b * c.lw, tg);
    }
    // channel reductions: 8 channels in-lane, then across the quad with DPP (out-of-level corners read zeros)
    float f[8];
    Corner8<VT>::unpack(r00, f); const float e1 = quad_sum(dot8(tg, f));
    Corner8<VT>::unpack(r01, f); const float e2 = quad_sum(dot8(tg, f));
    Corner8<VT>::unpack(r10, f); const float e3 = quad_sum(dot8(tg, f));
    Corner8<VT>::unpack(r11, f); const float e4 = quad_sum(dot8(tg, f));
    g_a = inside ? c.hh * (c.hw * e1 + c.lw * e2) + c.lh * (c.hw * e3 + c.lw * e4) : 0.f;
    g_w = (float)W * wgt * (c.hh * (e2 - e1) + c.lh * (e4 - e3));
    g_h = (float)H * wgt * (c.hw * (e3 - e1) + c.lw * (e4 - e2));
}

// SCATTER = false: only grad_sampling_loc / grad_attn_weight ("K1"; grad_value is then produced by
// the sorted scatter of msda_window.hip).
template <typename VT, int WAVES, bool SCATTER>
__global__ __launch_bounds__(kBlock, WAVES) void quad_backward_kernel(
    const VT *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ starts,
    const float *__restrict__ loc, const float *__restrict__ aw, const VT *__restrict__ grad_out, int total_qm,
    int S, int M, int Lq, unsigned value_bytes, float *__restrict__ g_value, float *__restrict__ g_loc,
    float *__restrict__ g_aw)
{
    const int t = xcd_block_id() * kBlock + threadIdx.x;
    int qm = t >> 2;
    const int sub = t & 3;
    const bool live = qm < total_qm;
    qm = live ? qm : total_qm - 1;
    const int m = qm % M;
    const int n = (qm / M) / Lq;
    const int row_bytes = M * kD * (int)sizeof(VT);
    const unsigned lane_byte = (unsigned)n * (unsigned)S * (unsigned)row_bytes
                               + (unsigned)(m * kD + sub * 8) * (unsigned)sizeof(VT);
    const __amdgpu_buffer_rsrc_t vr = __builtin_amdgcn_make_buffer_rsrc((void *)value, 0, value_bytes, 0x00020000);
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
        bwd_sample<VT, SCATTER>(vr, g_value, quad_bcast<0>(la.x), quad_bcast<0>(la.y), quad_bcast<0>(wa.x), H, W, start,
                                row_bytes, lane_byte, live, tg, rw.x, ra.x, ra.y);
        __builtin_amdgcn_sched_barrier(0);
        bwd_sample<VT, SCATTER>(vr, g_value, quad_bcast<0>(la.z), quad_bcast<0>(la.w), quad_bcast<0>(wa.y), H, W, start,
                                row_bytes, lane_byte, live, tg, rw.y, ra.z, ra.w);
        __builtin_amdgcn_sched_barrier(0);
        bwd_sample<VT, SCATTER>(vr, g_value, quad_bcast<0>(lb.x), quad_bcast<0>(lb.y), quad_bcast<0>(wa.z), H, W, start,
                                row_bytes, lane_byte, live, tg, rw.z, rb.x, rb.y);
        __builtin_amdgcn_sched_barrier(0);
        bwd_sample<VT, SCATTER>(vr, g_value, quad_bcast<0>(lb.z), quad_bcast<0>(lb.w), quad_bcast<0>(wa.w), H, W, start,
                                row_bytes, lane_byte, live, tg, rw.w, rb.z, rb.w);
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

// Geometry-sharing K1 (grad_sampling_loc / grad_attn_weight only): as quad_forward_shared_kernel, quad lane p
// owns point p of every level -- it computes that sample's geometry once, the quad fetches the corner offsets
// with DPP broadcasts, all four lanes take part in the channel dot products (8 channels each + quad reduction),
// and lane p turns the four reduced dots of ITS sample into (g_x, g_y, g_aw).  PMC on the kernel above: VALU
// busy 75 % of the kernel time with ~45 of ~150 instructions per sample being the replicated geometry.
//
// REFDIM != 0: the backward of the module's sampling geometry runs as the epilogue (softmax backward over the quad,
// offset scaling; msda_geometry.h) and the kernel writes the gradient of the raw projection row [offsets | logits]
// in value's dtype -- grad_sampling_loc / grad_attn_weight (192 B per (query, head), float32) never reach memory.
template <typename VT, int WAVES, int REFDIM>
__global__ __launch_bounds__(kBlock, WAVES) void quad_backward_shared_kernel(
    const VT *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ starts,
    const float *__restrict__ loc, const float *__restrict__ aw, const VT *__restrict__ grad_out, int total_qm,
    int S, int M, int Lq, unsigned value_bytes, float *__restrict__ g_loc, float *__restrict__ g_aw,
    const float *__restrict__ ref, VT *__restrict__ g_qproj)
{
    // REFDIM | 8 (ablation build only, launch_quad_backward_gated): the kernel runs only when a gate word is non-zero -- the
    // fallback of cell_backward_kernel's "far sample: stop here" arm.  The word travels in the pointer argument the instantiation
    // does not use (g_loc with a geometry epilogue, ref without): the product instantiations' argument list stays what it is.
    constexpr bool GATED = (REFDIM & 8) != 0;
    constexpr int RDIM = REFDIM & 7;
    if (GATED) {
        const int *gate = RDIM != 0 ? reinterpret_cast<const int *>(g_loc) : reinterpret_cast<const int *>(ref);
        if (*gate == 0) return;
    }
    const int t = xcd_block_id() * kBlock + threadIdx.x;
    int qm = t >> 2;
    const int sub = t & 3;
    const bool live = qm < total_qm;
    qm = live ? qm : total_qm - 1;
    const int m = qm % M;
    const int n = (qm / M) / Lq;
    const int row_bytes = M * kD * (int)sizeof(VT);
    const unsigned head_byte = (unsigned)n * (unsigned)S * (unsigned)row_bytes + (unsigned)(m * kD) * (unsigned)sizeof(VT);
    const unsigned sub_byte = (unsigned)(sub * 8) * (unsigned)sizeof(VT);
    const __amdgpu_buffer_rsrc_t vr = __builtin_amdgcn_make_buffer_rsrc((void *)value, 0, value_bytes, 0x00020000);
    // memory side as in quad_backward_kernel: quad lane j loads / stores the 4 points of level j as whole
    // 16-byte vectors (128 + 64 contiguous bytes per quad); the per-point ownership below is a register
    // redistribution inside the quad (8-byte per-lane accesses measured 55 us slower)
    const float4 *loc4 = reinterpret_cast<const float4 *>(loc) + (long)qm * 8 + sub * 2;
    float4 la = loc4[0], lb = loc4[1];
    float4 wa = reinterpret_cast<const float4 *>(aw)[(long)qm * 4 + sub];
    float tg[8];
    Vec8<VT>::load(grad_out + (long)qm * kD + sub * 8, tg);
    float4 gla = make_float4(0.f, 0.f, 0.f, 0.f), glb = gla, ga = gla;
    auto pick = [&](float v0, float v1, float v2, float v3) {     // element `sub` of the level-owner's vector
        const float b0 = quad_bcast<0>(v0), b1 = quad_bcast<0>(v1), b2 = quad_bcast<0>(v2), b3 = quad_bcast<0>(v3);
        return sub == 0 ? b0 : sub == 1 ? b1 : sub == 2 ? b2 : b3;
    };
#pragma unroll 1
    for (int l = 0; l < kL; ++l) {
        const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1], start = (int)starts[l];
        // (after l rotations quad lane 0 holds level l) -> this lane's own point of the level
        const float px = pick(la.x, la.z, lb.x, lb.z), py = pick(la.y, la.w, lb.y, lb.w);
        const float pw = pick(wa.x, wa.y, wa.z, wa.w);
        const Corners c = corners_of<sizeof(VT)>(px, py, H, W, start, row_bytes, head_byte);
        const float h_im = fmaf(py, (float)H, -0.5f), w_im = fmaf(px, (float)W, -0.5f);
        const bool inside = (h_im > -1.f) && (w_im > -1.f) && (h_im < (float)H) && (w_im < (float)W);
        const float wgt = inside ? pw : 0.f;
        float out_a = 0.f, out_x = 0.f, out_y = 0.f;
#pragma unroll
        for (int s = 0; s < kP; ++s) {
            const unsigned o00 = quad_bcast_u(s, c.o00) + sub_byte, o01 = quad_bcast_u(s, c.o01) + sub_byte;
            const unsigned o10 = quad_bcast_u(s, c.o10) + sub_byte, o11 = quad_bcast_u(s, c.o11) + sub_byte;
            const typename Corner8<VT>::raw r00 = Corner8<VT>::load(vr, o00), r01 = Corner8<VT>::load(vr, o01);
            const typename Corner8<VT>::raw r10 = Corner8<VT>::load(vr, o10), r11 = Corner8<VT>::load(vr, o11);
            float f[8];
            Corner8<VT>::unpack(r00, f); const float e1 = quad_sum(dot8(tg, f));
            Corner8<VT>::unpack(r01, f); const float e2 = quad_sum(dot8(tg, f));
            Corner8<VT>::unpack(r10, f); const float e3 = quad_sum(dot8(tg, f));
            Corner8<VT>::unpack(r11, f); const float e4 = quad_sum(dot8(tg, f));
            // every lane evaluates the formulas with ITS sample's fractions; only lane s keeps the result
            const float a_ = inside ? c.hh * (c.hw * e1 + c.lw * e2) + c.lh * (c.hw * e3 + c.lw * e4) : 0.f;
            const float x_ = (float)W * wgt * (c.hh * (e2 - e1) + c.lh * (e4 - e3));
            const float y_ = (float)H * wgt * (c.hw * (e3 - e1) + c.lw * (e4 - e2));
            if (sub == s) { out_a = a_; out_x = x_; out_y = y_; }
            if (s & 1) __builtin_amdgcn_sched_barrier(0);
        }
        // point p's results live in lane p: gather the level's 4 points into vectors, owner lane l keeps them
        const float4 ra = make_float4(quad_bcast<0>(out_x), quad_bcast<0>(out_y), quad_bcast<1>(out_x), quad_bcast<1>(out_y));
        const float4 rb = make_float4(quad_bcast<2>(out_x), quad_bcast<2>(out_y), quad_bcast<3>(out_x), quad_bcast<3>(out_y));
        const float4 rw = make_float4(quad_bcast<0>(out_a), quad_bcast<1>(out_a), quad_bcast<2>(out_a), quad_bcast<3>(out_a));
        if (sub == l) { gla = ra; glb = rb; ga = rw; }
        quad_rotate(la); quad_rotate(lb); quad_rotate(wa);
    }
    if (RDIM == 0) {
        if (live) {
            float4 *gl4 = reinterpret_cast<float4 *>(g_loc) + (long)qm * 8 + sub * 2;
            gl4[0] = gla;
            gl4[1] = glb;
            reinterpret_cast<float4 *>(g_aw)[(long)qm * 4 + sub] = ga;
        }
    } else if (live) {          // (a quad is live or dead as a whole; after 4 rotations wa is this lane's level again)
        constexpr int RD = RDIM == 0 ? 2 : RDIM;
        const long row = qm / M;
        const float a[4] = {wa.x, wa.y, wa.z, wa.w};
        float g[4] = {ga.x, ga.y, ga.z, ga.w};
        const float gl[8] = {gla.x, gla.y, gla.z, gla.w, glb.x, glb.y, glb.z, glb.w};
        geom::backward<VT, RD>(g_qproj + row * (M * 48), ref + row * (kL * RD), shapes, m, M, sub, a, g, gl);
    }
}

}  // namespace

static unsigned value_bytes(const Problem &p)
{
    return (unsigned)((size_t)p.N * p.S * p.M * kD * (p.dtype == MSDA_F32 ? 4 : 2));
}

bool quad_supports(const Problem &p)
{
    // buffer addressing: the whole value tensor within 32-bit byte offsets, below the kOob sentinel
    if ((size_t)p.N * p.S * p.M * kD * (p.dtype == MSDA_F32 ? 4 : 2) >= 0xFFFFFF00ull - 64) return false;
    if ((long)p.S + 1 >= (1L << 23) || (long)p.M * kD * 4 >= (1L << 23)) return false;   // 24-bit multiplies
    if (p.dtype != MSDA_F32 && p.dtype != MSDA_BF16) return false;
    if (p.D != kD || p.L != kL || p.P != kP) return false;
    const long total_qm = (long)p.N * p.Lq * p.M;
    if (total_qm * 4 >= (1L << 31)) return false;
    if ((long)p.S * p.M * kD >= (1L << 31)) return false;   // per-image element offsets are 32-bit
    return true;
}

// the coarse-levels-in-LDS forward pays once a workgroup's staging (<= 96 KB) is spread over thousands of queries
bool coarse_forward_applies(const Problem &p)
{
    return p.dtype == MSDA_BF16 && quad_supports(p) && p.Lq >= 4096 && p.N * p.M <= 256;
}

static void launch_coarse_forward(const Problem &p, const Fused *f)
{
    const int pairs = p.N * p.M;
    const int wgs = 256 / pairs > 0 ? 256 / pairs : 1;
    const dim3 grid(pairs * wgs), block(kCoarseBlock);
#define MSDA_COARSE(RD, SAVE)                                                                                         \
    hipLaunchKernelGGL((quad_forward_coarse_kernel<RD, SAVE>), grid, block, kCoarseBytes, p.stream,                  \
                       (const bf16_t *)p.value, p.shapes, p.starts, (const float *)p.loc, (const float *)p.aw,       \
                       (const bf16_t *)(f ? f->qproj : nullptr), f ? f->ref : nullptr, p.N, p.S, p.M, p.Lq, wgs,     \
                       value_bytes(p), (bf16_t *)p.out, f ? f->loc_save : nullptr, f ? f->aw_save : nullptr,         \
                       ablation_env("RLIPV2_COARSE_STAGED", kL))
    RLIPV2_ONCE_PER_DEVICE(      // more than 64 KB of dynamic LDS has to be asked for, once per kernel and device
        (void)hipFuncSetAttribute((const void *)quad_forward_coarse_kernel<0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kCoarseBytes);
        (void)hipFuncSetAttribute((const void *)quad_forward_coarse_kernel<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kCoarseBytes);
        (void)hipFuncSetAttribute((const void *)quad_forward_coarse_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kCoarseBytes);
        (void)hipFuncSetAttribute((const void *)quad_forward_coarse_kernel<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kCoarseBytes);
        (void)hipFuncSetAttribute((const void *)quad_forward_coarse_kernel<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kCoarseBytes));
    if (!f && ablation_env("RLIPV2_COARSE_BLOCK", 1024) == 256) {      // ablation: small one-shot workgroups, nothing staged
        const int w256 = wgs * 4;
        hipLaunchKernelGGL((quad_forward_coarse_kernel<0, false, 256>), dim3(pairs * w256), dim3(256), kCoarseBytes, p.stream,
                           (const bf16_t *)p.value, p.shapes, p.starts, (const float *)p.loc, (const float *)p.aw,
                           (const bf16_t *)nullptr, (const float *)nullptr, p.N, p.S, p.M, p.Lq, w256, value_bytes(p),
                           (bf16_t *)p.out, (float *)nullptr, (float *)nullptr, 0);
        return;
    }
    if (!f) MSDA_COARSE(0, false);
    else if (f->refdim == 2) { if (f->loc_save) MSDA_COARSE(2, true); else MSDA_COARSE(2, false); }
    else { if (f->loc_save) MSDA_COARSE(4, true); else MSDA_COARSE(4, false); }
#undef MSDA_COARSE
}

void launch_quad_forward_coarse(const Problem &p) { launch_coarse_forward(p, nullptr); }

void launch_quad_forward(const Problem &p)
{
    const int dbg = ablation_env("RLIPV2_MSDA_DEBUG", 0);       // ablation builds only
    const int total_qm = p.N * p.Lq * p.M;
    const int grid = (int)(((long)total_qm * 4 + kBlock - 1) / kBlock);
    // geometry-sharing variant: measured 231 vs 268 us for float32 (model-like encoder input), no gain for bf16
    // (162 vs 165 us model-like, 296 vs 274 us uniform: that kernel is bound by the gather path, not by VALU)
    static const int shared = ablation_env("RLIPV2_MSDA_FWD_SHARED", -1);
    if ((shared == 1 || (shared == -1 && p.dtype == MSDA_F32)) && !dbg) {
        if (p.dtype == MSDA_F32)
            hipLaunchKernelGGL((quad_forward_shared_kernel<float, 6>), dim3(grid), dim3(kBlock), 0, p.stream,
                               (const float *)p.value, p.shapes, p.starts, (const float *)p.loc, (const float *)p.aw,
                               total_qm, p.S, p.M, p.Lq, value_bytes(p), (float *)p.out);
        else
            hipLaunchKernelGGL((quad_forward_shared_kernel<bf16_t, 5>), dim3(grid), dim3(kBlock), 0, p.stream,
                               (const bf16_t *)p.value, p.shapes, p.starts, (const float *)p.loc, (const float *)p.aw,
                               total_qm, p.S, p.M, p.Lq, value_bytes(p), (bf16_t *)p.out);
        return;
    }
    if (p.dtype == MSDA_F32)
        hipLaunchKernelGGL((quad_forward_kernel<float, 6, 2>), dim3(grid), dim3(kBlock), 0, p.stream,
                           (const float *)p.value, p.shapes, p.starts, (const float *)p.loc, (const float *)p.aw,
                           total_qm, p.S, p.M, p.Lq, value_bytes(p), (float *)p.out, dbg);
    else {
        // register budget = 512 / waves per SIMD: 4 (128 VGPRs, no spills) 147 us, 5 (96 VGPRs, 16 bytes spilled) 155-163 us,
        // 6 / 8 spill into the gather loop (415 / 823 us); 2-3 change nothing (the kernel needs ~110)
        static const int waves = ablation_env("RLIPV2_MSDA_FWD_WAVES", 4);
#define MSDA_FWD_BF16(W)                                                                                           \
        hipLaunchKernelGGL((quad_forward_kernel<bf16_t, W, 2>), dim3(grid), dim3(kBlock), 0, p.stream,             \
                           (const bf16_t *)p.value, p.shapes, p.starts, (const float *)p.loc, (const float *)p.aw, \
                           total_qm, p.S, p.M, p.Lq, value_bytes(p), (bf16_t *)p.out, dbg)
        if (waves == 5) MSDA_FWD_BF16(5); else if (waves == 3) MSDA_FWD_BF16(3);
        else MSDA_FWD_BF16(4);
#undef MSDA_FWD_BF16
    }
}

// window-staged forward (encoder self-attention shapes); see tile_forward_kernel
void launch_tile_forward(const Problem &p)
{
    static const int th = ablation_env("RLIPV2_MSDA_TILE_H", 8);
    const int dbg = ablation_env("RLIPV2_MSDA_DEBUG", 0);       // ablation builds only
    const int g = ablation_env("RLIPV2_MSDA_GRID", 0);
    const int lds = kWinBytes + 64;
#define MSDA_LAUNCH_TILE(VT, TH, WAVES, BLOCKS_PER_CU)                                                              \
    hipLaunchKernelGGL((tile_forward_kernel<VT, TH, WAVES>), dim3(g ? g : 256 * BLOCKS_PER_CU),               \
                       dim3(kTileW * TH * 4), lds, p.stream, (const VT *)p.value, p.shapes, p.starts,              \
                       (const float *)p.loc, (const float *)p.aw, p.N, p.S, p.M, p.Lq, value_bytes(p), (VT *)p.out, dbg)
    if (p.dtype == MSDA_F32) {
        if (th == 8) MSDA_LAUNCH_TILE(float, 8, 4, 2); else MSDA_LAUNCH_TILE(float, 16, 4, 1);
    } else {
        if (th == 8) MSDA_LAUNCH_TILE(bf16_t, 8, 4, 2);
        else if (th == 12) MSDA_LAUNCH_TILE(bf16_t, 12, 5, 1);
        else MSDA_LAUNCH_TILE(bf16_t, 16, 4, 1);
    }
#undef MSDA_LAUNCH_TILE
}

void launch_quad_backward(const Problem &p)
{
    const int total_qm = p.N * p.Lq * p.M;
    const int grid = (int)(((long)total_qm * 4 + kBlock - 1) / kBlock);
    if (p.dtype == MSDA_F32)
        hipLaunchKernelGGL((quad_backward_kernel<float, 3, true>), dim3(grid), dim3(kBlock), 0, p.stream,
                           (const float *)p.value, p.shapes, p.starts, (const float *)p.loc, (const float *)p.aw,
                           (const float *)p.grad_out, total_qm, p.S, p.M, p.Lq, value_bytes(p), (float *)p.g_value,
                           (float *)p.g_loc, (float *)p.g_aw);
    else
        hipLaunchKernelGGL((quad_backward_kernel<bf16_t, 4, true>), dim3(grid), dim3(kBlock), 0, p.stream,
                           (const bf16_t *)p.value, p.shapes, p.starts, (const float *)p.loc, (const float *)p.aw,
                           (const bf16_t *)p.grad_out, total_qm, p.S, p.M, p.Lq, value_bytes(p), (float *)p.g_value,
                           (float *)p.g_loc, (float *)p.g_aw);
}

void launch_quad_backward_reduce(const Problem &p)
{
    const int total_qm = p.N * p.Lq * p.M;
    const int grid = (int)(((long)total_qm * 4 + kBlock - 1) / kBlock);
    static const int shared = ablation_env("RLIPV2_MSDA_K1_SHARED", 1);
    if (shared) {
        if (p.dtype == MSDA_F32)
            hipLaunchKernelGGL((quad_backward_shared_kernel<float, 4, 0>), dim3(grid), dim3(kBlock), 0, p.stream,
                               (const float *)p.value, p.shapes, p.starts, (const float *)p.loc, (const float *)p.aw,
                               (const float *)p.grad_out, total_qm, p.S, p.M, p.Lq, value_bytes(p), (float *)p.g_loc,
                               (float *)p.g_aw, (const float *)nullptr, (float *)nullptr);
        else {
            static const int waves = ablation_env("RLIPV2_MSDA_K1_WAVES", 4);
#define MSDA_K1_BF16(W)                                                                                                \
            hipLaunchKernelGGL((quad_backward_shared_kernel<bf16_t, W, 0>), dim3(grid), dim3(kBlock), 0, p.stream,     \
                               (const bf16_t *)p.value, p.shapes, p.starts, (const float *)p.loc, (const float *)p.aw, \
                               (const bf16_t *)p.grad_out, total_qm, p.S, p.M, p.Lq, value_bytes(p), (float *)p.g_loc, \
                               (float *)p.g_aw, (const float *)nullptr, (bf16_t *)nullptr)
            if (waves == 5) MSDA_K1_BF16(5); else if (waves == 3) MSDA_K1_BF16(3);
            else MSDA_K1_BF16(4);     // (4: 763 us whole backward, 5: 778, 6: 944, 8: 1194 -- spills)
#undef MSDA_K1_BF16
        }
        return;
    }
    if (p.dtype == MSDA_F32)
        hipLaunchKernelGGL((quad_backward_kernel<float, 4, false>), dim3(grid), dim3(kBlock), 0, p.stream,
                           (const float *)p.value, p.shapes, p.starts, (const float *)p.loc, (const float *)p.aw,
                           (const float *)p.grad_out, total_qm, p.S, p.M, p.Lq, value_bytes(p), (float *)p.g_value,
                           (float *)p.g_loc, (float *)p.g_aw);
    else
        hipLaunchKernelGGL((quad_backward_kernel<bf16_t, 4, false>), dim3(grid), dim3(kBlock), 0, p.stream,
                           (const bf16_t *)p.value, p.shapes, p.starts, (const float *)p.loc, (const float *)p.aw,
                           (const bf16_t *)p.grad_out, total_qm, p.S, p.M, p.Lq, value_bytes(p), (float *)p.g_value,
                           (float *)p.g_loc, (float *)p.g_aw);
}

// ---- fused sampling geometry (msda_fused_forward / msda_fused_backward_ws) ---------------------------------------------
void launch_quad_forward_fused(const Problem &p, const Fused &f)
{
    if (coarse_forward_applies(p) && ablation_env("RLIPV2_MSDA_FWD_COARSE", 0)) {     // (ablation builds: see msda_api.hip pick())
        launch_coarse_forward(p, &f);
        return;
    }
    const int total_qm = p.N * p.Lq * p.M;
    const int grid = (int)(((long)total_qm * 4 + kBlock - 1) / kBlock);
    const bool save = f.loc_save != nullptr;
#define MSDA_FWD_FUSED(VT, RD, SAVE, W)                                                                                \
    hipLaunchKernelGGL((quad_forward_fused_kernel<VT, RD, SAVE, W>), dim3(grid), dim3(kBlock), 0, p.stream,           \
                       (const VT *)p.value, p.shapes, p.starts, (const VT *)f.qproj, f.ref, total_qm, p.S, p.M, p.Lq, \
                       value_bytes(p), (VT *)p.out, f.loc_save, f.aw_save)
#define MSDA_FWD_FUSED_T(VT, W)                                                                                        \
    do {                                                                                                               \
        if (f.refdim == 2) { if (save) MSDA_FWD_FUSED(VT, 2, true, W); else MSDA_FWD_FUSED(VT, 2, false, W); }         \
        else { if (save) MSDA_FWD_FUSED(VT, 4, true, W); else MSDA_FWD_FUSED(VT, 4, false, W); }                       \
    } while (0)
    if (p.dtype == MSDA_F32) MSDA_FWD_FUSED_T(float, 3);
    else MSDA_FWD_FUSED_T(bf16_t, 4);
#undef MSDA_FWD_FUSED_T
#undef MSDA_FWD_FUSED
}

void launch_quad_backward_reduce_fused(const Problem &p, const Fused &f)
{
    const int total_qm = p.N * p.Lq * p.M;
    const int grid = (int)(((long)total_qm * 4 + kBlock - 1) / kBlock);
#define MSDA_K1_FUSED(VT, RD)                                                                                          \
    hipLaunchKernelGGL((quad_backward_shared_kernel<VT, 4, RD>), dim3(grid), dim3(kBlock), 0, p.stream,               \
                       (const VT *)p.value, p.shapes, p.starts, (const float *)p.loc, (const float *)p.aw,            \
                       (const VT *)p.grad_out, total_qm, p.S, p.M, p.Lq, value_bytes(p), (float *)nullptr,            \
                       (float *)nullptr, f.ref, (VT *)f.g_qproj)
    if (p.dtype == MSDA_F32) { if (f.refdim == 2) MSDA_K1_FUSED(float, 2); else MSDA_K1_FUSED(float, 4); }
    else { if (f.refdim == 2) MSDA_K1_FUSED(bf16_t, 2); else MSDA_K1_FUSED(bf16_t, 4); }
#undef MSDA_K1_FUSED
}

#ifdef MSDA_ABLATION
// K1 (with or without the geometry epilogue) behind a gate word: runs only if *gate != 0 (see quad_backward_shared_kernel)
void launch_quad_backward_gated(const Problem &p, const Fused *f, const int *gate)
{
    const int total_qm = p.N * p.Lq * p.M;
    const int grid = (int)(((long)total_qm * 4 + kBlock - 1) / kBlock);
    float *gate_f = reinterpret_cast<float *>(const_cast<int *>(gate));
#define MSDA_K1_GATED(VT, RD, GLOC, GAW, REF, GQ)                                                                      \
    hipLaunchKernelGGL((quad_backward_shared_kernel<VT, 4, (RD) | 8>), dim3(grid), dim3(kBlock), 0, p.stream,          \
                       (const VT *)p.value, p.shapes, p.starts, (const float *)p.loc, (const float *)p.aw,            \
                       (const VT *)p.grad_out, total_qm, p.S, p.M, p.Lq, value_bytes(p), GLOC, GAW, REF, GQ)
    // (only ever launched behind cell_backward_kernel: bfloat16 calls)
    if (p.dtype != MSDA_BF16) return;
    if (!f) MSDA_K1_GATED(bf16_t, 0, (float *)p.g_loc, (float *)p.g_aw, (const float *)gate_f, (bf16_t *)nullptr);
    else if (f->refdim == 2) MSDA_K1_GATED(bf16_t, 2, gate_f, (float *)nullptr, f->ref, (bf16_t *)f->g_qproj);
    else MSDA_K1_GATED(bf16_t, 4, gate_f, (float *)nullptr, f->ref, (bf16_t *)f->g_qproj);
#undef MSDA_K1_GATED
}
#endif

}  // namespace msda
