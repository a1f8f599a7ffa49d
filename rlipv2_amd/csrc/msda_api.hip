// msda_api.hip -- the C ABI of librlipv2_msda.so (declared in include/rlipv2_msda.h).
//
// Host side of the drop-in for the reference's extension module (reference:
// models/ops/src/vision.cpp:13-16 -> ms_deform_attn.h:36-77 -> cuda/ms_deform_attn_cuda.cu).
// Argument checking mirrors what the reference asserts (contiguity / device are properties of
// raw pointers the caller vouches for; the batch-chunk rule is msda_check_im2col_step);
// kernel launch failures are RETURNED (the reference only printf()s them,
// ms_deform_im2col_cuda.cuh:948-952).
#include "msda_internal.h"

using namespace msda;

namespace {

int validate(int dtype, int N, int S, int M, int D, int L, int Lq, int P)
{
    if (dtype != MSDA_F32 && dtype != MSDA_F64 && dtype != MSDA_BF16) return MSDA_ERR_BAD_DTYPE;
    if (N < 0 || S < 0 || M < 0 || D < 0 || L < 0 || Lq < 0 || P < 0) return MSDA_ERR_BAD_SHAPE;
    return MSDA_OK;
}

size_t value_elem(int dtype) { return dtype == MSDA_F64 ? 8 : dtype == MSDA_F32 ? 4 : 2; }
size_t loc_elem(int dtype) { return dtype == MSDA_F64 ? 8 : 4; }
size_t grad_value_elem(int dtype) { return dtype == MSDA_F64 ? 8 : 4; }

bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

int finish_launch()
{
    return hipGetLastError() == hipSuccess ? MSDA_OK : MSDA_ERR_LAUNCH;
}

int pick(bool backward, const Problem &p)
{
    // backward, model head shape: reduce kernel + sorted scatter kernel (msda_window.hip).  The quad
    // kernel's own scatter (strided 4-byte global atomics, 64 sectors per instruction) measured
    // 34 ms at the encoder shape against 4.6 ms for the generic kernel's 128-byte rows and 1.1 ms
    // for the sorted scatter, so it is never picked automatically.
    if (backward && window_supports(p, true)) return MSDA_VARIANT_WINDOW;
    // forward: the window-staged tile kernel wins for float32 rows (201 vs 259 us at the encoder shape: half
    // the LDS-resident pixels per byte of texture traffic saved); for bf16 the direct-gather kernel is faster
    // (164 vs 215 us: both are VALU-bound on the 32 unpack + 32 FMA per sample, and the tile kernel adds its
    // bounding-box / staging phases), see DESIGN.md section 4
    if (!backward && p.dtype == MSDA_F32 && window_supports(p, false)) return MSDA_VARIANT_WINDOW;
    // (MSDA_VARIANT_COARSE -- the two coarsest levels resident in LDS -- is an explicit choice, not an automatic one:
    //  measured at the encoder shape 160 vs 146 us on model-like locations, 176 vs 263 us on uniform ones)
    if (!backward && quad_supports(p)) return MSDA_VARIANT_QUAD;
    return MSDA_VARIANT_GENERIC;
}

}  // namespace

extern "C" {

int msda_abi_version(void) { return RLIPV2_MSDA_ABI_VERSION; }

const char *msda_strerror(int status)
{
    switch (status) {
        case MSDA_OK: return "ok";
        case MSDA_ERR_BAD_DTYPE: return "unsupported dtype (expected MSDA_F32, MSDA_F64 or MSDA_BF16)";
        case MSDA_ERR_BAD_SHAPE: return "bad shape: negative dimension or index range exceeded";
        case MSDA_ERR_NULL_POINTER: return "null pointer for a non-empty tensor";
        case MSDA_ERR_IM2COL_STEP: return "batch must divide im2col_step: batch % min(batch, im2col_step) != 0";
        case MSDA_ERR_LAUNCH: return "HIP kernel launch failed";
        case MSDA_ERR_BAD_VARIANT: return "requested kernel variant does not support this problem";
        case MSDA_ERR_ALIGNMENT: return "tensor pointer not 16-byte aligned";
        default: return "unknown status";
    }
}

const char *msda_variant_name(int variant)
{
    switch (variant) {
        case MSDA_VARIANT_AUTO: return "auto";
        case MSDA_VARIANT_GENERIC: return "generic";
        case MSDA_VARIANT_QUAD: return "quad";
        case MSDA_VARIANT_WINDOW: return "window";
        case MSDA_VARIANT_DEST: return "dest";
        case MSDA_VARIANT_COARSE: return "coarse";
        case MSDA_VARIANT_CELL: return "cell";
        default: return "?";
    }
}

int msda_check_im2col_step(int batch, int im2col_step)
{
    if (batch <= 0) return MSDA_OK;
    const int step = batch < im2col_step ? batch : im2col_step;   // .cu:50
    if (step <= 0 || batch % step != 0) return MSDA_ERR_IM2COL_STEP;
    return MSDA_OK;
}

int64_t msda_algorithmic_bytes(int dtype, int backward, int N, int S, int M, int D, int L, int Lq, int P)
{
    const int64_t sv = (int64_t)value_elem(dtype), sl = (int64_t)loc_elem(dtype), sg = (int64_t)grad_value_elem(dtype);
    const int64_t v = (int64_t)S * M * D, lo = (int64_t)Lq * M * L * P * 2, a = (int64_t)Lq * M * L * P,
                  o = (int64_t)Lq * M * D;
    if (!backward) return (int64_t)N * (v * sv + lo * sl + a * sl + o * sv);
    return (int64_t)N * (v * (sv + sg) + 2 * lo * sl + 2 * a * sl + o * sv);
}

int msda_pick_variant(int backward, int dtype, int N, int S, int M, int D, int L, int Lq, int P)
{
    Problem p{};
    p.dtype = dtype; p.N = N; p.S = S; p.M = M; p.D = D; p.L = L; p.Lq = Lq; p.P = P;
    return pick(backward != 0, p);
}

int msda_forward_ex(int variant, int dtype, const void *value, const int64_t *spatial_shapes,
                    const int64_t *level_start, const void *sampling_loc, const void *attn_weight, int N, int S,
                    int M, int D, int L, int Lq, int P, void *out, void *stream)
{
    const int st = validate(dtype, N, S, M, D, L, Lq, P);
    if (st != MSDA_OK) return st;
    const long out_elems = (long)N * Lq * M * D;
    if (out_elems == 0) return MSDA_OK;                      // empty output: nothing to do
    if (!out) return MSDA_ERR_NULL_POINTER;
    const long samples = (long)N * Lq * M * L * P;
    if (samples == 0 || (long)N * S == 0) {                  // no samples / empty pyramid -> zeros (at::zeros, .cu:54)
        return hipMemsetAsync(out, 0, out_elems * value_elem(dtype), (hipStream_t)stream) == hipSuccess
                   ? MSDA_OK : MSDA_ERR_LAUNCH;
    }
    if (!value || !spatial_shapes || !level_start || !sampling_loc || !attn_weight) return MSDA_ERR_NULL_POINTER;
    if ((long)N * Lq * M >= (1L << 31) / 64) return MSDA_ERR_BAD_SHAPE;

    Problem p{};
    p.dtype = dtype; p.N = N; p.S = S; p.M = M; p.D = D; p.L = L; p.Lq = Lq; p.P = P;
    p.value = value; p.shapes = spatial_shapes; p.starts = level_start; p.loc = sampling_loc; p.aw = attn_weight;
    p.out = out; p.stream = (hipStream_t)stream;

    int v = variant == MSDA_VARIANT_AUTO ? pick(false, p) : variant;
    if (v != MSDA_VARIANT_GENERIC &&
        !(aligned16(value) && aligned16(sampling_loc) && aligned16(attn_weight) && aligned16(out))) {
        if (variant != MSDA_VARIANT_AUTO) return MSDA_ERR_ALIGNMENT;
        v = MSDA_VARIANT_GENERIC;
    }
    (void)hipGetLastError();
    switch (v) {
        case MSDA_VARIANT_GENERIC: launch_generic_forward(p); break;
        case MSDA_VARIANT_QUAD:
            if (!quad_supports(p)) return MSDA_ERR_BAD_VARIANT;
            launch_quad_forward(p);
            break;
        case MSDA_VARIANT_WINDOW:
            if (!window_supports(p, false)) return MSDA_ERR_BAD_VARIANT;
            launch_window_forward(p);
            break;
        case MSDA_VARIANT_COARSE:
            if (!coarse_forward_applies(p)) return MSDA_ERR_BAD_VARIANT;
            launch_quad_forward_coarse(p);
            break;
        default: return MSDA_ERR_BAD_VARIANT;
    }
    return finish_launch();
}

int msda_backward_ex(int variant, int dtype, const void *value, const int64_t *spatial_shapes,
                     const int64_t *level_start, const void *sampling_loc, const void *attn_weight,
                     const void *grad_out, int N, int S, int M, int D, int L, int Lq, int P, void *grad_value,
                     void *grad_sampling_loc, void *grad_attn_weight, void *stream)
{
    const int st = validate(dtype, N, S, M, D, L, Lq, P);
    if (st != MSDA_OK) return st;
    const bool prezeroed = (variant & MSDA_FLAG_GRAD_VALUE_ZEROED) != 0;
    variant &= 0xff;
    hipStream_t hs = (hipStream_t)stream;
    const long v_elems = (long)N * S * M * D;
    const long samples = (long)N * Lq * M * L * P;
    (void)hipGetLastError();
    if (v_elems > 0) {
        if (!grad_value) return MSDA_ERR_NULL_POINTER;
        if (!prezeroed &&
            hipMemsetAsync(grad_value, 0, v_elems * grad_value_elem(dtype), hs) != hipSuccess) return MSDA_ERR_LAUNCH;
    }
    if (samples == 0) return MSDA_OK;
    if (!grad_sampling_loc || !grad_attn_weight) return MSDA_ERR_NULL_POINTER;
    if (v_elems == 0 || D == 0) {   // samples exist but nothing to sample from: all-zero gradients
        if (hipMemsetAsync(grad_sampling_loc, 0, samples * 2 * loc_elem(dtype), hs) != hipSuccess ||
            hipMemsetAsync(grad_attn_weight, 0, samples * loc_elem(dtype), hs) != hipSuccess)
            return MSDA_ERR_LAUNCH;
        return MSDA_OK;
    }
    if (!value || !spatial_shapes || !level_start || !sampling_loc || !attn_weight || !grad_out)
        return MSDA_ERR_NULL_POINTER;
    if ((long)N * Lq * M >= (1L << 31) / 64) return MSDA_ERR_BAD_SHAPE;

    Problem p{};
    p.dtype = dtype; p.N = N; p.S = S; p.M = M; p.D = D; p.L = L; p.Lq = Lq; p.P = P;
    p.value = value; p.shapes = spatial_shapes; p.starts = level_start; p.loc = sampling_loc; p.aw = attn_weight;
    p.grad_out = grad_out; p.g_value = grad_value; p.g_loc = grad_sampling_loc; p.g_aw = grad_attn_weight;
    p.stream = hs;

    int v = variant == MSDA_VARIANT_AUTO ? pick(true, p) : variant;
    if (v != MSDA_VARIANT_GENERIC &&
        !(aligned16(value) && aligned16(sampling_loc) && aligned16(attn_weight) && aligned16(grad_out) &&
          aligned16(grad_value) && aligned16(grad_sampling_loc) && aligned16(grad_attn_weight))) {
        if (variant != MSDA_VARIANT_AUTO) return MSDA_ERR_ALIGNMENT;
        v = MSDA_VARIANT_GENERIC;
    }
    switch (v) {
        case MSDA_VARIANT_GENERIC: launch_generic_backward(p); break;
        case MSDA_VARIANT_QUAD:
            if (!quad_supports(p)) return MSDA_ERR_BAD_VARIANT;
            launch_quad_backward(p);
            break;
        case MSDA_VARIANT_WINDOW:
            if (!window_supports(p, true)) return MSDA_ERR_BAD_VARIANT;
            launch_window_backward(p);
            break;
        default: return MSDA_ERR_BAD_VARIANT;
    }
    return finish_launch();
}

size_t msda_backward_workspace_bytes(int dtype, const int64_t *spatial_shapes_host, int N, int S, int M, int D, int L,
                                     int Lq, int P)
{
    if (validate(dtype, N, S, M, D, L, Lq, P) != MSDA_OK || !spatial_shapes_host) return 0;
    if ((long)N * S * M * D == 0 || (long)N * Lq * M * L * P == 0) return 0;
    Problem p{};
    p.dtype = dtype; p.N = N; p.S = S; p.M = M; p.D = D; p.L = L; p.Lq = Lq; p.P = P;
    return dest_workspace_bytes(p, spatial_shapes_host);
}

int msda_forward_hs(int variant, int dtype, const void *value, const int64_t *spatial_shapes, const int64_t *level_start,
                    const int64_t *spatial_shapes_host, const void *sampling_loc, const void *attn_weight, int N, int S,
                    int M, int D, int L, int Lq, int P, void *out, void *stream)
{
    if (variant != MSDA_VARIANT_CELL)
        return msda_forward_ex(variant, dtype, value, spatial_shapes, level_start, sampling_loc, attn_weight, N, S, M, D, L,
                               Lq, P, out, stream);
    const int st = validate(dtype, N, S, M, D, L, Lq, P);
    if (st != MSDA_OK) return st;
    if (!spatial_shapes_host) return MSDA_ERR_BAD_VARIANT;
    if (!value || !spatial_shapes || !level_start || !sampling_loc || !attn_weight || !out) return MSDA_ERR_NULL_POINTER;
    if (!(aligned16(value) && aligned16(sampling_loc) && aligned16(attn_weight) && aligned16(out))) return MSDA_ERR_ALIGNMENT;
    Problem p{};
    p.dtype = dtype; p.N = N; p.S = S; p.M = M; p.D = D; p.L = L; p.Lq = Lq; p.P = P;
    p.value = value; p.shapes = spatial_shapes; p.starts = level_start; p.loc = sampling_loc; p.aw = attn_weight;
    p.out = out; p.stream = (hipStream_t)stream;
    if (L > 0 && !dest_shapes_consistent(p, spatial_shapes_host)) return MSDA_ERR_BAD_SHAPE;
    if (!cell_forward_supports(p, spatial_shapes_host)) return MSDA_ERR_BAD_VARIANT;
    (void)hipGetLastError();
    launch_cell_forward(p, spatial_shapes_host, nullptr);
    return hipGetLastError() == hipSuccess ? MSDA_OK : MSDA_ERR_LAUNCH;
}

int msda_backward_plan_info(int dtype, const int64_t *spatial_shapes_host, int N, int S, int M, int D, int L, int Lq, int P,
                            int32_t *out, int out_len)
{
    static_assert(MSDA_PLAN_LEVEL_FIELDS == 14 && MSDA_PLAN_TAIL_FIELDS == 5, "plan layout of patch_plan_info");
    if (validate(dtype, N, S, M, D, L, Lq, P) != MSDA_OK || !spatial_shapes_host) return 0;
    if ((long)N * S * M * D == 0 || (long)N * Lq * M * L * P == 0) return 0;
    Problem p{};
    p.dtype = dtype; p.N = N; p.S = S; p.M = M; p.D = D; p.L = L; p.Lq = Lq; p.P = P;
    return patch_plan_info(p, spatial_shapes_host, out, out_len);
}

int msda_backward_ws(int variant, int dtype, const void *value, const int64_t *spatial_shapes,
                     const int64_t *level_start, const int64_t *spatial_shapes_host, const void *sampling_loc,
                     const void *attn_weight, const void *grad_out, int N, int S, int M, int D, int L, int Lq, int P,
                     void *grad_value, void *grad_sampling_loc, void *grad_attn_weight, void *workspace,
                     size_t workspace_bytes, void *stream)
{
    const int st = validate(dtype, N, S, M, D, L, Lq, P);
    if (st != MSDA_OK) return st;
    const bool out_bf16 = (variant & MSDA_FLAG_GRAD_VALUE_BF16) != 0;
    const int v = variant & 0xff;
    Problem p{};
    p.dtype = dtype; p.N = N; p.S = S; p.M = M; p.D = D; p.L = L; p.Lq = Lq; p.P = P;
    p.value = value; p.shapes = spatial_shapes; p.starts = level_start; p.loc = sampling_loc; p.aw = attn_weight;
    p.grad_out = grad_out; p.g_value = grad_value; p.g_loc = grad_sampling_loc; p.g_aw = grad_attn_weight;
    p.stream = (hipStream_t)stream;
    if (spatial_shapes_host && L > 0 && !dest_shapes_consistent(p, spatial_shapes_host)) return MSDA_ERR_BAD_SHAPE;
    const bool nonempty = (long)N * S * M * D > 0 && (long)N * Lq * M * L * P > 0;
    if (!nonempty) {   // degenerate problems: zero-filled gradients (bfloat16 grad_value: half the bytes)
        const long v_elems = (long)N * S * M * D, samples = (long)N * Lq * M * L * P;
        hipStream_t hs = (hipStream_t)stream;
        if (v_elems > 0 && (!grad_value || hipMemsetAsync(grad_value, 0, v_elems * (out_bf16 ? 2 : grad_value_elem(dtype)), hs) != hipSuccess))
            return grad_value ? MSDA_ERR_LAUNCH : MSDA_ERR_NULL_POINTER;
        if (samples > 0) {
            if (!grad_sampling_loc || !grad_attn_weight) return MSDA_ERR_NULL_POINTER;
            if (hipMemsetAsync(grad_sampling_loc, 0, samples * 2 * loc_elem(dtype), hs) != hipSuccess ||
                hipMemsetAsync(grad_attn_weight, 0, samples * loc_elem(dtype), hs) != hipSuccess)
                return MSDA_ERR_LAUNCH;
        }
        return MSDA_OK;
    }
    const size_t need = nonempty && spatial_shapes_host ? dest_workspace_bytes(p, spatial_shapes_host) : 0;
    const bool can = need > 0 && workspace && workspace_bytes >= need &&
                     aligned16(value) && aligned16(sampling_loc) && aligned16(attn_weight) && aligned16(grad_out) &&
                     aligned16(grad_value) && aligned16(grad_sampling_loc) && aligned16(grad_attn_weight) &&
                     aligned16(workspace);
    if ((v == MSDA_VARIANT_AUTO || v == MSDA_VARIANT_DEST) && can) {
        if (!value || !spatial_shapes || !level_start || !sampling_loc || !attn_weight || !grad_out || !grad_value ||
            !grad_sampling_loc || !grad_attn_weight)
            return MSDA_ERR_NULL_POINTER;
        if (out_bf16 && dtype != MSDA_BF16) return MSDA_ERR_BAD_VARIANT;
        (void)hipGetLastError();
        // (K1 forked onto a second stream beside the grad_value pass was measured: 678 vs 633 us -- K1's 44 k
        //  workgroups fill the chip first, nothing overlaps)
        // grad_sampling_loc / grad_attn_weight (K1) + grad_value (every row written once)
        launch_backward_dest(p, nullptr, spatial_shapes_host, workspace, out_bf16);
        return finish_launch();
    }
    if (v == MSDA_VARIANT_DEST || out_bf16) return MSDA_ERR_BAD_VARIANT;
    return msda_backward_ex(variant, dtype, value, spatial_shapes, level_start, sampling_loc, attn_weight, grad_out, N, S,
                            M, D, L, Lq, P, grad_value, grad_sampling_loc, grad_attn_weight, stream);
}

int msda_fused_supported(int dtype, const int64_t *spatial_shapes_host, int refdim, int N, int S, int M, int D, int L,
                         int Lq, int P)
{
    if (validate(dtype, N, S, M, D, L, Lq, P) != MSDA_OK) return 0;
    if (dtype != MSDA_F32 && dtype != MSDA_BF16) return 0;
    if (refdim != 2 && refdim != 4) return 0;
    if ((long)N * S * M * D == 0 || (long)N * Lq * M * L * P == 0) return 0;
    Problem p{};
    p.dtype = dtype; p.N = N; p.S = S; p.M = M; p.D = D; p.L = L; p.Lq = Lq; p.P = P;
    if (!quad_supports(p)) return 0;
    if (!spatial_shapes_host) return 1;                     // forward only
    return dest_supports(p, spatial_shapes_host) ? 2 : 1;   // 2: the fused backward is available as well
}

int msda_fused_forward(int dtype, const void *value, const int64_t *spatial_shapes, const int64_t *level_start,
                       const void *qproj, const float *ref, int refdim, int N, int S, int M, int D, int L, int Lq, int P,
                       void *out, float *loc_save, float *aw_save, void *stream)
{
    if (!msda_fused_supported(dtype, nullptr, refdim, N, S, M, D, L, Lq, P)) return MSDA_ERR_BAD_VARIANT;
    if (!value || !spatial_shapes || !level_start || !qproj || !ref || !out) return MSDA_ERR_NULL_POINTER;
    if ((loc_save == nullptr) != (aw_save == nullptr)) return MSDA_ERR_NULL_POINTER;
    if ((long)N * Lq * M >= (1L << 31) / 64) return MSDA_ERR_BAD_SHAPE;
    if (!(aligned16(value) && aligned16(qproj) && aligned16(ref) && aligned16(out) && aligned16(loc_save) &&
          aligned16(aw_save)))
        return MSDA_ERR_ALIGNMENT;
    Problem p{};
    p.dtype = dtype; p.N = N; p.S = S; p.M = M; p.D = D; p.L = L; p.Lq = Lq; p.P = P;
    p.value = value; p.shapes = spatial_shapes; p.starts = level_start; p.out = out; p.stream = (hipStream_t)stream;
    Fused f{};
    f.qproj = qproj; f.ref = ref; f.refdim = refdim; f.loc_save = loc_save; f.aw_save = aw_save;
    (void)hipGetLastError();
    launch_quad_forward_fused(p, f);
    return finish_launch();
}

int msda_fused_forward_hs(int variant, int dtype, const void *value, const int64_t *spatial_shapes, const int64_t *level_start,
                          const int64_t *spatial_shapes_host, const void *qproj, const float *ref, int refdim, int N, int S,
                          int M, int D, int L, int Lq, int P, void *out, float *loc_save, float *aw_save, void *stream)
{
    if (variant != MSDA_VARIANT_CELL)
        return msda_fused_forward(dtype, value, spatial_shapes, level_start, qproj, ref, refdim, N, S, M, D, L, Lq, P, out,
                                  loc_save, aw_save, stream);
    if (!msda_fused_supported(dtype, nullptr, refdim, N, S, M, D, L, Lq, P) || !spatial_shapes_host) return MSDA_ERR_BAD_VARIANT;
    if (!value || !spatial_shapes || !level_start || !qproj || !ref || !out || !loc_save || !aw_save) return MSDA_ERR_NULL_POINTER;
    if (!(aligned16(value) && aligned16(qproj) && aligned16(ref) && aligned16(out) && aligned16(loc_save) && aligned16(aw_save)))
        return MSDA_ERR_ALIGNMENT;
    Problem p{};
    p.dtype = dtype; p.N = N; p.S = S; p.M = M; p.D = D; p.L = L; p.Lq = Lq; p.P = P;
    p.value = value; p.shapes = spatial_shapes; p.starts = level_start; p.out = out; p.stream = (hipStream_t)stream;
    if (!dest_shapes_consistent(p, spatial_shapes_host)) return MSDA_ERR_BAD_SHAPE;
    if (!cell_forward_supports(p, spatial_shapes_host)) return MSDA_ERR_BAD_VARIANT;
    Fused f{};
    f.qproj = qproj; f.ref = ref; f.refdim = refdim; f.loc_save = loc_save; f.aw_save = aw_save;
    (void)hipGetLastError();
    launch_cell_forward(p, spatial_shapes_host, &f);
    return finish_launch();
}

int msda_fused_backward_ws(int flags, int dtype, const void *value, const int64_t *spatial_shapes,
                           const int64_t *level_start, const int64_t *spatial_shapes_host, const void *sampling_loc,
                           const void *attn_weight, const float *ref, int refdim, const void *grad_out, int N, int S,
                           int M, int D, int L, int Lq, int P, void *grad_value, void *grad_qproj, void *workspace,
                           size_t workspace_bytes, void *stream)
{
    if (msda_fused_supported(dtype, spatial_shapes_host, refdim, N, S, M, D, L, Lq, P) != 2) return MSDA_ERR_BAD_VARIANT;
    const bool out_bf16 = (flags & MSDA_FLAG_GRAD_VALUE_BF16) != 0;
    if (out_bf16 && dtype != MSDA_BF16) return MSDA_ERR_BAD_VARIANT;
    if (!value || !spatial_shapes || !level_start || !sampling_loc || !attn_weight || !ref || !grad_out || !grad_value ||
        !grad_qproj || !workspace)
        return MSDA_ERR_NULL_POINTER;
    Problem p{};
    p.dtype = dtype; p.N = N; p.S = S; p.M = M; p.D = D; p.L = L; p.Lq = Lq; p.P = P;
    p.value = value; p.shapes = spatial_shapes; p.starts = level_start; p.loc = sampling_loc; p.aw = attn_weight;
    p.grad_out = grad_out; p.g_value = grad_value; p.stream = (hipStream_t)stream;
    if (workspace_bytes < dest_workspace_bytes(p, spatial_shapes_host)) return MSDA_ERR_BAD_SHAPE;
    if (!(aligned16(value) && aligned16(sampling_loc) && aligned16(attn_weight) && aligned16(grad_out) &&
          aligned16(grad_value) && aligned16(grad_qproj) && aligned16(ref) && aligned16(workspace)))
        return MSDA_ERR_ALIGNMENT;
    Fused f{};
    f.ref = ref; f.refdim = refdim; f.g_qproj = grad_qproj;
    (void)hipGetLastError();
    launch_backward_dest(p, &f, spatial_shapes_host, workspace, out_bf16);   // grad of the projection row + grad_value
    return finish_launch();
}

size_t msda_records_bytes(int dtype, const int64_t *spatial_shapes_host, int N, int S, int M, int D, int L, int Lq, int P)
{
    if (validate(dtype, N, S, M, D, L, Lq, P) != MSDA_OK || !spatial_shapes_host) return 0;
    if ((long)N * S * M * D == 0 || (long)N * Lq * M * L * P == 0) return 0;
    Problem p{};
    p.dtype = dtype; p.N = N; p.S = S; p.M = M; p.D = D; p.L = L; p.Lq = Lq; p.P = P;
    p.value = &p;                                   // (cell_forward_supports wants a value pointer; only the shape matters here)
    if (!dest_shapes_consistent(p, spatial_shapes_host) || !dest_supports(p, spatial_shapes_host)) return 0;
    return cell_records_bytes(p, spatial_shapes_host);
}

int msda_records_forward(int dtype, const void *value, const int64_t *spatial_shapes, const int64_t *level_start,
                         const int64_t *spatial_shapes_host, const void *qproj, const float *ref, int refdim,
                         float *sampling_loc, float *attn_weight, int N, int S, int M, int D, int L, int Lq, int P, void *out,
                         void *records, size_t records_bytes, void *stream)
{
    const int st = validate(dtype, N, S, M, D, L, Lq, P);
    if (st != MSDA_OK) return st;
    if (refdim != 0 && refdim != 2 && refdim != 4) return MSDA_ERR_BAD_VARIANT;
    const size_t need = msda_records_bytes(dtype, spatial_shapes_host, N, S, M, D, L, Lq, P);
    if (need == 0) return MSDA_ERR_BAD_VARIANT;
    if (!value || !spatial_shapes || !level_start || !out || !records) return MSDA_ERR_NULL_POINTER;
    if (refdim != 0 && (!qproj || !ref)) return MSDA_ERR_NULL_POINTER;
    // (module operands: both NULL = the records are the whole saved state, no float32 locations / weights are written)
    if (refdim == 0 ? (!sampling_loc || !attn_weight) : ((sampling_loc == nullptr) != (attn_weight == nullptr))) return MSDA_ERR_NULL_POINTER;
    if (records_bytes < need) return MSDA_ERR_BAD_SHAPE;
    if (!(aligned16(value) && aligned16(sampling_loc) && aligned16(attn_weight) && aligned16(out) && aligned16(records) &&
          aligned16(qproj) && aligned16(ref)))
        return MSDA_ERR_ALIGNMENT;
    Problem p{};
    p.dtype = dtype; p.N = N; p.S = S; p.M = M; p.D = D; p.L = L; p.Lq = Lq; p.P = P;
    p.value = value; p.shapes = spatial_shapes; p.starts = level_start; p.loc = sampling_loc; p.aw = attn_weight;
    p.out = out; p.stream = (hipStream_t)stream;
    Fused f{};
    f.qproj = qproj; f.ref = ref; f.refdim = refdim; f.loc_save = sampling_loc; f.aw_save = attn_weight;
    (void)hipGetLastError();
    launch_cell_forward(p, spatial_shapes_host, refdim ? &f : nullptr, records);
    return finish_launch();
}

int msda_records_backward(int flags, int dtype, const void *value, const int64_t *spatial_shapes, const int64_t *level_start,
                          const int64_t *spatial_shapes_host, const float *sampling_loc, const float *attn_weight,
                          const float *ref, int refdim, const void *grad_out, int N, int S, int M, int D, int L, int Lq, int P,
                          void *grad_value, void *grad_sampling_loc, void *grad_attn_weight, void *grad_qproj,
                          const void *records, size_t records_bytes, void *workspace, size_t workspace_bytes, void *stream)
{
    const int st = validate(dtype, N, S, M, D, L, Lq, P);
    if (st != MSDA_OK) return st;
    if (refdim != 0 && refdim != 2 && refdim != 4) return MSDA_ERR_BAD_VARIANT;
    const size_t need = msda_records_bytes(dtype, spatial_shapes_host, N, S, M, D, L, Lq, P);
    if (need == 0) return MSDA_ERR_BAD_VARIANT;
    const bool out_bf16 = (flags & MSDA_FLAG_GRAD_VALUE_BF16) != 0;
    if (!value || !spatial_shapes || !level_start || !grad_out || !grad_value || !records || !workspace) return MSDA_ERR_NULL_POINTER;
    if (refdim == 0 ? (!grad_sampling_loc || !grad_attn_weight || !sampling_loc || !attn_weight) : (!ref || !grad_qproj))
        return MSDA_ERR_NULL_POINTER;
    if ((sampling_loc == nullptr) != (attn_weight == nullptr)) return MSDA_ERR_NULL_POINTER;
    Problem p{};
    p.dtype = dtype; p.N = N; p.S = S; p.M = M; p.D = D; p.L = L; p.Lq = Lq; p.P = P;
    p.value = value; p.shapes = spatial_shapes; p.starts = level_start; p.loc = sampling_loc; p.aw = attn_weight;
    p.grad_out = grad_out; p.g_value = grad_value; p.g_loc = grad_sampling_loc; p.g_aw = grad_attn_weight;
    p.stream = (hipStream_t)stream;
    if (records_bytes < need || workspace_bytes < dest_workspace_bytes(p, spatial_shapes_host)) return MSDA_ERR_BAD_SHAPE;
    if (!(aligned16(value) && aligned16(sampling_loc) && aligned16(attn_weight) && aligned16(grad_out) && aligned16(grad_value) &&
          aligned16(grad_sampling_loc) && aligned16(grad_attn_weight) && aligned16(grad_qproj) && aligned16(ref) &&
          aligned16(records) && aligned16(workspace)))
        return MSDA_ERR_ALIGNMENT;
    Fused f{};
    f.ref = ref; f.refdim = refdim; f.g_qproj = grad_qproj;
    (void)hipGetLastError();
    launch_backward_dest(p, refdim ? &f : nullptr, spatial_shapes_host, workspace, out_bf16, records,
                         (flags & MSDA_FLAG_RECORDS_SWAP) != 0);
    return finish_launch();
}

int msda_forward(int dtype, const void *value, const int64_t *spatial_shapes, const int64_t *level_start,
                 const void *sampling_loc, const void *attn_weight, int N, int S, int M, int D, int L, int Lq, int P,
                 void *out, void *stream)
{
    return msda_forward_ex(MSDA_VARIANT_AUTO, dtype, value, spatial_shapes, level_start, sampling_loc, attn_weight,
                           N, S, M, D, L, Lq, P, out, stream);
}

int msda_backward(int dtype, const void *value, const int64_t *spatial_shapes, const int64_t *level_start,
                  const void *sampling_loc, const void *attn_weight, const void *grad_out, int N, int S, int M,
                  int D, int L, int Lq, int P, void *grad_value, void *grad_sampling_loc, void *grad_attn_weight,
                  void *stream)
{
    return msda_backward_ex(MSDA_VARIANT_AUTO, dtype, value, spatial_shapes, level_start, sampling_loc, attn_weight,
                            grad_out, N, S, M, D, L, Lq, P, grad_value, grad_sampling_loc, grad_attn_weight, stream);
}

}  // extern "C"
