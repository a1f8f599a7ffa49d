/*
 * rlipv2_msda.h -- C ABI of the MI355X-native multi-scale deformable attention library
 * (librlipv2_msda.so, built from rlipv2_amd/csrc/ for gfx950).
 *
 * This is the drop-in boundary for the reference's native extension module
 * "MultiScaleDeformableAttention" (reference: models/ops/src/vision.cpp:13-16, dispatch
 * models/ops/src/ms_deform_attn.h:36-77, host launchers
 * models/ops/src/cuda/ms_deform_attn_cuda.cu:20-80 and :83-153; the byte-identical twin lives
 * under models/dab_deformable/ops/src/).  The reference binds its kernels through
 * pybind/ATen; the entry points below carry the same operands as plain pointers and sizes
 * so that any host (Python ctypes, C++, a torch extension) can bind them.  INTEGRATION.md
 * shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions shared by every entry point
 * ---------------------------------------
 *  - All data pointers are DEVICE pointers on the current HIP device, contiguous row-major:
 *      value           [N, S, M, D]        S = sum_l H_l * W_l, level-major, then (h, w)
 *      spatial_shapes  int64 [L, 2]        (H_l, W_l)            -- on device, as in the
 *      level_start     int64 [L]           first row of level l     reference (.cu:67-68)
 *      sampling_loc    [N, Lq, M, L, P, 2] (x, y), normalised to the level extent
 *      attn_weight     [N, Lq, M, L, P]
 *      out / grad_out  [N, Lq, M*D]
 *  - `dtype` selects the element type of value / out / grad_out:
 *      MSDA_F32, MSDA_F64 : every tensor has that type (the reference's two types,
 *                           AT_DISPATCH_FLOATING_TYPES at .cu:64).
 *      MSDA_BF16          : value / out / grad_out are bfloat16; sampling_loc, attn_weight
 *                           and ALL gradients (grad_value included) are float32; the
 *                           accumulation is float32.  (New in this build; the reference has no
 *                           16-bit path.)
 *  - Inputs are borrowed; outputs are caller-allocated.  Nothing is allocated, freed or
 *    synchronised inside; work is enqueued on `stream` (a hipStream_t; NULL = the default
 *    stream) and the call returns immediately, like the reference's launch on
 *    at::cuda::getCurrentCUDAStream() (.cu:65).  Re-entrant: no global scratch state.
 *  - Return value: MSDA_OK (0) or a negative msda_status; msda_strerror() names it.  Unlike
 *    the reference, which only printf()s a failed launch (ms_deform_im2col_cuda.cuh:948-952),
 *    a launch failure is reported to the caller.
 *  - The reference's `im2col_step` argument only chunks the batch loop on the host
 *    (.cu:50-61) and does not change results; its divisibility precondition
 *    (batch % min(batch, im2col_step) == 0, .cu:50-52) is checked by msda_check_im2col_step
 *    so bindings can reproduce the reference's error.
 */
#ifndef RLIPV2_MSDA_H
#define RLIPV2_MSDA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RLIPV2_MSDA_ABI_VERSION 1

typedef enum msda_dtype {
    MSDA_F32 = 0,
    MSDA_F64 = 1,
    MSDA_BF16 = 2
} msda_dtype;

typedef enum msda_status {
    MSDA_OK = 0,
    MSDA_ERR_BAD_DTYPE = -1,
    MSDA_ERR_BAD_SHAPE = -2,    /* a dimension <= 0 (empty tensors are MSDA_OK no-ops), or too large */
    MSDA_ERR_NULL_POINTER = -3,
    MSDA_ERR_IM2COL_STEP = -4,  /* batch % min(batch, im2col_step) != 0 */
    MSDA_ERR_LAUNCH = -5,       /* hipGetLastError() != hipSuccess after a launch */
    MSDA_ERR_BAD_VARIANT = -6,
    MSDA_ERR_ALIGNMENT = -7     /* a pointer is not aligned for the vector width the fast path needs */
} msda_status;

/* Kernel selection.  MSDA_VARIANT_AUTO is what product code uses; the others exist so the
 * benchmark and the parity tests can pin one implementation. */
typedef enum msda_variant {
    MSDA_VARIANT_AUTO = 0,
    MSDA_VARIANT_GENERIC = 1,   /* any M, D, L, P; f32 / f64 / bf16: one wave per (n, q, m) */
    MSDA_VARIANT_QUAD = 2,      /* D = 32, L*P = 16: four lanes per (n, q, m), direct gathers */
    MSDA_VARIANT_WINDOW = 3,    /* D = 32, L*P = 16: LDS-staged sampling windows per query tile */
    MSDA_VARIANT_DEST = 4,      /* backward only, D = 32, L*P = 16: destination-stationary grad_value (no float atomics,
                                   deterministic); needs a workspace and a host copy of spatial_shapes: msda_backward_ws */
    MSDA_VARIANT_COARSE = 5,    /* forward only, bfloat16, D = 32, L*P = 16, Lq >= 4096: direct gathers for the fine levels,
                                   the rows of the coarse levels resident in LDS per (image, head) workgroup */
    MSDA_VARIANT_CELL = 6       /* forward only, through msda_forward_hs only (needs the host copy of the level shapes),
                                   bfloat16, D = 32, L = P = 4, Lq == S: per-cell sampling windows in LDS, bilinear sums
                                   on the matrix cores (csrc/msda_patch.hip: cell_forward_kernel).  EXPERIMENTAL: never
                                   picked by MSDA_VARIANT_AUTO */
} msda_variant;

/* Replaces ms_deform_attn_forward (reference models/ops/src/ms_deform_attn.h:36-53,
 * cuda/ms_deform_attn_cuda.cu:20-80).  Writes every element of out. */
int msda_forward(int dtype,
                 const void *value, const int64_t *spatial_shapes, const int64_t *level_start,
                 const void *sampling_loc, const void *attn_weight,
                 int N, int S, int M, int D, int L, int Lq, int P,
                 void *out, void *stream);

/* Replaces ms_deform_attn_backward (reference models/ops/src/ms_deform_attn.h:56-77,
 * cuda/ms_deform_attn_cuda.cu:83-153).  grad_value is zero-filled INSIDE the call (an async
 * memset on `stream`, the reference's at::zeros_like at .cu:121) and then accumulated into;
 * grad_sampling_loc and grad_attn_weight are fully overwritten.  For MSDA_BF16, grad_value is
 * float32 [N, S, M, D]. */
int msda_backward(int dtype,
                  const void *value, const int64_t *spatial_shapes, const int64_t *level_start,
                  const void *sampling_loc, const void *attn_weight, const void *grad_out,
                  int N, int S, int M, int D, int L, int Lq, int P,
                  void *grad_value, void *grad_sampling_loc, void *grad_attn_weight,
                  void *stream);

/* msda_forward_ex with a HOST copy of spatial_shapes (int64 [L, 2], same values as the device tensor; NULL: exactly
 * msda_forward_ex).  Variants whose launch geometry depends on the pyramid need it: MSDA_VARIANT_CELL (returns
 * MSDA_ERR_BAD_VARIANT without it or when the problem is not a bfloat16 encoder call with Lq == S; MSDA_ERR_BAD_SHAPE
 * when sum(H_l * W_l) != S). */
int msda_forward_hs(int variant, int dtype,
                    const void *value, const int64_t *spatial_shapes, const int64_t *level_start,
                    const int64_t *spatial_shapes_host,
                    const void *sampling_loc, const void *attn_weight,
                    int N, int S, int M, int D, int L, int Lq, int P,
                    void *out, void *stream);

/* Flag OR-ed into the `variant` argument of msda_backward_ex: the caller has already zero-filled
 * grad_value on `stream`, skip the memset inside (lets a profiler time the kernel alone). */
#define MSDA_FLAG_GRAD_VALUE_ZEROED 0x100

/* Flag for msda_backward_ws with MSDA_BF16 and the DEST variant: grad_value is written as bfloat16 [N, S, M, D]
 * (every row has exactly one writer there, so the float32 staging tensor and its cast are not needed). */
#define MSDA_FLAG_GRAD_VALUE_BF16 0x200

/* Same as the two calls above with an explicit kernel choice (msda_variant | flags). */
int msda_forward_ex(int variant, int dtype,
                    const void *value, const int64_t *spatial_shapes, const int64_t *level_start,
                    const void *sampling_loc, const void *attn_weight,
                    int N, int S, int M, int D, int L, int Lq, int P,
                    void *out, void *stream);
int msda_backward_ex(int variant, int dtype,
                     const void *value, const int64_t *spatial_shapes, const int64_t *level_start,
                     const void *sampling_loc, const void *attn_weight, const void *grad_out,
                     int N, int S, int M, int D, int L, int Lq, int P,
                     void *grad_value, void *grad_sampling_loc, void *grad_attn_weight,
                     void *stream);

/* Backward with a caller-provided workspace and a HOST copy of spatial_shapes (int64 [L, 2], same values as the
 * device tensor).  The host copy lets the library size its grids and workspace without a device sync and check
 * sum(H_l * W_l) == S -- the assert of the reference module (models/ops/modules/ms_deform_attn.py:96), which the
 * reference pays a sync for; a mismatch returns MSDA_ERR_BAD_SHAPE.  With it MSDA_VARIANT_AUTO picks the
 * destination-stationary grad_value pass (MSDA_VARIANT_DEST): no float atomics, no zero-fill, results repeatable bit
 * for bit.  spatial_shapes_host == NULL or workspace_bytes too small: falls back to msda_backward_ex (AUTO) or returns
 * MSDA_ERR_BAD_VARIANT (explicit DEST / bfloat16 grad_value).  msda_backward_workspace_bytes returns 0 when the DEST
 * variant does not support the problem.  The workspace is scratch: its contents need not survive the call, and
 * concurrent calls on different streams need different workspaces. */
size_t msda_backward_workspace_bytes(int dtype, const int64_t *spatial_shapes_host,
                                     int N, int S, int M, int D, int L, int Lq, int P);
/* Diagnostics (host only, no device work): the launch plan of the encoder backward's cell + patch route for a pyramid
 * -- for every level H, W, patches (rows, columns), neighbourhood radius and extent in cells, reciprocal of the extent,
 * first mask slot, waves per patch, patches per wave, first workgroup item and item count; then cells (rows, columns),
 * mask slots per (image, head), items per (image, head), bytes of the binning table.  `out` receives
 * 4 * MSDA_PLAN_LEVEL_FIELDS + MSDA_PLAN_TAIL_FIELDS int32 values, level-major.  Returns the count written, 0 when the
 * problem does not take that route (then msda_backward_ws uses the sorting pass), -1 when out_len is too small. */
#define MSDA_PLAN_LEVEL_FIELDS 14
#define MSDA_PLAN_TAIL_FIELDS 5
int msda_backward_plan_info(int dtype, const int64_t *spatial_shapes_host,
                            int N, int S, int M, int D, int L, int Lq, int P, int32_t *out, int out_len);
int msda_backward_ws(int variant, int dtype,
                     const void *value, const int64_t *spatial_shapes, const int64_t *level_start,
                     const int64_t *spatial_shapes_host,
                     const void *sampling_loc, const void *attn_weight, const void *grad_out,
                     int N, int S, int M, int D, int L, int Lq, int P,
                     void *grad_value, void *grad_sampling_loc, void *grad_attn_weight,
                     void *workspace, size_t workspace_bytes, void *stream);

/* Fused "sampling geometry" of the MSDeformAttn module (reference models/ops/modules/ms_deform_attn.py:101-112:
 * view + softmax over the L*P logits + offsets/(W,H) or offsets/P*wh*0.5 + reference point), L = 4, P = 4.
 *   qproj [R, M*L*P*3] (qdtype MSDA_F32 or MSDA_BF16): M*L*P*2 offsets then M*L*P logits, R = N*Lq rows
 *   ref   [R, L, refdim] float32, refdim 2 (points) or 4 (boxes)
 *   -> sampling_loc [R, M, L, P, 2], attn_weight [R, M, L, P], float32.
 * Backward: gradients of the projection row (qdtype) and, if g_ref != NULL, of the reference points
 * (float32, zero-filled inside). */
int msda_prepare_forward(int qdtype, const void *qproj, const float *ref, int refdim, const int64_t *spatial_shapes,
                         int R, int M, int L, int P, float *sampling_loc, float *attn_weight, void *stream);
int msda_prepare_backward(int qdtype, const void *qproj, const float *ref, int refdim, const int64_t *spatial_shapes,
                          const float *attn_weight, const float *grad_sampling_loc, const float *grad_attn_weight,
                          int R, int M, int L, int P, void *grad_qproj, float *grad_ref, void *stream);

/* MSDeformAttn's sampling geometry AND the sampling + aggregation in one launch each way (reference
 * models/ops/modules/ms_deform_attn.py:101-117: the view / softmax / normalise / add chain feeding
 * MSDeformAttnFunction.apply): the forward reads the raw projection rows qproj [N*Lq, M*L*P*3] (value's dtype;
 * offsets then logits, the layout of msda_prepare_forward) and ref [N*Lq, L, refdim] instead of float32
 * sampling_loc / attn_weight; loc_save / aw_save (both or neither) receive the float32 locations / weights for a
 * later backward pass.  The backward takes those two tensors back and writes grad_value and the gradient of qproj
 * (value's dtype) -- grad_sampling_loc / grad_attn_weight never reach memory; the reference points get no gradient
 * here (callers that need one use msda_prepare_backward).  D = 32, L = P = 4, MSDA_F32 / MSDA_BF16.
 * msda_fused_supported: 0 = no, 1 = forward only, 2 = forward and backward (needs the host copy of the shapes:
 * grad_value comes from the destination-stationary pass, workspace of msda_backward_workspace_bytes). */
int msda_fused_supported(int dtype, const int64_t *spatial_shapes_host, int refdim,
                         int N, int S, int M, int D, int L, int Lq, int P);
int msda_fused_forward(int dtype,
                       const void *value, const int64_t *spatial_shapes, const int64_t *level_start,
                       const void *qproj, const float *ref, int refdim,
                       int N, int S, int M, int D, int L, int Lq, int P,
                       void *out, float *loc_save, float *aw_save, void *stream);
/* msda_fused_forward with an explicit kernel choice and the host copy of the level shapes: MSDA_VARIANT_AUTO is
 * msda_fused_forward; MSDA_VARIANT_CELL (experimental, see the enum) needs loc_save / aw_save -- it is the train step's
 * forward, whose saved locations / weights the backward pass reads -- and a bfloat16 encoder call (Lq == S). */
int msda_fused_forward_hs(int variant, int dtype,
                          const void *value, const int64_t *spatial_shapes, const int64_t *level_start,
                          const int64_t *spatial_shapes_host,
                          const void *qproj, const float *ref, int refdim,
                          int N, int S, int M, int D, int L, int Lq, int P,
                          void *out, float *loc_save, float *aw_save, void *stream);
int msda_fused_backward_ws(int flags, int dtype,
                           const void *value, const int64_t *spatial_shapes, const int64_t *level_start,
                           const int64_t *spatial_shapes_host,
                           const void *sampling_loc, const void *attn_weight, const float *ref, int refdim,
                           const void *grad_out,
                           int N, int S, int M, int D, int L, int Lq, int P,
                           void *grad_value, void *grad_qproj,
                           void *workspace, size_t workspace_bytes, void *stream);

/* ---- the "records" route of a bfloat16 encoder call (Lq == S, D = 32, L = P = 4) -------------------------------------------
 * (csrc/msda_cell_forward.inc with EMIT, csrc/msda_cell_records.inc; rounds 5-6, written without a GPU: explicit entry points,
 *  never picked by the AUTO variants.)  The forward pass (MSDA_VARIANT_CELL's kernel) leaves, in ONE caller-owned buffer of
 * msda_records_bytes, what the backward pass would otherwise recompute from float32 sampling_loc / attn_weight: a 2-byte
 * record per sample (the LDS-window pixel of its top-left corner), the LDS window table of every (image, head, cell) and the
 * patch masks + 48-byte group records (locations and weights of a (query, head, level)) of the grad_value pass -- 222 MB per
 * N = 4 call at 800 x 1333 (round 5's 16-byte records: 408 MB; the product route saves 136 MB of float32 locations / weights
 * and moves another 194 MB of masks + group records inside its backward).  The backward pass then runs no bounding boxes, no
 * window placement, no corner clipping and no binning (per sample: the two bilinear fractions from the group record's location,
 * six instructions): gradients of the locations / weights (reference ms_deform_im2col_cuda.cuh:87-159) from the
 * records on v_mfma_f32_4x4x4_16B_bf16, grad_value from the matrix-core patch pass.  Against msda_backward_ws /
 * msda_fused_backward_ws on the same call: grad_value bit-identical (the same patch pass on the same masks and group records);
 * the other gradients bit-identical on the lane-level model of tools/emu/ and, on the device, within a rounding of their type
 * (the compiler contracts the reference's float32 formulas into FMAs per kernel: profiles/r05_records_route_static.txt).
 *   refdim 0: the op's signature -- sampling_loc / attn_weight are INPUTS of both calls (qproj, ref, grad_qproj NULL);
 *   refdim 2 / 4: the module's operands (msda_fused_forward) -- sampling_loc / attn_weight are OUTPUTS of the forward and
 *                 inputs of the backward, or both NULL in both calls: the records then are the whole saved state (the group
 *                 records hold the same floats; the sorting fallback for "far" samples rebuilds them inside the workspace);
 *                 grad_sampling_loc / grad_attn_weight NULL.
 * msda_records_bytes: 0 = the route does not take this call.  flags of the backward: MSDA_FLAG_GRAD_VALUE_BF16 (required:
 * the patch pass writes bfloat16 or float32 rows; pass it for bfloat16 grad_value), MSDA_FLAG_RECORDS_SWAP (operand order of
 * the 4x4x4 products, an experiment arm).  workspace: msda_backward_workspace_bytes, as for msda_backward_ws. */
#define MSDA_FLAG_RECORDS_SWAP 0x400
size_t msda_records_bytes(int dtype, const int64_t *spatial_shapes_host, int N, int S, int M, int D, int L, int Lq, int P);
int msda_records_forward(int dtype,
                         const void *value, const int64_t *spatial_shapes, const int64_t *level_start,
                         const int64_t *spatial_shapes_host,
                         const void *qproj, const float *ref, int refdim,
                         float *sampling_loc, float *attn_weight,
                         int N, int S, int M, int D, int L, int Lq, int P,
                         void *out, void *records, size_t records_bytes, void *stream);
int msda_records_backward(int flags, int dtype,
                          const void *value, const int64_t *spatial_shapes, const int64_t *level_start,
                          const int64_t *spatial_shapes_host,
                          const float *sampling_loc, const float *attn_weight, const float *ref, int refdim,
                          const void *grad_out,
                          int N, int S, int M, int D, int L, int Lq, int P,
                          void *grad_value, void *grad_sampling_loc, void *grad_attn_weight, void *grad_qproj,
                          const void *records, size_t records_bytes,
                          void *workspace, size_t workspace_bytes, void *stream);

/* The reference's batch-chunking precondition (cuda/ms_deform_attn_cuda.cu:50-52). */
int msda_check_im2col_step(int batch, int im2col_step);

/* Algorithmic bytes of one call (each tensor once, metadata ignored):
 * forward  N*[S*M*D*sv + Lq*M*L*P*2*sl + Lq*M*L*P*sl + Lq*M*D*sv]
 * backward N*[S*M*D*(sv+sg) + 2*Lq*M*L*P*2*sl + 2*Lq*M*L*P*sl + Lq*M*D*sv]
 * with sv = sizeof(value element), sl = sizeof(loc element), sg = sizeof(grad_value element). */
int64_t msda_algorithmic_bytes(int dtype, int backward, int N, int S, int M, int D, int L, int Lq, int P);

const char *msda_strerror(int status);
int msda_abi_version(void);
/* Name of the kernel msda_{forward,backward} would pick for this problem ("generic", "quad", ...). */
const char *msda_variant_name(int variant);
int msda_pick_variant(int backward, int dtype, int N, int S, int M, int D, int L, int Lq, int P);

/* ---- one head of 256 channels, few queries (csrc/msda_rows.hip; round 5, not yet run on hardware) --------------------------
 * Backward of msda_forward called with M = 1, D = 256: `value` = an unprojected memory [N, S, 256], the queries = (query, head)
 * pairs (rlipv2_amd/deform_attn.py: MSDeformAttn._sampled_projection -- the decoders' cross-attention of
 * models/ops/modules/ms_deform_attn.py:98-118 with sampling and value projection exchanged).  Same gradients as
 * msda_backward (formulas: ms_deform_im2col_cuda.cuh:87-159) without float atomics: every row of grad_src is written exactly
 * once (no zero-fill needed), sums in sample order (bit-repeatable).  dtype MSDA_F32 or MSDA_BF16 (src / grad_out / grad_src);
 * sampling_loc [N, Q, L, P, 2], attn_weight [N, Q, L, P] and their gradients float32.  shapes_host: host copy of the int64
 * [L, 2] level shapes (the launch plan is built from it); level_start on the device as everywhere. */
int msda_rows_backward_supported(int dtype, const int64_t *shapes_host, int N, int S, int C, int L, int Q, int P);

int msda_rows_backward(int dtype, const void *src, const int64_t *level_start, const int64_t *shapes_host,
                       const void *sampling_loc, const void *attn_weight, const void *grad_out, int N, int S, int C, int L, int Q,
                       int P, void *grad_src, void *grad_sampling_loc, void *grad_attn_weight, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* RLIPV2_MSDA_H */
