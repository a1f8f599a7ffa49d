// msda_internal.h -- host-side declarations shared by the kernel files and the C ABI.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rlipv2_msda.h"
#include "once_per_device.h"

// Ablation / tuning switches (environment variables read by the launchers, `dbg` bits inside kernels that skip
// work and produce WRONG results) exist only in builds with -DMSDA_ABLATION (make -C rlipv2_amd/csrc ablation ->
// librlipv2_msda_ablation.so, loaded through RLIPV2_LIB_PATH by the tools/ scripts).  The shipped library reads no
// environment and its kernels contain none of those branches: MSDA_DBG(x) folds to 0.
#ifdef MSDA_ABLATION
#include <cstdlib>
#define MSDA_DBG(x) (x)
namespace msda {
inline int ablation_env(const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; }
}
#else
#define MSDA_DBG(x) 0
namespace msda {
constexpr int ablation_env(const char *, int dflt) { return dflt; }
}
#endif

namespace msda {

struct Problem {
    int dtype;
    int N, S, M, D, L, Lq, P;
    const void *value;
    const int64_t *shapes;  // device, [L,2] (H, W)
    const int64_t *starts;  // device, [L]
    const void *loc;
    const void *aw;
    // forward
    void *out;
    // backward
    const void *grad_out;
    void *g_value;
    void *g_loc;
    void *g_aw;
    hipStream_t stream;
};

// operands of the fused sampling-geometry entry points (msda_fused_forward / msda_fused_backward_ws)
struct Fused {
    const void *qproj;      // [N*Lq, M*L*P*3] in value's dtype: offsets then logits
    const float *ref;       // [N*Lq, L, refdim]
    int refdim;             // 2 or 4
    float *loc_save;        // forward: float32 sampling_loc / attn_weight written for the backward pass (or NULL)
    float *aw_save;
    void *g_qproj;          // backward: gradient of qproj
};

// each launcher enqueues on p.stream and returns; the caller checks hipGetLastError()
void launch_generic_forward(const Problem &p);
void launch_generic_backward(const Problem &p);

bool quad_supports(const Problem &p);
void launch_quad_forward(const Problem &p);
void launch_quad_backward(const Problem &p);
void launch_quad_backward_reduce(const Problem &p);   // grad_loc / grad_aw only (no grad_value)
void launch_tile_forward(const Problem &p);           // window-staged forward (Lq == S)
void launch_quad_forward_fused(const Problem &p, const Fused &f);
bool coarse_forward_applies(const Problem &p);        // bfloat16, many queries: coarse levels resident in LDS
void launch_quad_forward_coarse(const Problem &p);
void launch_quad_backward_reduce_fused(const Problem &p, const Fused &f);   // writes f.g_qproj instead of g_loc / g_aw
#ifdef MSDA_ABLATION
void launch_quad_backward_gated(const Problem &p, const Fused *f, const int *gate);   // K1 (+ epilogue with f) iff *gate != 0
#endif

bool window_supports(const Problem &p, bool backward);
void launch_window_forward(const Problem &p);
void launch_window_backward(const Problem &p);

// destination-stationary grad_value (msda_dest.hip); shapes_host = host copy of spatial_shapes
bool dest_supports(const Problem &p, const int64_t *shapes_host);
int dest_shapes_consistent(const Problem &p, const int64_t *shapes_host);
size_t dest_workspace_bytes(const Problem &p, const int64_t *shapes_host);
// the whole backward pass of a call with host shapes: K1 (f != nullptr: with the fused geometry epilogue) + grad_value
// `records` (optional): the buffer cell_forward_kernel<., EMIT> filled for this call -- the "records" route of msda_patch.hip
void launch_backward_dest(const Problem &p, const Fused *f, const int64_t *shapes_host, void *workspace, bool out_bf16,
                          const void *records = nullptr, bool records_swap = false);
// encoder calls (Lq == S) with bfloat16 gradients: wave-autonomous 4x4-pixel patches on the matrix cores (msda_patch.hip).
// `ctl` = the zeroed control block of launch_dest_scatter (word 60 = "a sample was out of reach, fall back"),
// `mask_ws` = patch_workspace_bytes of scratch
bool patch_supports(const Problem &p, const int64_t *shapes_host);
size_t patch_workspace_bytes(const Problem &p, const int64_t *shapes_host);
void launch_patch_dest(const Problem &p, const int64_t *shapes_host, int *ctl, void *mask_ws, bool out_bf16, bool binned,
                       void *gcell_ws = nullptr, bool gcell_filled = false);
// ablation build only (else 0): offset, inside a patch_workspace_bytes buffer, of the room for the cell-major grad_out copy
size_t patch_gcell_offset(const Problem &p, const int64_t *shapes_host);
// ... preceded by cell_backward_kernel: K1's work (grad_sampling_loc / grad_attn_weight, or with `f` the projection row's
// gradient) from LDS-resident value windows on v_mfma_f32_4x4x4_16B_bf16, plus the binning for the patch pass
bool cell_backward_supports(const Problem &p, const int64_t *shapes_host);
// forward pass of an encoder call from LDS windows on the matrix cores (explicit variant MSDA_VARIANT_CELL; msda_patch.hip)
bool cell_forward_supports(const Problem &p, const int64_t *shapes_host);
void launch_cell_forward(const Problem &p, const int64_t *shapes_host, const Fused *f, void *records = nullptr);   // f: fused geometry (saves loc / aw)
// the "records" route (experiment, msda_cell_records.inc): the forward leaves per-sample records + the patch pass's masks and
// group records in one buffer of cell_records_bytes, the backward pass consumes them (no geometry, no binning)
bool cell_records_supports(const Problem &p, const int64_t *shapes_host);
size_t cell_records_bytes(const Problem &p, const int64_t *shapes_host);
const int *launch_cell_records_backward(const Problem &p, const Fused *f, const int64_t *shapes_host, const void *records,
                                        bool out_bf16, bool swap, void *gcell_ws = nullptr);
void launch_records_unbin(const Problem &p, const int64_t *shapes_host, const void *records, float *loc, float *aw, const int *gate);
// the plan of the cell + patch route as int32 values (include/rlipv2_msda.h: msda_backward_plan_info); 0 = route not taken
int patch_plan_info(const Problem &p, const int64_t *shapes_host, int32_t *out, int out_len);
void launch_cell_backward(const Problem &p, const Fused *f, const int64_t *shapes_host, int *ctl, void *mask_ws);
// few queries (decoders): one workgroup per (image, head, level), records sorted by pixel in LDS (msda_sparse.hip)
bool sparse_dest_supports(const Problem &p, const int64_t *shapes_host);
void launch_sparse_dest(const Problem &p, const int64_t *shapes_host, bool out_bf16);

}  // namespace msda
