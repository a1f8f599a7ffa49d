"""ctypes binding of librlipv2_msda.so (C ABI: include/rlipv2_msda.h).

The library is built in-tree by ``rlipv2_amd/csrc/Makefile`` (hipcc, gfx950).  Loading fails
loudly -- there is no fallback implementation behind this module.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librlipv2_msda.so")     # (tools load another build of the same C ABI through use_library())
CSRC_DIR = os.path.join(_HERE, "csrc")

# enum msda_dtype / msda_variant (include/rlipv2_msda.h)
MSDA_F32, MSDA_F64, MSDA_BF16 = 0, 1, 2
VARIANT_AUTO, VARIANT_GENERIC, VARIANT_QUAD, VARIANT_WINDOW, VARIANT_DEST, VARIANT_COARSE, VARIANT_CELL = 0, 1, 2, 3, 4, 5, 6
VARIANTS = {"auto": VARIANT_AUTO, "generic": VARIANT_GENERIC, "quad": VARIANT_QUAD, "window": VARIANT_WINDOW,
            "dest": VARIANT_DEST, "coarse": VARIANT_COARSE, "cell": VARIANT_CELL}
FLAG_GRAD_VALUE_ZEROED, FLAG_GRAD_VALUE_BF16, FLAG_RECORDS_SWAP = 0x100, 0x200, 0x400

EXPORTS = (
    "msda_forward", "msda_backward", "msda_forward_ex", "msda_backward_ex", "msda_check_im2col_step",
    "msda_algorithmic_bytes", "msda_strerror", "msda_abi_version", "msda_variant_name", "msda_pick_variant",
    "msda_prepare_forward", "msda_prepare_backward", "msda_backward_workspace_bytes", "msda_backward_ws",
    "msda_backward_plan_info", "msda_forward_hs", "msda_fused_forward_hs",
    "msda_fused_supported", "msda_fused_forward", "msda_fused_backward_ws", "msda_rows_backward_supported", "msda_rows_backward",
    "msda_records_bytes", "msda_records_forward", "msda_records_backward",
    # include/rlipv2_linear.h
    "linear_wgrad_workspace_bytes", "linear_wgrad_supported", "linear_wgrad_bf16",
    "linear_expand_supported", "linear_expand_bf16",
    # include/rlipv2_norm.h
    "add_layernorm_supported", "add_layernorm_workspace_bytes", "add_layernorm_forward_bf16",
    "add_layernorm_backward_bf16", "layernorm_wide_supported", "layernorm_wide_forward_bf16", "layernorm_wide_backward_bf16",
    # include/rlipv2_optim.h
    "adamw_abi_sizes", "adamw_grad_sqnorm_bf16", "adamw_step_bf16", "adamw_step_scaled_bf16",
    # include/rlipv2_alif.h
    "alif_attention_supported", "alif_attention_padded_tv", "alif_attention_forward_bf16",
    "alif_attention_softmax_backward_bf16",
    # include/rlipv2_swin.h
    "window_attention_supported", "window_attention_forward_bf16", "window_attention_backward_bf16",
    "window_attention_rows_forward_bf16", "window_attention_rows_backward_bf16",
    # include/rlipv2_elementwise.h
    "add_relu_bf16", "affine_relu_bf16", "affine_relu_backward_bf16",
    # include/rlipv2_groupnorm.h
    "groupnorm_tokens_supported", "groupnorm_tokens_workspace_bytes", "groupnorm_tokens_forward_bf16",
    "groupnorm_tokens_backward_bf16",
    # include/rlipv2_decoder.h
    "dab_refine_boxes", "dab_reference_embed",
    # include/rlipv2_matcher.h
    "hoi_assign_batch",
)

# include/rlipv2_msda_cpu.h (librlipv2_msda_cpu.so: the CPU twins, no HIP)
CPU_LIB_PATH = os.path.join(_HERE, "librlipv2_msda_cpu.so")     # (sanitizer builds: use_cpu_library())
CPU_EXPORTS = ("msda_forward_cpu", "msda_backward_cpu", "msda_cpu_strerror", "msda_cpu_abi_version")

_lib = None
_cpu_lib = None


def use_library(path: str) -> None:
    """Load another build of the C-ABI library (the ablation / timeline / emulated builds of tools/ and tests/) instead of the
    in-tree product library.  An explicit call a tool makes BEFORE the first kernel call -- the package itself reads no library
    location from the environment.  Raises if the product library has already been loaded (two builds in one process would
    each keep their own per-device state)."""
    global LIB_PATH
    if _lib is not None and os.path.abspath(path) != os.path.abspath(LIB_PATH):
        raise RuntimeError(f"use_library({path!r}): {LIB_PATH} is already loaded in this process")
    LIB_PATH = os.path.abspath(path)


def use_cpu_library(path: str) -> None:
    """the same for the CPU twins (include/rlipv2_msda_cpu.h), e.g. an ASan / UBSan build"""
    global CPU_LIB_PATH
    if _cpu_lib is not None and os.path.abspath(path) != os.path.abspath(CPU_LIB_PATH):
        raise RuntimeError(f"use_cpu_library({path!r}): {CPU_LIB_PATH} is already loaded in this process")
    CPU_LIB_PATH = os.path.abspath(path)


def build(verbose: bool = False) -> str:
    """Compile the HIP sources for gfx950 (cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC_DIR, "-j4"]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.run(cmd, check=True)
    return LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension has not been built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C rlipv2_amd/csrc`). "
            "rlipv2_amd has no CPU / PyTorch fallback for this op.")
    L = ctypes.CDLL(LIB_PATH)
    vp, i = ctypes.c_void_p, ctypes.c_int
    dims = [i] * 7
    L.msda_forward.argtypes = [i, vp, vp, vp, vp, vp, *dims, vp, vp]
    L.msda_backward.argtypes = [i, vp, vp, vp, vp, vp, vp, *dims, vp, vp, vp, vp]
    L.msda_forward_ex.argtypes = [i, i, vp, vp, vp, vp, vp, *dims, vp, vp]
    L.msda_backward_ex.argtypes = [i, i, vp, vp, vp, vp, vp, vp, *dims, vp, vp, vp, vp]
    L.msda_backward_workspace_bytes.argtypes = [i, vp, *dims]
    L.msda_backward_workspace_bytes.restype = ctypes.c_size_t
    L.msda_forward_hs.argtypes = [i, i, vp, vp, vp, vp, vp, vp, *dims, vp, vp]
    L.msda_forward_hs.restype = i
    L.msda_fused_forward_hs.argtypes = [i, i, vp, vp, vp, vp, vp, vp, i, *dims, vp, vp, vp, vp]
    L.msda_fused_forward_hs.restype = i
    L.msda_backward_plan_info.argtypes = [i, vp, *dims, vp, i]
    L.msda_backward_plan_info.restype = i
    L.msda_backward_ws.argtypes = [i, i, vp, vp, vp, vp, vp, vp, vp, *dims, vp, vp, vp, vp, ctypes.c_size_t, vp]
    L.msda_backward_ws.restype = i
    for f in (L.msda_forward, L.msda_backward, L.msda_forward_ex, L.msda_backward_ex):
        f.restype = i
    L.msda_prepare_forward.argtypes = [i, vp, vp, i, vp, i, i, i, i, vp, vp, vp]
    L.msda_prepare_backward.argtypes = [i, vp, vp, i, vp, vp, vp, vp, i, i, i, i, vp, vp, vp]
    L.msda_prepare_forward.restype = L.msda_prepare_backward.restype = i
    L.msda_fused_supported.argtypes = [i, vp, i, *dims]
    L.msda_fused_forward.argtypes = [i, vp, vp, vp, vp, vp, i, *dims, vp, vp, vp, vp]
    L.msda_fused_backward_ws.argtypes = [i, i, vp, vp, vp, vp, vp, vp, vp, i, vp, *dims, vp, vp, vp, ctypes.c_size_t, vp]
    L.msda_fused_supported.restype = L.msda_fused_forward.restype = L.msda_fused_backward_ws.restype = i
    L.msda_rows_backward_supported.argtypes = [i, vp, i, i, i, i, i, i]
    L.msda_rows_backward.argtypes = [i, vp, vp, vp, vp, vp, vp, i, i, i, i, i, i, vp, vp, vp, vp]
    L.msda_rows_backward_supported.restype = L.msda_rows_backward.restype = i
    L.msda_records_bytes.argtypes = [i, vp, *dims]
    L.msda_records_bytes.restype = ctypes.c_size_t
    L.msda_records_forward.argtypes = [i, vp, vp, vp, vp, vp, vp, i, vp, vp, *dims, vp, vp, ctypes.c_size_t, vp]
    L.msda_records_backward.argtypes = [i, i, vp, vp, vp, vp, vp, vp, vp, i, vp, *dims, vp, vp, vp, vp, vp, ctypes.c_size_t, vp,
                                        ctypes.c_size_t, vp]
    L.msda_records_forward.restype = L.msda_records_backward.restype = i
    L.msda_check_im2col_step.argtypes = [i, i]
    L.msda_check_im2col_step.restype = i
    L.msda_algorithmic_bytes.argtypes = [i, i, *dims]
    L.msda_algorithmic_bytes.restype = ctypes.c_int64
    L.msda_strerror.argtypes = [i]
    L.msda_strerror.restype = ctypes.c_char_p
    L.msda_abi_version.restype = i
    L.msda_variant_name.argtypes = [i]
    L.msda_variant_name.restype = ctypes.c_char_p
    L.msda_pick_variant.argtypes = [i, i, *dims]
    L.msda_pick_variant.restype = i
    L.linear_wgrad_workspace_bytes.argtypes = [i, i, i]
    L.linear_wgrad_workspace_bytes.restype = ctypes.c_size_t
    L.linear_wgrad_supported.argtypes = [i, i, i]
    L.linear_wgrad_supported.restype = i
    L.linear_wgrad_bf16.argtypes = [vp, vp, i, i, i, vp, vp, i, vp, ctypes.c_size_t, vp]
    L.linear_wgrad_bf16.restype = i
    L.linear_expand_supported.argtypes = [i, i, i]
    L.linear_expand_supported.restype = i
    L.linear_expand_bf16.argtypes = [vp, vp, vp, vp, i, i, i, i, vp, vp]
    L.linear_expand_bf16.restype = i
    lg, f32 = ctypes.c_long, ctypes.c_float
    L.add_layernorm_supported.argtypes = [lg, i]
    L.add_layernorm_supported.restype = i
    L.add_layernorm_workspace_bytes.argtypes = [lg, i]
    L.add_layernorm_workspace_bytes.restype = ctypes.c_size_t
    L.add_layernorm_forward_bf16.argtypes = [vp, vp, vp, vp, lg, i, f32, vp, vp, vp, vp]
    L.add_layernorm_forward_bf16.restype = i
    L.add_layernorm_backward_bf16.argtypes = [vp, vp, vp, vp, vp, vp, lg, i, vp, vp, vp, vp, ctypes.c_size_t, vp]
    L.add_layernorm_backward_bf16.restype = i
    L.layernorm_wide_supported.argtypes = [lg, i]
    L.layernorm_wide_supported.restype = i
    L.layernorm_wide_forward_bf16.argtypes = [vp, vp, vp, vp, lg, i, f32, vp, vp, vp, vp, vp]
    L.layernorm_wide_forward_bf16.restype = i
    L.layernorm_wide_backward_bf16.argtypes = [vp, vp, vp, vp, vp, vp, lg, i, vp, vp]
    L.layernorm_wide_backward_bf16.restype = i
    ip = ctypes.POINTER(ctypes.c_int)
    L.adamw_abi_sizes.argtypes = [ip, ip, ip, ip]
    L.adamw_abi_sizes.restype = i
    L.adamw_grad_sqnorm_bf16.argtypes = [vp, vp, i, vp, vp]
    L.adamw_grad_sqnorm_bf16.restype = i
    L.adamw_step_bf16.argtypes = [vp, vp, i, vp, f32, vp, i, vp]
    L.adamw_step_bf16.restype = i
    L.adamw_step_scaled_bf16.argtypes = [vp, vp, i, vp, f32, f32, vp, i, vp]
    L.adamw_step_scaled_bf16.restype = i
    L.alif_attention_supported.argtypes = [i, i, i, i, i]
    L.alif_attention_padded_tv.argtypes = [i]
    L.alif_attention_forward_bf16.argtypes = [vp, vp, vp, vp, vp, vp, f32, i, i, i, i, vp, vp, vp, vp, vp]
    L.alif_attention_softmax_backward_bf16.argtypes = [vp, vp, vp, vp, vp, vp, f32, i, i, i, i, vp, vp, vp, vp]
    L.alif_attention_softmax_backward_bf16.restype = i
    L.alif_attention_supported.restype = L.alif_attention_padded_tv.restype = L.alif_attention_forward_bf16.restype = i
    L.window_attention_supported.argtypes = [i, i, i, i]
    L.window_attention_forward_bf16.argtypes = [vp, vp, vp, vp, i, i, i, i, f32, vp, vp]
    L.window_attention_backward_bf16.argtypes = [vp, vp, vp, vp, vp, i, i, i, i, f32, vp, vp]
    L.window_attention_supported.restype = L.window_attention_forward_bf16.restype = L.window_attention_backward_bf16.restype = i
    L.window_attention_rows_forward_bf16.argtypes = [vp, vp, vp, i, i, vp, vp, vp, i, i, i, i, f32, vp, vp]
    L.window_attention_rows_backward_bf16.argtypes = [vp, vp, vp, i, i, vp, vp, vp, vp, i, i, i, i, f32, vp, vp, vp]
    L.window_attention_rows_forward_bf16.restype = L.window_attention_rows_backward_bf16.restype = i
    L.add_relu_bf16.argtypes = [vp, vp, vp, lg, vp]
    L.add_relu_bf16.restype = i
    L.affine_relu_bf16.argtypes = [vp, vp, vp, vp, lg, i, vp]
    L.affine_relu_backward_bf16.argtypes = [vp, vp, vp, vp, lg, i, vp]
    L.affine_relu_bf16.restype = L.affine_relu_backward_bf16.restype = i
    L.groupnorm_tokens_supported.argtypes = [i, i, i]
    L.groupnorm_tokens_workspace_bytes.argtypes = [i, ip, i]
    L.groupnorm_tokens_workspace_bytes.restype = ctypes.c_size_t
    pp = ctypes.POINTER(vp)
    L.groupnorm_tokens_forward_bf16.argtypes = [pp, ip, i, i, pp, pp, f32, vp, vp, vp, vp, ctypes.c_size_t, vp]
    L.groupnorm_tokens_backward_bf16.argtypes = [vp, pp, ip, i, i, pp, vp, vp, pp, pp, pp, vp, ctypes.c_size_t, vp]
    L.groupnorm_tokens_supported.restype = L.groupnorm_tokens_forward_bf16.restype = i
    L.groupnorm_tokens_backward_bf16.restype = i
    L.dab_refine_boxes.argtypes = [vp, i, vp, vp, lg, f32, vp]
    L.dab_reference_embed.argtypes = [vp, vp, vp, vp, i, i, i, i, vp, vp, i, vp]
    L.dab_refine_boxes.restype = L.dab_reference_embed.restype = i
    L.hoi_assign_batch.argtypes = [vp, i, i, i, ip, vp, vp, lg]
    L.hoi_assign_batch.restype = lg
    _lib = L
    return L


def cpu_lib() -> ctypes.CDLL:
    """librlipv2_msda_cpu.so: serves CPU tensors only (a CUDA tensor never reaches it; `lib()` above has no fallback)."""
    global _cpu_lib
    if _cpu_lib is not None:
        return _cpu_lib
    if not os.path.exists(CPU_LIB_PATH):
        raise RuntimeError(f"{CPU_LIB_PATH} is missing: run `make -C rlipv2_amd/csrc` (or __graft_entry__.build())")
    L = ctypes.CDLL(CPU_LIB_PATH)
    vp, i = ctypes.c_void_p, ctypes.c_int
    L.msda_forward_cpu.argtypes = [i, vp, vp, vp, vp, vp, *([i] * 7), vp]
    L.msda_backward_cpu.argtypes = [i, vp, vp, vp, vp, vp, vp, *([i] * 7), vp, vp, vp]
    L.msda_forward_cpu.restype = L.msda_backward_cpu.restype = L.msda_cpu_abi_version.restype = i
    L.msda_cpu_strerror.argtypes = [i]
    L.msda_cpu_strerror.restype = ctypes.c_char_p
    _cpu_lib = L
    return L


def strerror(status: int) -> str:
    return lib().msda_strerror(int(status)).decode()


def algorithmic_bytes(dtype: int, backward: bool, N, S, M, D, L, Lq, P) -> int:
    return int(lib().msda_algorithmic_bytes(dtype, int(backward), N, S, M, D, L, Lq, P))
