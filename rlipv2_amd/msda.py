"""Host-side mirror of the reference's MSDA operator interface, bound to the HIP library.

Drop-in surface (same names, argument order and error behaviour as the reference):

* ``ms_deform_attn_forward`` / ``ms_deform_attn_backward`` -- the two functions the reference's
  extension module ``MultiScaleDeformableAttention`` exports (models/ops/src/vision.cpp:13-16;
  semantics models/ops/src/cuda/ms_deform_attn_cuda.cu:20-153).
* ``MSDeformAttnFunction`` -- the autograd Function of
  models/ops/functions/ms_deform_attn_func.py:25-42 (6 positional args, gradients for
  value / sampling_locations / attention_weights, once-differentiable).

Differences, all additive: bfloat16 ``value`` is accepted (sampling locations and attention
weights are then float32, or bfloat16 which is upcast; gradients come back in the inputs'
dtypes); a failed kernel launch raises instead of being printf()'d
(ms_deform_im2col_cuda.cuh:948-952); CPU tensors are served by the CPU twins of the two entry points
(include/rlipv2_msda_cpu.h, csrc/msda_cpu.cpp: plain C++ / OpenMP) where the reference raises
"Not implemented on the CPU" (models/ops/src/ms_deform_attn.h:54) and its models fall back to
``ms_deform_attn_core_pytorch``.  That is a dispatch on the tensors' device, not a fallback: CUDA tensors only
ever run the HIP library, and a missing HIP library raises.
"""
from __future__ import annotations

import contextlib
import ctypes
import os

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib, roofline

_DTYPES = {torch.float32: _lib.MSDA_F32, torch.float64: _lib.MSDA_F64, torch.bfloat16: _lib.MSDA_BF16}

# kernel override for benchmarks / tests ("auto" in product use)
_variant_fwd = _lib.VARIANT_AUTO
_variant_bwd = _lib.VARIANT_AUTO
# the train step's encoder forward through cell_forward_kernel (see ms_deform_attn_fused_forward): an experiment that has
# not run on hardware; a plain attribute that only tools / bench.py --msda-fwd-cell set, never read from the environment
fused_forward_cell = False
# The "records" route of the encoder's fused call (csrc/msda_cell_forward.inc EMIT, csrc/msda_cell_records.inc): the forward pass
# leaves per-sample records, window tables and the patch pass's masks; the backward pass runs no sample geometry and no binning.
# Bit-identical to the product kernels on the lane-level model (tests/test_records_emulated.py); NEVER run on hardware: a plain
# attribute that only tools / bench.py's experiments set.  records_swap: the operand order of the 4x4x4 products in which every lane
# receives all four dots (16 DPP moves per level and group less; same products, same sums); False: cell_backward_kernel's order.
records_route = False
records_swap = True
_records_out = [None]      # the records tensor of the last ms_deform_attn_fused_forward call under records_route (take_records)


def take_records():
    """the records buffer the last ms_deform_attn_fused_forward call filled under msda.records_route (None otherwise); handed over
    once -- to FusedMSDeformAttnFunction.forward, which saves it for the backward pass in place of sampling_loc / attn_weight"""
    r, _records_out[0] = _records_out[0], None
    return r
# name of the kernel variant the last call of each direction ran (read by bench.py's roofline line)
last_variant = {}


def _on_device(t) -> bool:
    """(the fused entry points ask here, not t.is_cuda: tests/test_records_emulated.py runs them on the lane-level model of the
    kernels, where host tensors stand in for device tensors)"""
    return t.is_cuda


@contextlib.contextmanager
def _launch(t):
    """device guard of a launch; yields the stream its kernels go to"""
    with torch.cuda.device(t.device):
        yield torch.cuda.current_stream().cuda_stream


def set_variant(forward: str, backward: str = None) -> None:
    """Pin one kernel implementation ("auto", "generic", "quad", "window") per direction."""
    global _variant_fwd, _variant_bwd
    _variant_fwd = _lib.VARIANTS[forward]
    _variant_bwd = _lib.VARIANTS[forward if backward is None else backward]


def _check_inputs(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, extra=()):
    named = [("value", value), ("spatial_shapes", spatial_shapes), ("level_start_index", level_start_index),
             ("sampling_loc", sampling_loc), ("attn_weight", attn_weight), *extra]
    for name, t in named:
        if not t.is_contiguous():
            raise RuntimeError(f"{name} tensor has to be contiguous")   # ms_deform_attn_cuda.cu:28-32
        if value.is_cuda and not t.is_cuda:
            raise RuntimeError(f"{name} must be a CUDA tensor")         # ms_deform_attn_cuda.cu:34-38
        if t.device != value.device:
            raise RuntimeError(f"{name} is on {t.device}, value is on {value.device}")
    if value.dtype not in _DTYPES:
        raise RuntimeError(f"ms_deform_attn: unsupported dtype {value.dtype}")
    if spatial_shapes.dtype != torch.int64 or level_start_index.dtype != torch.int64:
        raise RuntimeError("spatial_shapes and level_start_index must be int64 tensors")
    if value.dim() != 4 or sampling_loc.dim() != 6 or attn_weight.dim() != 5:
        raise RuntimeError("expected value [N,S,M,D], sampling_loc [N,Lq,M,L,P,2], attn_weight [N,Lq,M,L,P]")
    # the kernels index every operand with the dimensions taken from value / spatial_shapes / sampling_loc: a
    # mismatched (e.g. un-broadcast) operand would be read out of bounds (the reference does not check either; its
    # out-of-bounds reads are undefined behaviour -- here they are an error)
    N, M = value.shape[0], value.shape[2]
    L = spatial_shapes.shape[0]
    Lq, P = sampling_loc.shape[1], sampling_loc.shape[4]
    if tuple(sampling_loc.shape) != (N, Lq, M, L, P, 2) or tuple(attn_weight.shape) != (N, Lq, M, L, P):
        raise RuntimeError(f"sampling_loc {tuple(sampling_loc.shape)} / attn_weight {tuple(attn_weight.shape)} do not match "
                           f"value {tuple(value.shape)} and {L} levels: expected [{N},{Lq},{M},{L},{P},2] and [{N},{Lq},{M},{L},{P}]")
    if tuple(spatial_shapes.shape) != (L, 2) or level_start_index.numel() != L:
        raise RuntimeError("spatial_shapes must be [L, 2] and level_start_index [L]")
    for name, t in extra:
        if name == "grad_output" and (t.dim() != 3 or t.shape[0] != N or t.shape[1] != Lq or t.shape[2] != M * value.shape[3]):
            raise RuntimeError(f"grad_output {tuple(t.shape)} does not match [N, Lq, M*D] = [{N}, {Lq}, {M * value.shape[3]}]")


def _dims(value, spatial_shapes, sampling_loc):
    N, S, M, D = value.shape
    L = spatial_shapes.shape[0]
    Lq, P = sampling_loc.shape[1], sampling_loc.shape[4]
    return N, S, M, D, L, Lq, P


def _aux_dtype(value):
    """dtype of sampling_loc / attn_weight / all gradients as the kernels see them."""
    return torch.float64 if value.dtype == torch.float64 else torch.float32


def host_shapes(spatial_shapes):
    """Host copy of the int64 [L, 2] level shapes, cached on the tensor object.  Callers that build the pyramid on
    the host attach it up front (``attach_host_shapes``); otherwise the first use reads the tensor back once -- the
    sync the reference module pays on every call for its ``sum(H*W) == Len_in`` assert (ms_deform_attn.py:96).
    Returns None while a HIP graph is being captured and nothing is cached."""
    cached = getattr(spatial_shapes, "_msda_host_shapes", None)
    if cached is not None and cached[0] == spatial_shapes._version:
        return cached[1]
    if torch.cuda.is_current_stream_capturing():
        return None
    hs = tuple(int(v) for v in spatial_shapes.reshape(-1).tolist())
    attach_host_shapes(spatial_shapes, hs)
    return hs


def attach_host_shapes(spatial_shapes, shapes):
    """Record the host-side values of a device ``spatial_shapes`` tensor (list of (H, W))."""
    flat = tuple(int(v) for hw in shapes for v in (hw if isinstance(hw, (tuple, list)) else (hw,)))
    spatial_shapes._msda_host_shapes = (spatial_shapes._version, flat)
    return spatial_shapes


def _raise(status):
    raise RuntimeError(f"ms_deform_attn: {_lib.strerror(status)} (status {status})")


def _cpu_call(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output=None):
    """CPU tensors: the CPU twins of the two entry points (include/rlipv2_msda_cpu.h, csrc/msda_cpu.cpp).
    float64 runs in float64, everything else (float32, bfloat16) in float32; results come back in the operands' dtypes,
    like the GPU path.  Only ever reached with CPU tensors."""
    L = _lib.cpu_lib()
    N, S, M, D, nL, Lq, P = _dims(value, spatial_shapes, sampling_loc)
    work = torch.float64 if value.dtype == torch.float64 else torch.float32
    code = 1 if work == torch.float64 else 0
    v, loc, aw = (t if t.dtype == work else t.to(work) for t in (value, sampling_loc, attn_weight))
    if grad_output is None:
        out = torch.empty((N, Lq, M * D), dtype=work)
        st = L.msda_forward_cpu(code, v.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(), loc.data_ptr(),
                                aw.data_ptr(), N, S, M, D, nL, Lq, P, out.data_ptr())
        res = [out.to(value.dtype)]
    else:
        go = grad_output if grad_output.dtype == work else grad_output.to(work)
        g_value, g_loc, g_aw = torch.empty(value.shape, dtype=work), torch.empty(loc.shape, dtype=work), torch.empty(aw.shape, dtype=work)
        st = L.msda_backward_cpu(code, v.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(), loc.data_ptr(),
                                 aw.data_ptr(), go.data_ptr(), N, S, M, D, nL, Lq, P, g_value.data_ptr(), g_loc.data_ptr(),
                                 g_aw.data_ptr())
        res = [g_value.to(value.dtype), g_loc.to(sampling_loc.dtype), g_aw.to(attn_weight.dtype)]
    if st:
        raise RuntimeError(f"ms_deform_attn (CPU): {L.msda_cpu_strerror(st).decode()} (status {st})")
    return res


def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step):
    """Reference: ms_deform_attn_forward (vision.cpp:14) -> out [N, Lq, M*D]."""
    _check_inputs(value, spatial_shapes, level_start_index, sampling_loc, attn_weight)
    if not value.is_cuda:
        step = min(value.shape[0], int(im2col_step))                                          # ms_deform_attn_cuda.cu:50-52
        if step <= 0 or value.shape[0] % step != 0:
            if value.shape[0] != 0:
                raise RuntimeError("ms_deform_attn: batch must divide im2col_step: batch % min(batch, im2col_step) != 0")
        return _cpu_call(value, spatial_shapes, level_start_index, sampling_loc, attn_weight)[0]
    L = _lib.lib()
    N, S, M, D, nL, Lq, P = _dims(value, spatial_shapes, sampling_loc)
    st = L.msda_check_im2col_step(N, int(im2col_step))
    if st:
        _raise(st)
    aux = _aux_dtype(value)
    loc = sampling_loc if sampling_loc.dtype == aux else sampling_loc.to(aux)
    aw = attn_weight if attn_weight.dtype == aux else attn_weight.to(aux)
    out = torch.empty((N, Lq, M * D), dtype=value.dtype, device=value.device)
    with torch.cuda.device(value.device):
        stream = torch.cuda.current_stream().cuda_stream
        if _variant_fwd == _lib.VARIANT_CELL:       # (explicit, experimental: needs the host copy of the level shapes)
            hs = host_shapes(spatial_shapes)
            hs_arr = (ctypes.c_int64 * len(hs))(*hs) if hs is not None else None
            st = L.msda_forward_hs(_variant_fwd, _DTYPES[value.dtype], value.data_ptr(), spatial_shapes.data_ptr(),
                                   level_start_index.data_ptr(), hs_arr, loc.data_ptr(), aw.data_ptr(),
                                   N, S, M, D, nL, Lq, P, out.data_ptr(), stream)
        else:
            st = L.msda_forward_ex(_variant_fwd, _DTYPES[value.dtype], value.data_ptr(), spatial_shapes.data_ptr(),
                                   level_start_index.data_ptr(), loc.data_ptr(), aw.data_ptr(),
                                   N, S, M, D, nL, Lq, P, out.data_ptr(), stream)
    if st:
        _raise(st)
    roofline.add(_lib.algorithmic_bytes(_DTYPES[value.dtype], False, N, S, M, D, nL, Lq, P))
    last_variant["fwd"] = L.msda_variant_name(
        _variant_fwd or L.msda_pick_variant(0, _DTYPES[value.dtype], N, S, M, D, nL, Lq, P)).decode()
    return out


def ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output,
                            im2col_step, host=None):
    """Reference: ms_deform_attn_backward (vision.cpp:15) -> [grad_value, grad_sampling_loc, grad_attn_weight].

    With a host copy of the level shapes available (``host_shapes``) the library runs its destination-stationary
    grad_value pass (csrc/msda_dest.hip): no float atomics, repeatable bit for bit, grad_value written directly in
    value's dtype."""
    _check_inputs(value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                  extra=(("grad_output", grad_output),))
    if not value.is_cuda:
        return _cpu_call(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output)
    L = _lib.lib()
    N, S, M, D, nL, Lq, P = _dims(value, spatial_shapes, sampling_loc)
    st = L.msda_check_im2col_step(N, int(im2col_step))
    if st:
        _raise(st)
    aux = _aux_dtype(value)
    loc = sampling_loc if sampling_loc.dtype == aux else sampling_loc.to(aux)
    aw = attn_weight if attn_weight.dtype == aux else attn_weight.to(aux)
    go = grad_output if grad_output.dtype == value.dtype else grad_output.to(value.dtype)
    g_loc = torch.empty(sampling_loc.shape, dtype=aux, device=value.device)
    g_aw = torch.empty(attn_weight.shape, dtype=aux, device=value.device)
    hs = None
    if _variant_bwd in (_lib.VARIANT_AUTO, _lib.VARIANT_DEST):
        hs = host if host is not None else host_shapes(spatial_shapes)
    ws_bytes = 0
    if hs is not None:
        hs_arr = (ctypes.c_int64 * len(hs))(*hs)
        if sum(hs[0::2][k] * hs[1::2][k] for k in range(len(hs) // 2)) != S:
            raise RuntimeError("ms_deform_attn: sum(H*W) of spatial_shapes != value.shape[1]")   # ms_deform_attn.py:96
        ws_bytes = int(L.msda_backward_workspace_bytes(_DTYPES[value.dtype], hs_arr, N, S, M, D, nL, Lq, P))
    with torch.cuda.device(value.device):
        stream = torch.cuda.current_stream().cuda_stream
        if ws_bytes:
            # every grad_value row has exactly one writer: written in value's dtype, no zero-fill
            flags = _lib.FLAG_GRAD_VALUE_BF16 if value.dtype == torch.bfloat16 else 0
            g_value = torch.empty(value.shape, dtype=value.dtype, device=value.device)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=value.device)
            st = L.msda_backward_ws(_variant_bwd | flags, _DTYPES[value.dtype], value.data_ptr(),
                                    spatial_shapes.data_ptr(), level_start_index.data_ptr(), hs_arr, loc.data_ptr(),
                                    aw.data_ptr(), go.data_ptr(), N, S, M, D, nL, Lq, P, g_value.data_ptr(),
                                    g_loc.data_ptr(), g_aw.data_ptr(), ws.data_ptr(), ws_bytes, stream)
        else:
            if _variant_bwd == _lib.VARIANT_DEST:
                _raise(-6)
            g_value = torch.empty(value.shape, dtype=aux, device=value.device)   # zero-filled inside the library
            st = L.msda_backward_ex(_variant_bwd, _DTYPES[value.dtype], value.data_ptr(), spatial_shapes.data_ptr(),
                                    level_start_index.data_ptr(), loc.data_ptr(), aw.data_ptr(), go.data_ptr(),
                                    N, S, M, D, nL, Lq, P, g_value.data_ptr(), g_loc.data_ptr(), g_aw.data_ptr(),
                                    stream)
    if st:
        _raise(st)
    roofline.add(_lib.algorithmic_bytes(_DTYPES[value.dtype], True, N, S, M, D, nL, Lq, P)
                 - (N * S * M * D * 2 if g_value.dtype == torch.bfloat16 else 0))
    last_variant["bwd"] = "dest" if ws_bytes else L.msda_variant_name(
        _variant_bwd or L.msda_pick_variant(1, _DTYPES[value.dtype], N, S, M, D, nL, Lq, P)).decode()
    if g_value.dtype != value.dtype:
        g_value = g_value.to(value.dtype)
    if g_loc.dtype != sampling_loc.dtype:
        g_loc = g_loc.to(sampling_loc.dtype)
    if g_aw.dtype != attn_weight.dtype:
        g_aw = g_aw.to(attn_weight.dtype)
    return [g_value, g_loc, g_aw]


class MSDeformAttnFunction(Function):
    """Reference: models/ops/functions/ms_deform_attn_func.py:25-42."""

    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights,
                im2col_step):
        ctx.im2col_step = im2col_step
        output = ms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                                        attention_weights, ctx.im2col_step)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                              attention_weights)
        ctx.host_shapes = host_shapes(value_spatial_shapes) if value.is_cuda else None
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, shapes, starts, loc, aw = ctx.saved_tensors
        g_value, g_loc, g_aw = ms_deform_attn_backward(value, shapes, starts, loc, aw, grad_output.contiguous(),
                                                       ctx.im2col_step, host=ctx.host_shapes)
        return g_value, None, None, g_loc, g_aw, None


class SampleRowsFunction(Function):
    """The op with ONE head of 256 channels and the (query, head) pairs as queries (deform_attn.MSDeformAttn._sampled_projection:
    `value` = the unprojected memory [N, S, 1, 256], sampling_locations [N, Q, 1, L, P, 2], attention_weights [N, Q, 1, L, P]).
    Forward: the library's forward as it is.  Backward: csrc/msda_rows.hip when it takes the call (CUDA, 256 channels, float32 /
    bfloat16, host copy of the level shapes) -- every row of the memory's gradient written once, no float atomics, sums in
    sample order -- and the library's general backward otherwise.  Same signature and gradients as MSDeformAttnFunction."""

    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights, im2col_step):
        ctx.im2col_step = im2col_step
        output = ms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                                        attention_weights, ctx.im2col_step)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights)
        ctx.host_shapes = host_shapes(value_spatial_shapes) if value.is_cuda else None
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, shapes, starts, loc, aw = ctx.saved_tensors
        hs = ctx.host_shapes
        N, S, M, D = value.shape
        nL, Q, P = shapes.shape[0], loc.shape[1], loc.shape[4]
        L = _lib.lib() if value.is_cuda else None
        arr = (ctypes.c_int64 * len(hs))(*hs) if hs is not None else None
        if (L is None or arr is None or M != 1 or value.dtype not in (torch.float32, torch.bfloat16) or loc.dtype != torch.float32
                or aw.dtype != torch.float32 or not L.msda_rows_backward_supported(_DTYPES[value.dtype], arr, N, S, D, nL, Q, P)):
            g_value, g_loc, g_aw = ms_deform_attn_backward(value, shapes, starts, loc, aw, grad_output.contiguous(),
                                                           ctx.im2col_step, host=hs)
            return g_value, None, None, g_loc, g_aw, None
        go = grad_output.contiguous()
        go = go if go.dtype == value.dtype else go.to(value.dtype)
        g_value = torch.empty_like(value)                             # every row is written by the kernel
        g_loc, g_aw = torch.empty_like(loc), torch.empty_like(aw)
        with torch.cuda.device(value.device):
            st = L.msda_rows_backward(_DTYPES[value.dtype], value.data_ptr(), starts.data_ptr(), arr, loc.data_ptr(), aw.data_ptr(),
                                      go.data_ptr(), N, S, D, nL, Q, P, g_value.data_ptr(), g_loc.data_ptr(), g_aw.data_ptr(),
                                      torch.cuda.current_stream().cuda_stream)
        if st:
            _raise(st)
        roofline.add(roofline.tensor_bytes(loc, aw, go, g_value, g_loc, g_aw))
        last_variant["bwd"] = "rows"
        return g_value, None, None, g_loc, g_aw, None


def fused_supported(value, spatial_shapes, reference_points, Lq, L, P, need_backward):
    """Can the fused geometry + sampling kernels (msda_fused_forward / msda_fused_backward_ws) take this call?"""
    if not _on_device(value) or value.dtype not in (torch.float32, torch.bfloat16) or value.dim() != 4:
        return False
    hs = host_shapes(spatial_shapes) if need_backward else None
    if need_backward and hs is None:
        return False
    N, S, M, D = value.shape
    arr = (ctypes.c_int64 * len(hs))(*hs) if hs is not None else None
    level = _lib.lib().msda_fused_supported(_DTYPES[value.dtype], arr, reference_points.shape[-1], N, S, M, D, L, Lq, P)
    return level >= (2 if need_backward else 1)


def ms_deform_attn_fused_forward(value, spatial_shapes, level_start_index, qproj, ref, save):
    """value [N, S, M, D], qproj [N, Lq, M*L*P*3] (value's dtype), ref [N, Lq, L, 2|4] float32 ->
    (out [N, Lq, M*D], sampling_loc | None, attn_weight | None): ms_deform_attn.py:101-117 in one launch.
    Under msda.records_route (encoder calls): (out, None, None) and the records buffer through take_records()."""
    L = _lib.lib()
    N, S, M, D = value.shape
    nL, Lq = spatial_shapes.shape[0], qproj.shape[1]
    P = qproj.shape[2] // (M * nL * 3)
    out = torch.empty((N, Lq, M * D), dtype=value.dtype, device=value.device)
    _records_out[0] = None
    if records_route and save and value.dtype == torch.bfloat16 and host_shapes(spatial_shapes) is not None:
        hs = host_shapes(spatial_shapes)
        hs_arr = (ctypes.c_int64 * len(hs))(*hs)
        rec_bytes = int(L.msda_records_bytes(_DTYPES[value.dtype], hs_arr, N, S, M, D, nL, Lq, P))
        if rec_bytes:                                      # (0: not a call the route takes -- the decoders, other shapes)
            # the records are the whole saved state: no float32 locations / weights (the group records hold the same floats)
            records = torch.empty(rec_bytes, dtype=torch.uint8, device=value.device)
            with _launch(value) as stream:
                st = L.msda_records_forward(_DTYPES[value.dtype], value.data_ptr(), spatial_shapes.data_ptr(),
                                            level_start_index.data_ptr(), hs_arr, qproj.data_ptr(), ref.data_ptr(), ref.shape[-1],
                                            None, None, N, S, M, D, nL, Lq, P, out.data_ptr(), records.data_ptr(), rec_bytes, stream)
            if st:
                _raise(st)
            roofline.add(roofline.tensor_bytes(value, qproj, ref, out, records))
            last_variant["fwd"] = "cell+geometry+records"
            _records_out[0] = records
            return out, None, None
    loc = torch.empty((N, Lq, M, nL, P, 2), dtype=torch.float32, device=value.device) if save else None
    aw = torch.empty((N, Lq, M, nL, P), dtype=torch.float32, device=value.device) if save else None
    if (fused_forward_cell and save and value.dtype == torch.bfloat16 and Lq == S and nL == 4 and P == 4 and D == 32
            and host_shapes(spatial_shapes) is not None):
        # EXPERIMENT (msda.fused_forward_cell = True; the kernel has not been validated on hardware yet): geometry, the saved float32
        # locations / weights and the sampling from LDS windows on the matrix cores in one kernel
        # (csrc/msda_cell_forward.inc: cell_forward_kernel<refdim, 0>)
        hs = host_shapes(spatial_shapes)
        hs_arr = (ctypes.c_int64 * len(hs))(*hs)
        with _launch(value) as stream:
            st = L.msda_fused_forward_hs(_lib.VARIANT_CELL, _DTYPES[value.dtype], value.data_ptr(), spatial_shapes.data_ptr(),
                                         level_start_index.data_ptr(), hs_arr, qproj.data_ptr(), ref.data_ptr(), ref.shape[-1],
                                         N, S, M, D, nL, Lq, P, out.data_ptr(), loc.data_ptr(), aw.data_ptr(), stream)
        if st:
            _raise(st)
        roofline.add(roofline.tensor_bytes(value, qproj, ref, out, loc, aw))
        last_variant["fwd"] = "cell+geometry"
        return out, loc, aw
    with _launch(value) as stream:
        st = L.msda_fused_forward(_DTYPES[value.dtype], value.data_ptr(), spatial_shapes.data_ptr(),
                                  level_start_index.data_ptr(), qproj.data_ptr(), ref.data_ptr(), ref.shape[-1],
                                  N, S, M, D, nL, Lq, P, out.data_ptr(), loc.data_ptr() if save else None,
                                  aw.data_ptr() if save else None, stream)
    if st:
        _raise(st)
    roofline.add(roofline.tensor_bytes(value, qproj, ref, out, loc, aw))
    last_variant["fwd"] = "quad+geometry"
    return out, loc, aw


def ms_deform_attn_fused_backward(value, spatial_shapes, level_start_index, loc, aw, ref, grad_output, host, records=None):
    """-> [grad_value (value's dtype), grad_qproj (value's dtype)]; needs the host copy of the level shapes.
    records: what the forward call left (take_records) under msda.records_route -- loc / aw are None then."""
    L = _lib.lib()
    N, S, M, D = value.shape
    nL, Lq = spatial_shapes.shape[0], grad_output.shape[1]
    P = loc.shape[4] if loc is not None else 4
    hs_arr = (ctypes.c_int64 * len(host))(*host)
    if sum(host[0::2][k] * host[1::2][k] for k in range(len(host) // 2)) != S:
        raise RuntimeError("ms_deform_attn: sum(H*W) of spatial_shapes != value.shape[1]")   # ms_deform_attn.py:96
    ws_bytes = int(L.msda_backward_workspace_bytes(_DTYPES[value.dtype], hs_arr, N, S, M, D, nL, Lq, P))
    go = grad_output if grad_output.dtype == value.dtype else grad_output.to(value.dtype)
    g_value = torch.empty(value.shape, dtype=value.dtype, device=value.device)
    g_qproj = torch.empty((N, Lq, M * nL * P * 3), dtype=value.dtype, device=value.device)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=value.device)
    flags = _lib.FLAG_GRAD_VALUE_BF16 if value.dtype == torch.bfloat16 else 0
    if records is not None:
        with _launch(value) as stream:
            st = L.msda_records_backward(flags | (_lib.FLAG_RECORDS_SWAP if records_swap else 0), _DTYPES[value.dtype],
                                         value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(), hs_arr,
                                         loc.data_ptr() if loc is not None else None, aw.data_ptr() if aw is not None else None,
                                         ref.data_ptr(), ref.shape[-1], go.data_ptr(),
                                         N, S, M, D, nL, Lq, P, g_value.data_ptr(), None, None, g_qproj.data_ptr(),
                                         records.data_ptr(), records.numel(), ws.data_ptr(), ws_bytes, stream)
        if st:
            _raise(st)
        roofline.add(roofline.tensor_bytes(value, records, ref, go, g_value, g_qproj))
        last_variant["bwd"] = "records+geometry"
        return [g_value, g_qproj]
    with _launch(value) as stream:
        st = L.msda_fused_backward_ws(flags, _DTYPES[value.dtype], value.data_ptr(), spatial_shapes.data_ptr(),
                                      level_start_index.data_ptr(), hs_arr, loc.data_ptr(), aw.data_ptr(),
                                      ref.data_ptr(), ref.shape[-1], go.data_ptr(), N, S, M, D, nL, Lq, P,
                                      g_value.data_ptr(), g_qproj.data_ptr(), ws.data_ptr(), ws_bytes, stream)
    if st:
        _raise(st)
    roofline.add(roofline.tensor_bytes(value, loc, aw, ref, go, g_value, g_qproj))
    last_variant["bwd"] = "dest+geometry"
    return [g_value, g_qproj]


class FusedMSDeformAttnFunction(Function):
    """(value, spatial_shapes, level_start_index, qproj, reference_points) -> out: the module's sampling geometry
    (ms_deform_attn.py:101-112) and MSDeformAttnFunction (ms_deform_attn_func.py:25-42) as ONE kernel each way
    (csrc/msda_quad.hip: quad_forward_fused_kernel, quad_backward_shared_kernel<.., REFDIM>).  The forward reads the
    raw projection rows; float32 sampling_loc / attn_weight are written only when a backward pass will need them, and
    their gradients never exist in memory.  No gradient for the reference points (the caller checks)."""

    @staticmethod
    def forward(ctx, value, spatial_shapes, level_start_index, qproj, reference_points, im2col_step):
        if not _on_device(value):
            raise RuntimeError("Not implemented on the CPU")                 # ms_deform_attn.h:54
        for name, t in (("value", value), ("spatial_shapes", spatial_shapes), ("level_start_index", level_start_index)):
            if not t.is_contiguous():
                raise RuntimeError(f"{name} tensor has to be contiguous")
        st = _lib.lib().msda_check_im2col_step(value.shape[0], int(im2col_step))
        if st:
            _raise(st)
        qproj = qproj.contiguous()
        ref = reference_points.float().contiguous()
        save = ctx.needs_input_grad[0] or ctx.needs_input_grad[3]
        out, loc, aw = ms_deform_attn_fused_forward(value, spatial_shapes, level_start_index, qproj, ref, save)
        if save:
            records = take_records()                       # (msda.records_route: instead of loc / aw)
            ctx.has_records = records is not None
            ctx.save_for_backward(value, spatial_shapes, level_start_index, ref, *([records] if ctx.has_records else [loc, aw]))
            ctx.host_shapes = host_shapes(spatial_shapes)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, shapes, starts, ref = ctx.saved_tensors[:4]
        loc, aw, records = (None, None, ctx.saved_tensors[4]) if ctx.has_records else (*ctx.saved_tensors[4:6], None)
        g_value, g_qproj = ms_deform_attn_fused_backward(value, shapes, starts, loc, aw, ref, grad_output.contiguous(),
                                                         ctx.host_shapes, records)
        return g_value, None, None, g_qproj, None, None


class SamplingGeometryFunction(Function):
    """qproj [N, Lq, M*L*P*3] (offsets then logits), reference_points [N, Lq, L, 2|4] ->
    (sampling_locations [N, Lq, M, L, P, 2], attention_weights [N, Lq, M, L, P]), both float32 -- the
    view / softmax / normalise / add chain of the reference module (ms_deform_attn.py:101-112) as one
    HIP kernel each way (csrc/msda_prep.hip).  L = P = 4."""

    @staticmethod
    def forward(ctx, qproj, reference_points, spatial_shapes, M, L, P):
        if not qproj.is_cuda:
            raise RuntimeError("Not implemented on the CPU")
        lib = _lib.lib()
        N, Lq, _ = qproj.shape
        qproj = qproj.contiguous()
        ref = reference_points.float().contiguous()
        loc = torch.empty((N, Lq, M, L, P, 2), dtype=torch.float32, device=qproj.device)
        aw = torch.empty((N, Lq, M, L, P), dtype=torch.float32, device=qproj.device)
        with torch.cuda.device(qproj.device):
            st = lib.msda_prepare_forward(_DTYPES[qproj.dtype], qproj.data_ptr(), ref.data_ptr(), ref.shape[-1],
                                          spatial_shapes.data_ptr(), N * Lq, M, L, P, loc.data_ptr(), aw.data_ptr(),
                                          torch.cuda.current_stream().cuda_stream)
        if st:
            _raise(st)
        roofline.add(roofline.tensor_bytes(qproj, ref, loc, aw))
        ctx.save_for_backward(qproj, ref, spatial_shapes, aw)
        ctx.dims = (M, L, P)
        ctx.ref_dtype = reference_points.dtype
        return loc, aw

    @staticmethod
    @once_differentiable
    def backward(ctx, g_loc, g_aw):
        qproj, ref, spatial_shapes, aw = ctx.saved_tensors
        M, L, P = ctx.dims
        lib = _lib.lib()
        N, Lq, _ = qproj.shape
        g_qproj = torch.empty_like(qproj)
        need_ref = ctx.needs_input_grad[1]
        g_ref = torch.empty_like(ref) if need_ref else None
        gl, ga = g_loc.float().contiguous(), g_aw.float().contiguous()      # kept alive across the launch
        with torch.cuda.device(qproj.device):
            st = lib.msda_prepare_backward(_DTYPES[qproj.dtype], qproj.data_ptr(), ref.data_ptr(), ref.shape[-1],
                                           spatial_shapes.data_ptr(), aw.data_ptr(),
                                           gl.data_ptr(), ga.data_ptr(),
                                           N * Lq, M, L, P, g_qproj.data_ptr(),
                                           g_ref.data_ptr() if need_ref else None,
                                           torch.cuda.current_stream().cuda_stream)
        if st:
            _raise(st)
        roofline.add(roofline.tensor_bytes(qproj, ref, aw, gl, ga, g_qproj, g_ref))
        return g_qproj, (g_ref.to(ctx.ref_dtype) if need_ref else None), None, None, None, None
