"""MSDeformAttn module (boundary B2): same constructor, forward signature, parameter names and
initialisation as the reference (models/ops/modules/ms_deform_attn.py:34-119 and its twin under
models/dab_deformable/ops/modules/), with the sampling + aggregation done by the HIP kernels
behind ``MSDeformAttnFunction``.

Host-side differences (results identical): the two query projections (`sampling_offsets`,
`attention_weights`) are issued as one GEMM over the concatenated weights, the padding-mask
fill is applied to the projected value without an extra copy, and on the encoder's 88 892-token inputs
the weight gradients of the projections run on the MFMA kernel of csrc/token_gemm.hip (linear.py).
"""
from __future__ import annotations

import math
import warnings

import torch
import torch.nn.functional as F
from torch import nn
from torch.nn.init import constant_, xavier_uniform_

from . import msda
from .linear import token_linear

# the autograd op; a module attribute so that CPU unit tests can substitute a checker
msda_function = msda.MSDeformAttnFunction
# use the fused sampling-geometry kernel on the GPU (tests switch it off to compare both routes)
fused_geometry = True
# fold that geometry into the sampling kernels themselves (msda.FusedMSDeformAttnFunction) when the reference points
# need no gradient
fused_sampling = True
# Cross-attention with FEW queries (the decoders: 150 / 300 queries into 88 892 memory tokens): sample the UNPROJECTED memory and
# project the few sampled rows, instead of projecting every memory token in every decoder layer (MSDeformAttn._sampled_projection).
# OFF: the formulation is exact (CPU tests against the standard one), what it costs on the GPU depends on kernels that have not
# been timed.  A plain attribute: tools/experiments_r05.py --stp times it; it is NOT one of routes.py's
# self-checked routes (nothing switches it on by itself).
sample_then_project = False
SAMPLE_THEN_PROJECT_MAX_FRACTION = 0.25       # taken when queries x heads <= this fraction of the memory tokens


class _AdjacentCat(torch.autograd.Function):
    """torch.cat([a, b], 0) of two parameters that live back to back in ONE buffer (`flat`): the forward hands out the
    buffer itself, the backward hands each parameter its slice of the incoming gradient -- no copy either way."""

    @staticmethod
    def forward(ctx, flat, a, b):
        ctx.rows = a.shape[0]
        return flat.detach()

    @staticmethod
    def backward(ctx, g):
        return None, g[:ctx.rows], g[ctx.rows:]


class _AdjacentCatN(torch.autograd.Function):
    """torch.cat(params, 0) of parameters that live back to back in ONE buffer (see _AdjacentCat)."""

    @staticmethod
    def forward(ctx, flat, *params):
        ctx.rows = [p.shape[0] for p in params]
        return flat.detach()

    @staticmethod
    def backward(ctx, g):
        out, r0 = [None], 0
        for r in ctx.rows:
            out.append(g[r0:r0 + r])
            r0 += r
        return tuple(out)


shared_parameter_storage = True


def adjacent_cat_n(owner, key, params):
    """`adjacent_cat` for any number of parameters (the query / key / value projections of a RoBERTa layer)."""
    if not shared_parameter_storage:
        return torch.cat(list(params), 0)
    flat = owner.__dict__.get(key)
    ok = flat is not None and flat.shape[0] == sum(p.shape[0] for p in params)
    if ok:
        off = flat.data_ptr()
        for p in params:
            ok = ok and p.data_ptr() == off and p.is_contiguous() and p.dtype == flat.dtype and p.device == flat.device
            off += p.numel() * p.element_size()
    if not ok:
        if torch.cuda.is_available() and params[0].is_cuda and torch.cuda.is_current_stream_capturing():
            return torch.cat(list(params), 0)                # never re-lay parameters out inside a capture
        with torch.no_grad():
            flat = torch.cat([p.detach() for p in params], 0)
            r0 = 0
            for p in params:
                p.data = flat[r0:r0 + p.shape[0]]
                r0 += p.shape[0]
        owner.__dict__[key] = flat
    return _AdjacentCatN.apply(flat, *params)


def adjacent_cat(owner, key, a, b):
    """cat([a, b], 0) for two parameters of `owner` without a kernel: their storage is (re-)laid out back to back in
    one buffer the first time (and whenever a `.to()` / `load` has separated them again), after which optimiser updates
    of the parameters are updates of the buffer.  Parameter objects, names and state_dict are untouched."""
    if not shared_parameter_storage:
        return torch.cat([a, b], 0)
    flat = owner.__dict__.get(key)
    n1 = a.numel() * a.element_size()
    if not (flat is not None and flat.dtype == a.dtype == b.dtype and flat.device == a.device == b.device
            and a.data_ptr() == flat.data_ptr() and b.data_ptr() == flat.data_ptr() + n1
            and a.is_contiguous() and b.is_contiguous() and flat.shape[0] == a.shape[0] + b.shape[0]):
        if torch.cuda.is_available() and a.is_cuda and torch.cuda.is_current_stream_capturing():
            return torch.cat([a, b], 0)                      # never re-lay parameters out inside a capture
        with torch.no_grad():
            flat = torch.cat([a.detach(), b.detach()], 0)
            a.data = flat[:a.shape[0]]
            b.data = flat[a.shape[0]:]
        owner.__dict__[key] = flat
    return _AdjacentCat.apply(flat, a, b)


def _is_power_of_2(n):
    if (not isinstance(n, int)) or (n < 0):
        raise ValueError("invalid input for _is_power_of_2: {} (type: {})".format(n, type(n)))
    return (n & (n - 1) == 0) and n != 0


class MSDeformAttn(nn.Module):
    trace = None          # optional callable (module, qproj, reference_points) -> (qproj, reference_points); tests only

    def __init__(self, d_model=256, n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        if d_model % n_heads != 0:
            raise ValueError('d_model must be divisible by n_heads, but got {} and {}'.format(d_model, n_heads))
        if not _is_power_of_2(d_model // n_heads):
            warnings.warn("MSDeformAttn: a head dimension that is not a power of 2 runs on the generic kernel; "
                          "the gfx950 fast paths are specialised for 32 channels per head.")
        self.im2col_step = 64
        self.d_model = d_model
        self.n_levels = n_levels
        self.n_heads = n_heads
        self.n_points = n_points

        self.sampling_offsets = nn.Linear(d_model, n_heads * n_levels * n_points * 2)
        self.attention_weights = nn.Linear(d_model, n_heads * n_levels * n_points)
        self.value_proj = nn.Linear(d_model, d_model)
        self.output_proj = nn.Linear(d_model, d_model)
        self._reset_parameters()

    def _reset_parameters(self):
        # reference ms_deform_attn.py:66-80: zero offset weights, bias = per-head compass direction
        # scaled by the point index; uniform attention; xavier value / output projections
        constant_(self.sampling_offsets.weight.data, 0.)
        thetas = torch.arange(self.n_heads, dtype=torch.float32) * (2.0 * math.pi / self.n_heads)
        grid = torch.stack([thetas.cos(), thetas.sin()], -1)
        grid = (grid / grid.abs().max(-1, keepdim=True)[0]).view(self.n_heads, 1, 1, 2)
        grid = grid.repeat(1, self.n_levels, self.n_points, 1)
        grid = grid * torch.arange(1, self.n_points + 1, dtype=torch.float32).view(1, 1, -1, 1)
        with torch.no_grad():
            self.sampling_offsets.bias = nn.Parameter(grid.reshape(-1))
        constant_(self.attention_weights.weight.data, 0.)
        constant_(self.attention_weights.bias.data, 0.)
        xavier_uniform_(self.value_proj.weight.data)
        constant_(self.value_proj.bias.data, 0.)
        xavier_uniform_(self.output_proj.weight.data)
        constant_(self.output_proj.bias.data, 0.)

    def forward(self, query, reference_points, input_flatten, input_spatial_shapes, input_level_start_index,
                input_padding_mask=None, value_grad_link=None):
        """query [N, Lq, C]; reference_points [N, Lq, L, 2|4] in [0, 1]; input_flatten [N, S, C];
        input_spatial_shapes int64 [L, 2]; input_level_start_index int64 [L]; input_padding_mask
        [N, S] (True = padding).  Returns [N, Lq, C]."""
        N, Len_q, _ = query.shape
        N, Len_in, _ = input_flatten.shape
        M, L, P = self.n_heads, self.n_levels, self.n_points
        # reference ms_deform_attn.py:96 asserts sum(H*W) == Len_in with a device read-back on every call; here
        # the check uses the host copy of the shapes (attached where the pyramid is built, else read back once
        # per shapes tensor; skipped only while a HIP graph is being captured with nothing cached)
        if input_spatial_shapes.is_cuda:
            hs = msda.host_shapes(input_spatial_shapes)
            if hs is not None:
                assert sum(hs[2 * k] * hs[2 * k + 1] for k in range(len(hs) // 2)) == Len_in
        else:
            assert int((input_spatial_shapes[:, 0] * input_spatial_shapes[:, 1]).sum()) == Len_in

        # (value_grad_link: not part of the reference's signature -- the encoder layer's linked attention block, encoder.py;
        #  or attached to the image memory by linear.shared_input for the decoders' value projections)
        if value_grad_link is None:
            value_grad_link = getattr(input_flatten, "value_grad_link", None)
        stp = (sample_then_project and value_grad_link is None and L == 4 and P == 4
               and Len_q * M <= SAMPLE_THEN_PROJECT_MAX_FRACTION * Len_in and reference_points.shape[-1] in (2, 4))
        value = None
        if not stp:
            value = token_linear(input_flatten, self.value_proj.weight, self.value_proj.bias, grad_link=value_grad_link)
            if input_padding_mask is not None:
                value = value.masked_fill(input_padding_mask[..., None], 0.0)
            value = value.view(N, Len_in, M, self.d_model // M)

        n_off = M * L * P * 2
        # (the two projections' parameters share one buffer: "concatenating" them is free, see adjacent_cat)
        qproj = token_linear(query,
                             adjacent_cat(self, "_qproj_w", self.sampling_offsets.weight, self.attention_weights.weight),
                             adjacent_cat(self, "_qproj_b", self.sampling_offsets.bias, self.attention_weights.bias))
        if MSDeformAttn.trace is not None:
            # test instrumentation (tests/test_modules_gpu.py: the bf16 model against a float32 run that is handed the SAME
            # projection rows / reference points, so that both runs take the same floor() decisions in the sampling)
            qproj, reference_points = MSDeformAttn.trace(self, qproj, reference_points)
        if (qproj.is_cuda and fused_geometry and L == 4 and P == 4 and reference_points.shape[-1] in (2, 4)
                and qproj.dtype in (torch.float32, torch.bfloat16)):
            if stp:
                locations, weights = msda.SamplingGeometryFunction.apply(qproj, reference_points, input_spatial_shapes, M, L, P)
                return self._sampled_projection(input_flatten, input_padding_mask, input_spatial_shapes,
                                                input_level_start_index, locations, weights)
            if (fused_sampling and msda_function is msda.MSDeformAttnFunction and qproj.dtype == value.dtype
                    and not reference_points.requires_grad
                    and msda.fused_supported(value, input_spatial_shapes, reference_points, Len_q, L, P,
                                             torch.is_grad_enabled() and (value.requires_grad or qproj.requires_grad))):
                # geometry + sampling + aggregation in one launch each way: the float32 locations / weights are
                # written once for the backward pass and never read in the forward, their gradients never exist
                output = msda.FusedMSDeformAttnFunction.apply(value, input_spatial_shapes, input_level_start_index,
                                                              qproj, reference_points, self.im2col_step)
                return token_linear(output, self.output_proj.weight, self.output_proj.bias)
            # one HIP kernel instead of view + softmax + divide + add (+ their backward passes)
            locations, weights = msda.SamplingGeometryFunction.apply(qproj, reference_points, input_spatial_shapes,
                                                                     M, L, P)
            if value.dtype == torch.float64:
                locations, weights = locations.double(), weights.double()
            output = msda_function.apply(value, input_spatial_shapes, input_level_start_index, locations, weights,
                                         self.im2col_step)
            return token_linear(output, self.output_proj.weight, self.output_proj.bias)
        # the module's own arithmetic (also what the CPU unit tests exercise)
        offsets = qproj[..., :n_off].reshape(N, Len_q, M, L, P, 2)
        logits = qproj[..., n_off:].reshape(N, Len_q, M, L * P)
        if logits.dtype in (torch.bfloat16, torch.float16):
            logits = logits.float()                    # softmax and sampling geometry stay in float32
        weights = F.softmax(logits, -1).view(N, Len_q, M, L, P)

        if reference_points.shape[-1] == 2:
            normalizer = torch.stack([input_spatial_shapes[..., 1], input_spatial_shapes[..., 0]], -1)
            locations = reference_points[:, :, None, :, None, :] \
                + offsets / normalizer[None, None, None, :, None, :]
        elif reference_points.shape[-1] == 4:
            locations = reference_points[:, :, None, :, None, :2] \
                + offsets / P * reference_points[:, :, None, :, None, 2:] * 0.5
        else:
            raise ValueError(
                'Last dim of reference_points must be 2 or 4, but get {} instead.'.format(reference_points.shape[-1]))
        if input_flatten.dtype == torch.bfloat16:
            locations = locations.float()
        if stp:
            return self._sampled_projection(input_flatten, input_padding_mask, input_spatial_shapes, input_level_start_index,
                                            locations.contiguous(), weights.contiguous())
        output = msda_function.apply(value, input_spatial_shapes, input_level_start_index,
                                     locations.contiguous(), weights.contiguous(), self.im2col_step)
        return self.output_proj(output)

    def _sampled_projection(self, src, padding_mask, shapes, starts, locations, weights):
        """The cross-attention of ms_deform_attn.py:98-118 with sampling and value projection exchanged (both are linear):

            out[q, h] = sum_s a_s bilinear(value_h)(loc_s),   value_h = keep (W_h src + b_h)          (keep = 1 - padding mask)
                      = W_h z[q, h] + b_h z1[q, h],   z[q, h] = sum_s a_s bilinear(keep src)(loc_s)  (all 256 channels),
                                                      z1[q, h] = sum_s a_s bilinear(keep)(loc_s)      (zero padding, masked pixels)

        z and z1 are the SAME op called with one "head" of 256 (1) channels and the (query, head) pairs as its queries --
        `locations.view(N, Lq * M, 1, L, P, 2)` is a view, `src` is the value tensor as it is.  Per decoder layer that replaces
        the [88 892, 256] x [256, 256] value projection (forward, input gradient, weight gradient over all tokens, a 45 MB
        zero-fill) by M products [N * Lq, 256] x [256, 32]; the memory's gradient becomes a scatter of 256-channel rows."""
        N, S, C = src.shape
        M, L, P = self.n_heads, self.n_levels, self.n_points
        Lq = locations.shape[1]
        D = C // M
        if padding_mask is not None:
            keep = (~padding_mask).to(src.dtype)[..., None]
            src = src * keep
        else:
            keep = self.__dict__.get("_ones")
            if keep is None or keep.shape[:2] != (N, S) or keep.dtype != src.dtype or keep.device != src.device:
                keep = self.__dict__["_ones"] = torch.ones(N, S, 1, dtype=src.dtype, device=src.device)
        loc1 = locations.reshape(N, Lq * M, 1, L, P, 2)
        aw1 = weights.reshape(N, Lq * M, 1, L, P)
        # (the memory's rows: on the GPU the backward of this call has its own kernels, csrc/msda_rows.hip)
        rows_function = msda.SampleRowsFunction if (msda_function is msda.MSDeformAttnFunction and src.is_cuda) else msda_function
        z = rows_function.apply(src.reshape(N, S, 1, C), shapes, starts, loc1, aw1, self.im2col_step)          # [N, Lq M, C]
        z1 = msda_function.apply(keep.reshape(N, S, 1, 1), shapes, starts, loc1, aw1, self.im2col_step)        # [N, Lq M, 1]
        w = self.value_proj.weight.view(M, D, C)                                                                # rows h D .. of W
        out = torch.einsum("nqhc,hdc->nqhd", z.view(N, Lq, M, C), w) + z1.view(N, Lq, M, 1) * self.value_proj.bias.view(M, D)
        return token_linear(out.reshape(N, Lq, C), self.output_proj.weight, self.output_proj.bias)
