"""Fused residual add + LayerNorm for 256-channel token tensors (csrc/add_layernorm.hip, C ABI
include/rlipv2_norm.h): `norm(src + branch)` of the post-norm encoder layer (reference
models/dab_deformable/deformable_transformer.py:1261-1300) as one HBM pass per direction, and of the decoder
layers (:1346-1401), where the [4, 150..300, 256] tensors are launch-bound: 1 launch instead of add + layer_norm
forward, 2 instead of 3 backward (-0.17 ms per train step, A/B on one box).

Other widths (the 768-channel text layers) and float32 autocast runs keep PyTorch's add + layer_norm.
`norm.MIN_ROWS` is the row threshold (256).
"""
from __future__ import annotations


import torch
import torch.nn.functional as F

from . import _lib, roofline

MIN_ROWS = 256
enabled = True

_workspaces = {}


def _workspace(device, nbytes):
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _workspaces[key] = buf
    return buf


def supported(a, b, weight, bias) -> bool:
    if not (enabled and a.is_cuda and a.dtype == torch.bfloat16 and weight is not None and bias is not None
            and weight.dtype == torch.bfloat16 and bias.dtype == torch.bfloat16):
        return False
    if b is not None and (b.shape != a.shape or b.dtype != a.dtype):
        return False
    rows = a.numel() // a.shape[-1]
    return rows >= MIN_ROWS and bool(_lib.lib().add_layernorm_supported(rows, a.shape[-1]))


def _check(st, what):
    if st:
        raise RuntimeError(f"{what}: " + _lib.strerror(st))


class GradLink:
    """Python-side hand-over between two backward nodes of one residual block (see linear.ffn_residual_norm): the
    LayerNorm's backward leaves the gradient tensor it returns for BOTH addends here, the branch's backward recognises it
    as its own incoming gradient and accumulates the branch's input gradient into it in place.  `first_creates`: no
    LayerNorm in the block (linear.shared_input) -- the first Linear's input gradient becomes the accumulator."""
    dx = None
    first_creates = False
    broken = False          # a node could not use the accumulator: from then on every node returns its gradient normally


class AddLayerNormFunction(torch.autograd.Function):
    """y = LayerNorm(a + b); the gradient of a and of b is the same tensor."""

    @staticmethod
    def forward(ctx, a, b, weight, bias, eps, link=None):
        ctx.link = link
        if not a.is_cuda:
            raise RuntimeError("Not implemented on the CPU")
        a = a.contiguous()
        b = None if b is None else b.contiguous()
        C = a.shape[-1]
        rows = a.numel() // C
        y = torch.empty_like(a)
        mean = torch.empty(rows, dtype=torch.float32, device=a.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=a.device)
        L = _lib.lib()
        _check(L.add_layernorm_forward_bf16(a.data_ptr(), None if b is None else b.data_ptr(), weight.data_ptr(),
                                            bias.data_ptr(), rows, C, float(eps), y.data_ptr(), mean.data_ptr(),
                                            rstd.data_ptr(), torch.cuda.current_stream(a.device).cuda_stream),
               "add_layernorm_forward")
        roofline.add(roofline.tensor_bytes(a, b, weight, bias, y, mean, rstd))
        ctx.save_for_backward(a, b, weight, mean, rstd)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        a, b, weight, mean, rstd = ctx.saved_tensors
        dy = dy.contiguous()
        C = a.shape[-1]
        rows = a.numel() // C
        L = _lib.lib()
        dx = torch.empty_like(a)
        dgamma = torch.empty_like(weight)
        dbeta = torch.empty_like(weight)
        ws = _workspace(a.device, L.add_layernorm_workspace_bytes(rows, C))
        _check(L.add_layernorm_backward_bf16(dy.data_ptr(), a.data_ptr(), None if b is None else b.data_ptr(),
                                             weight.data_ptr(), mean.data_ptr(), rstd.data_ptr(), rows, C,
                                             dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), ws.data_ptr(),
                                             ws.numel(), torch.cuda.current_stream(a.device).cuda_stream),
               "add_layernorm_backward")
        roofline.add(roofline.tensor_bytes(dy, a, b, weight, mean, rstd, dx, dgamma, dbeta))
        if ctx.link is not None:
            ctx.link.dx = dx
        return dx, (dx if b is not None else None), dgamma, dbeta, None, None


def add_layer_norm(a, b, norm: torch.nn.LayerNorm):
    """norm(a + b) (b may be None) with the module's weight / bias / eps."""
    if supported(a, b, norm.weight, norm.bias) and len(norm.normalized_shape) == 1:
        return AddLayerNormFunction.apply(a, b, norm.weight, norm.bias, norm.eps, None)
    return norm(a if b is None else a + b)


# ---- LayerNorm at the Swin widths, with the pre-norm block's residual add (csrc/layernorm_wide.hip) ---------------------------
# GPU-only route (never run on hardware): OFF until rlipv2_amd/routes.validate() has compared the Swin train step with it
# against the plain ops (`add` + `F.layer_norm`) on the caller's own model and batch.
fused_wide_layer_norm = False


def _on_device(t) -> bool:
    """the kernels take device pointers (the lane-level model of tools/emu/ replaces this in tests/test_swin_fused_emulated.py)"""
    return t.is_cuda


def wide_supported(x, norm) -> bool:
    """the fused route can take `norm(x)`: CUDA bfloat16, a supported width, frozen affine parameters (the reference freezes
    every norm of its Swin backbones, models/swin/backbone.py:66-69; trainable ones keep PyTorch's op and its dgamma / dbeta)"""
    w, b = norm.weight, norm.bias
    if not (fused_wide_layer_norm and enabled and _on_device(x) and x.dtype == torch.bfloat16 and w is not None and b is not None
            and w.dtype == torch.bfloat16 and b.dtype == torch.bfloat16 and len(norm.normalized_shape) == 1
            and not w.requires_grad and not b.requires_grad and not torch.is_autocast_enabled()):
        return False
    C = x.shape[-1]
    return norm.normalized_shape[0] == C and bool(_lib.lib().layernorm_wide_supported(x.numel() // C, C))


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream if t.is_cuda else None


def _wide_forward(a, b, weight, bias, eps):
    C = a.shape[-1]
    rows = a.numel() // C
    y = torch.empty_like(a)
    s = torch.empty_like(a) if b is not None else None
    mean = torch.empty(rows, dtype=torch.float32, device=a.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=a.device)
    _check(_lib.lib().layernorm_wide_forward_bf16(a.data_ptr(), None if b is None else b.data_ptr(), weight.data_ptr(),
                                                  bias.data_ptr(), rows, C, float(eps), y.data_ptr(),
                                                  None if b is None else s.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                                  _stream(a)), "layernorm_wide_forward")
    roofline.add(roofline.tensor_bytes(a, b, weight, bias, y, s, mean, rstd))
    return s, y, mean, rstd


def _wide_backward(dy, ds, x, weight, mean, rstd):
    C = x.shape[-1]
    dx = torch.empty_like(x)
    _check(_lib.lib().layernorm_wide_backward_bf16(dy.data_ptr(), None if ds is None else ds.data_ptr(), x.data_ptr(),
                                                   weight.data_ptr(), mean.data_ptr(), rstd.data_ptr(), x.numel() // C, C,
                                                   dx.data_ptr(), _stream(x)), "layernorm_wide_backward")
    roofline.add(roofline.tensor_bytes(dy, ds, x, weight, mean, rstd, dx))
    return dx


class WideLayerNormFunction(torch.autograd.Function):
    """y = LN(a) at a Swin width (frozen affine parameters): one pass per direction."""

    @staticmethod
    def forward(ctx, a, weight, bias, eps):
        a = a.contiguous()
        _, y, mean, rstd = _wide_forward(a, None, weight, bias, eps)
        ctx.save_for_backward(a, weight, mean, rstd)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        a, weight, mean, rstd = ctx.saved_tensors
        return _wide_backward(dy.contiguous(), None, a, weight, mean, rstd), None, None, None


class WideAddLayerNormFunction(torch.autograd.Function):
    """(s, y) = (a + b, LN(a + b)) for the pre-norm residual block; the backward takes the gradients of BOTH outputs and
    returns dLN(dy) + ds for a and for b (the same tensor) -- one pass per direction where add + layer_norm are five."""

    @staticmethod
    def forward(ctx, a, b, weight, bias, eps):
        s, y, mean, rstd = _wide_forward(a.contiguous(), b.contiguous(), weight, bias, eps)
        ctx.save_for_backward(s, weight, mean, rstd)
        return s, y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, ds, dy):
        s, weight, mean, rstd = ctx.saved_tensors
        if dy is None:                             # only the sum was used downstream
            return ds, ds, None, None, None
        dx = _wide_backward(dy.contiguous(), None if ds is None else ds.contiguous(), s, weight, mean, rstd)
        return dx, dx, None, None, None


def residual_pre_norm(x, branch, norm: torch.nn.LayerNorm):
    """(x + branch, norm(x + branch)) -- the step between two sub-blocks of a pre-norm transformer block (reference
    models/swin/swin_transformer.py:386-401: `x = shortcut + drop_path(x)` followed by the next `norm(x)`).  `branch` None:
    (x, norm(x)).  One fused pass per direction on the GPU when the width is supported, the plain ops otherwise."""
    if wide_supported(x, norm):
        if branch is None:
            return x, WideLayerNormFunction.apply(x, norm.weight, norm.bias, norm.eps)
        if branch.shape == x.shape and branch.dtype == x.dtype:
            return WideAddLayerNormFunction.apply(x, branch, norm.weight, norm.bias, norm.eps)
    s = x if branch is None else x + branch
    return s, norm(s)


# ---- GroupNorm(32, 256) of the token-major feature pyramid (csrc/groupnorm_tokens.hip, include/rlipv2_groupnorm.h) ----
level_group_norm_enabled = True


def _ptr_array(tensors):
    import ctypes
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def _int_array(values):
    import ctypes
    return (ctypes.c_int * len(values))(*values)


def level_group_norm_supported(xs, norms) -> bool:
    if not (level_group_norm_enabled and 1 <= len(xs) <= 4 and len(xs) == len(norms)):
        return False
    for x, gn in zip(xs, norms):
        if not (isinstance(gn, torch.nn.GroupNorm) and gn.affine and x.is_cuda and x.dim() == 3
                and x.dtype == torch.bfloat16 and gn.weight.dtype == torch.bfloat16 and gn.bias.dtype == torch.bfloat16
                and x.shape[0] == xs[0].shape[0] and gn.eps == norms[0].eps
                and _lib.lib().groupnorm_tokens_supported(x.shape[-1], gn.num_groups, len(xs))):
            return False
    return not torch.is_autocast_enabled()


class LevelGroupNormFunction(torch.autograd.Function):
    """(x_0 .. x_{L-1}, gamma_0, beta_0, .. ) -> [N, sum(hw_l), 256]: every level normalised into its slice of the
    flattened pyramid (reference: input_proj's GroupNorm, models/hoi.py:1936-1957, + the flatten / cat of
    models/dab_deformable/deformable_transformer.py:520-547)."""

    @staticmethod
    def forward(ctx, eps, n_levels, *args):
        xs = [a.contiguous() for a in args[:n_levels]]
        gammas = list(args[n_levels:2 * n_levels])
        betas = list(args[2 * n_levels:3 * n_levels])
        if not xs[0].is_cuda:
            raise RuntimeError("Not implemented on the CPU")
        N = xs[0].shape[0]
        hw = [x.shape[1] for x in xs]
        L = _lib.lib()
        out = torch.empty(N, sum(hw), xs[0].shape[2], dtype=xs[0].dtype, device=xs[0].device)
        mean = torch.empty(n_levels, N, 32, dtype=torch.float32, device=out.device)
        rstd = torch.empty_like(mean)
        hw_c = _int_array(hw)
        ws = _workspace(out.device, L.groupnorm_tokens_workspace_bytes(N, hw_c, n_levels))
        _check(L.groupnorm_tokens_forward_bf16(_ptr_array(xs), hw_c, n_levels, N, _ptr_array(gammas), _ptr_array(betas),
                                               float(eps), out.data_ptr(), mean.data_ptr(), rstd.data_ptr(), ws.data_ptr(),
                                               ws.numel(), torch.cuda.current_stream(out.device).cuda_stream),
               "groupnorm_tokens_forward")
        roofline.add(2 * roofline.tensor_bytes(*xs) + roofline.tensor_bytes(out))
        ctx.save_for_backward(mean, rstd, *xs, *gammas)
        ctx.n_levels = n_levels
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        n = ctx.n_levels
        mean, rstd = ctx.saved_tensors[:2]
        xs = list(ctx.saved_tensors[2:2 + n])
        gammas = list(ctx.saved_tensors[2 + n:2 + 2 * n])
        dy = dy.contiguous()
        N = xs[0].shape[0]
        hw = [x.shape[1] for x in xs]
        L = _lib.lib()
        dxs = [torch.empty_like(x) for x in xs]
        dgs = [torch.empty_like(g) for g in gammas]
        dbs = [torch.empty_like(g) for g in gammas]
        hw_c = _int_array(hw)
        ws = _workspace(dy.device, L.groupnorm_tokens_workspace_bytes(N, hw_c, n))
        _check(L.groupnorm_tokens_backward_bf16(dy.data_ptr(), _ptr_array(xs), hw_c, n, N, _ptr_array(gammas),
                                                mean.data_ptr(), rstd.data_ptr(), _ptr_array(dxs), _ptr_array(dgs),
                                                _ptr_array(dbs), ws.data_ptr(), ws.numel(),
                                                torch.cuda.current_stream(dy.device).cuda_stream),
               "groupnorm_tokens_backward")
        roofline.add(2 * roofline.tensor_bytes(dy, *xs) + roofline.tensor_bytes(*dxs))
        return (None, None, *dxs, *dgs, *dbs)


def level_group_norm(xs, norms):
    """[N, hw_l, 256] token-major projections of the pyramid levels + their GroupNorm modules -> the flattened,
    normalised [N, sum(hw_l), 256] tensor the encoder consumes."""
    return LevelGroupNormFunction.apply(norms[0].eps, len(xs), *xs, *[gn.weight for gn in norms],
                                        *[gn.bias for gn in norms])
