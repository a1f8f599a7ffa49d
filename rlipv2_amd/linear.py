"""Linear layers applied to the flattened multi-scale feature map ([N, S, C] with N*S = 88 892 tokens at
batch 4): forward and input gradient stay library GEMMs, the weight + bias gradient -- a GEMM that
reduces over the tokens into a small [out, in] square, which the library runs at a few percent of the
HBM rate -- is the hand-written MFMA kernel of csrc/token_gemm.hip (C ABI: include/rlipv2_linear.h).

Used by MSDeformAttn (value_proj, the fused sampling_offsets + attention_weights projection, output_proj;
reference models/ops/modules/ms_deform_attn.py:59-62) and by the encoder layer's FFN (linear1 / linear2,
reference models/dab_deformable/deformable_transformer.py:571-576).
"""
from __future__ import annotations

import ctypes
import os

import torch
import torch.nn.functional as F

from . import _lib, roofline

# rows below this go to the library (its weight-gradient GEMM is fine when the reduction is short)
MIN_ROWS = 256
enabled = True

_workspaces = {}


TUNED_TABLE = os.environ.get("RLIPV2_TUNED_GEMM_TABLE") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned",
                                                                         "gemm_gfx950.csv")


def use_tuned_library_gemms(path: str = TUNED_TABLE) -> bool:
    """Lookup-only use of the hipBLASLt solution table recorded by `tools/tune_gemms.py` for the token-major
    library GEMMs (forward / input gradient of the big Linears; the default heuristic is 10-35 % slower on
    them).  Nothing is tuned at run time and nothing is written; a table recorded for other library versions
    is rejected by PyTorch's validator and the defaults stay in force.  RLIPV2_TUNED_GEMMS=0 disables it."""
    if os.environ.get("RLIPV2_TUNED_GEMMS", "1") == "0" or not os.path.exists(path):
        return False
    import torch.cuda.tunable as tunable
    tunable.enable(True)
    tunable.tuning_enable(False)
    tunable.record_untuned_enable(False)
    ok = bool(tunable.read_file(path))
    import tempfile
    tunable.set_filename(os.path.join(tempfile.gettempdir(), "rlipv2_tunableop_%d.csv" % os.getpid()))   # exit-time dump: out of the tree
    return ok


def _workspace(device, nbytes):
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 26), dtype=torch.uint8, device=device)
        _workspaces[key] = buf
    return buf


# Feature widths that are multiples of 64 but not of 128 (stage 0 of the Swin backbones: 192 / 576 channels) are not shapes of the
# weight-gradient kernel (128 x 128 tiles); with this switch the narrow operand is zero-padded to the next multiple of 128 (one
# extra pass over a [T, 192]-sized tensor) instead of leaving the product to the library GEMM, which runs these token-reducing
# shapes at 4-9 % of the HBM rate.  OFF: not timed; `bench.py --set linear.pad_wgrad_to_128=1`.
pad_wgrad_to_128 = False
PAD_WGRAD_MIN_ROWS = 16384


def _padded_width(n: int) -> int:
    return (n + 127) // 128 * 128


def supported(x: torch.Tensor, weight: torch.Tensor) -> bool:
    if not (enabled and x.is_cuda and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16):
        return False
    rows = x.numel() // x.shape[-1]
    M, K = weight.shape[0], weight.shape[1]
    if rows >= MIN_ROWS and bool(_lib.lib().linear_wgrad_supported(rows, M, K)):
        return True
    return (pad_wgrad_to_128 and rows >= PAD_WGRAD_MIN_ROWS and M % 64 == 0 and K % 64 == 0 and M >= 64 and K >= 64
            and bool(_lib.lib().linear_wgrad_supported(rows, _padded_width(M), _padded_width(K))))


def _wgrad_call(dy: torch.Tensor, x: torch.Tensor, with_bias: bool, out_dtype):
    """the kernel on contiguous [T, M] / [T, K] operands of a supported shape"""
    T, M = dy.shape
    K = x.shape[1]
    L = _lib.lib()
    nbytes = L.linear_wgrad_workspace_bytes(T, M, K)
    if nbytes == 0:
        raise RuntimeError(f"linear_wgrad: unsupported shape T={T} M={M} K={K} (M, K must be multiples of 128)")
    ws = _workspace(dy.device, nbytes)
    f32 = out_dtype == torch.float32
    dw = torch.empty(M, K, dtype=out_dtype, device=dy.device)
    db = torch.empty(M, dtype=out_dtype, device=dy.device) if with_bias else None
    st = L.linear_wgrad_bf16(dy.data_ptr(), x.data_ptr(), T, M, K, dw.data_ptr(),
                             db.data_ptr() if with_bias else None, int(f32), ws.data_ptr(), ws.numel(),
                             torch.cuda.current_stream(dy.device).cuda_stream)
    if st:
        raise RuntimeError("linear_wgrad: " + _lib.strerror(st))
    roofline.add(roofline.tensor_bytes(dy, x, dw, db), 2 * T * M * K)
    return dw, db


def linear_wgrad(dy: torch.Tensor, x: torch.Tensor, with_bias: bool = True, out_dtype=torch.bfloat16):
    """dW [M, K] = dy[T, M]^T x[T, K] and db [M] = dy.sum(0) (bf16 inputs, float32 accumulation)."""
    if not (dy.is_cuda and x.is_cuda):
        raise RuntimeError("Not implemented on the CPU")
    if dy.dtype != torch.bfloat16 or x.dtype != torch.bfloat16:
        raise RuntimeError("linear_wgrad: bfloat16 operands expected")
    dy = dy.reshape(-1, dy.shape[-1]).contiguous()
    x = x.reshape(-1, x.shape[-1]).contiguous()
    if x.shape[0] != dy.shape[0]:
        raise RuntimeError("linear_wgrad: dy and x disagree on the number of rows")
    return _wgrad_maybe_padded(dy, x, with_bias, out_dtype)


def _wgrad_maybe_padded(dy, x, with_bias, out_dtype):
    M, K = dy.shape[1], x.shape[1]
    Mp, Kp = (_padded_width(M), _padded_width(K)) if pad_wgrad_to_128 else (M, K)
    if (Mp, Kp) == (M, K):
        return _wgrad_call(dy, x, with_bias, out_dtype)
    # zero columns contribute zero rows / columns of dW (and zero entries of db): the result is the top-left block
    if Mp != M:
        dy = F.pad(dy, (0, Mp - M))
    if Kp != K:
        x = F.pad(x, (0, Kp - K))
    dw, db = _wgrad_call(dy, x, with_bias, out_dtype)
    return dw[:M, :K].contiguous(), (db[:M].contiguous() if db is not None else None)


EXPAND_MIN_ROWS = 4096      # below this the library GEMM + elementwise pair is launch-bound anyway


def expand_supported(a: torch.Tensor, n: int) -> bool:
    return (enabled and a.is_cuda and a.dtype == torch.bfloat16
            and bool(_lib.lib().linear_expand_supported(a.numel() // a.shape[-1], n, a.shape[-1])))


def expand_gemm(a: torch.Tensor, b: torch.Tensor, bias=None, mask=None, relu: bool = False) -> torch.Tensor:
    """epilogue(a[..., 256] @ b[N, 256]^T) -> [..., N] (bf16, float32 accumulation) on the hand-written MFMA kernel
    `csrc/expand_gemm.hip`: bias and ReLU applied in the workgroup, `mask` (a saved ReLU output, shape [..., N])
    zeroes the result where mask <= 0 -- `threshold_backward` without its own pass over the [T, N] tensor."""
    if not a.is_cuda:
        raise RuntimeError("Not implemented on the CPU")
    if a.dtype != torch.bfloat16 or b.dtype != torch.bfloat16:
        raise RuntimeError("expand_gemm: bfloat16 operands expected")
    K = a.shape[-1]
    a2 = a.reshape(-1, K).contiguous()
    b = b.contiguous()
    T, N = a2.shape[0], b.shape[0]
    if b.shape[1] != K:
        raise RuntimeError("expand_gemm: a and b disagree on the reduction width")
    if bias is not None:
        bias = bias.to(torch.bfloat16).contiguous()
    if mask is not None:
        if mask.dtype != torch.bfloat16 or mask.numel() != T * N:
            raise RuntimeError("expand_gemm: mask must be bfloat16 of the output's shape")
        mask = mask.contiguous()
    c = torch.empty(T, N, dtype=torch.bfloat16, device=a.device)
    st = _lib.lib().linear_expand_bf16(a2.data_ptr(), b.data_ptr(), bias.data_ptr() if bias is not None else None,
                                       mask.data_ptr() if mask is not None else None, T, N, K, int(relu), c.data_ptr(),
                                       torch.cuda.current_stream(a.device).cuda_stream)
    if st:
        raise RuntimeError(f"expand_gemm: {_lib.strerror(st)} (T={T} N={N} K={K})")
    roofline.add(roofline.tensor_bytes(a2, b, bias, mask, c), 2 * T * N * K)
    return c.view(*a.shape[:-1], N)


class FusedFFNFunction(torch.autograd.Function):
    """linear2(relu(linear1(x))) of the encoder layer (`forward_ffn`, dab_deformable/deformable_transformer.py:1285-1289,
    dropout 0) as ONE autograd node, so that the backward can hand the ReLU mask to the GEMM that produces the
    hidden gradient: dh = (dy W2) * (h > 0) leaves `expand_gemm` already masked.  Saves per layer and step the
    1.1 GB `threshold_backward` pass over the [88 892, 2048] tensors; both weight gradients on `token_gemm.hip`."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, link=None):
        # forward stays on the library (bias + ReLU in the hipBLASLt epilogue, tuned solution): measured 135-145 us
        # against 150 us for expand_gemm(bias, relu) at N = 2048 -- the fused mask is where the own kernel pays
        h = _linear_forward(x, w1, b1, True)
        y = F.linear(h, w2, b2)
        ctx.save_for_backward(x, w1, w2, h)
        ctx.link = link
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, w1, w2, h = ctx.saved_tensors
        link = ctx.link
        # residual block (ffn_residual_norm): `dy` is the very tensor the LayerNorm's backward also returned as the gradient of
        # the residual input, i.e. of `x` -- the block's d_x = dy + dh W1 is then ONE GEMM accumulating into it (beta = 1)
        # instead of a GEMM + autograd's sum of two 45 MB tensors
        shared = (link is not None and link.dx is not None and dy.is_contiguous() and ctx.needs_input_grad[0]
                  and (link.dx is dy or (link.dx.data_ptr() == dy.data_ptr() and link.dx.shape == dy.shape
                                         and link.dx.stride() == dy.stride() and link.dx.dtype == dy.dtype)))
        dy = dy.contiguous()
        dw2, db2 = linear_wgrad(dy, h, with_bias=True, out_dtype=w2.dtype)
        dh = expand_gemm(dy, w2.t(), mask=h)                     # w2.t().contiguous(): 1 MB, inside expand_gemm
        dw1, db1 = linear_wgrad(dh, x, with_bias=True, out_dtype=w1.dtype)
        if shared:
            dy.view(-1, dy.shape[-1]).addmm_(dh.view(-1, dh.shape[-1]), w1)       # (after its last use as dy above)
            dx = None                                            # delivered through the residual's gradient
        else:
            dx = dh.matmul(w1) if ctx.needs_input_grad[0] else None
        return dx, dw1, db1, dw2, db2, None


class _Alias(torch.autograd.Function):
    """x as a new autograd node: whatever consumes the alias is known to the code that made it (exactly the block's branch
    nodes and its residual), so the node's gradient buffer holds the LayerNorm's tensor by reference until all of them ran.
    Its backward runs last and drops the link's reference."""

    @staticmethod
    def forward(ctx, x, link=None):
        ctx.link = link
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        if ctx.link is not None:
            ctx.link.dx = None
        return g, None


class _AddInto(torch.autograd.Function):
    """x + pos (the query of the encoder's self-attention, deformable_transformer.py:1291) for a linked block: the gradient of
    `x` is added into the link's tensor in place (the same add autograd would do, but the buffer stays the one every node of
    the block accumulates into); `pos` receives the gradient as usual."""

    @staticmethod
    def forward(ctx, x, pos, link):
        ctx.link = link
        return x + pos

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        acc = ctx.link.dx
        g_pos = g if ctx.needs_input_grad[1] else None
        if acc is not None and acc.shape == g.shape and acc.dtype == g.dtype:
            acc.add_(g)
            return None, g_pos, None
        if acc is not None:
            ctx.link.dx, ctx.link.broken = None, True            # (see TokenLinearFunction.backward)
        return g, g_pos, None


def _ffn_block_ok(x, linear1, linear2, norm):
    from . import norm as N
    w1, b1, w2, b2 = linear1.weight, linear1.bias, linear2.weight, linear2.bias
    T = x.numel() // x.shape[-1]
    return (not torch.is_autocast_enabled() and torch.is_grad_enabled() and b1 is not None and b2 is not None
            and w1.requires_grad and w2.requires_grad and x.requires_grad and T >= EXPAND_MIN_ROWS
            and x.dtype == w1.dtype == w2.dtype and expand_supported(x, w1.shape[0]) and supported(x, w1)
            and w2.shape[0] % 128 == 0 and w1.shape[0] % 128 == 0 and fused_ffn_enabled
            and len(norm.normalized_shape) == 1 and N.supported(x, x, norm.weight, norm.bias))


fused_ffn_enabled = True          # (attributes: flipped by tests / `bench.py --set linear.fused_ffn_enabled=0` for A/B runs)
row_vector_gemm = True
# GPU-only route, OFF until rlipv2_amd/routes.validate() has compared it with the plain nodes on the caller's own step
residual_gradient_in_gemm = False


def shared_input(x):
    """`x` for several token-major Linears that all read it (the image memory under the six decoder layers' value projections,
    deformable_transformer.py:1346-1401): their input gradients d_l W_l are summed by the GEMMs themselves -- the first one's
    result becomes the accumulator, the others add into it (beta = 1) -- instead of five autograd sums of 45 MB tensors.  The
    link travels as an attribute of the returned alias (`value_grad_link`, read by MSDeformAttn.forward).  The alias must be consumed ONLY by such Linears: any other
    consumer's gradient would be summed by autograd into a new tensor and later in-place contributions would be lost."""
    from . import norm as N
    if not (residual_gradient_in_gemm and enabled and torch.is_grad_enabled() and x.requires_grad and x.is_cuda
            and x.dtype == torch.bfloat16 and not torch.is_autocast_enabled()
            and x.numel() // x.shape[-1] >= MIN_ROWS):          # (the route every consumer below will take: TokenLinearFunction)
        return x
    link = N.GradLink()
    link.first_creates = True
    xa = _Alias.apply(x, link)
    xa.value_grad_link = link
    return xa


def attention_block_link_ok(src, pos, value_weight, norm):
    """the encoder layer's attention block can run linked (encoder.py): fused add + LayerNorm and the token-major Linear route
    for the value projection, same shapes throughout"""
    from . import norm as N
    return (residual_gradient_in_gemm and pos is not None and pos.shape == src.shape and pos.dtype == src.dtype
            and torch.is_grad_enabled() and src.requires_grad and not torch.is_autocast_enabled()
            and supported(src, value_weight) and len(norm.normalized_shape) == 1
            and N.supported(src, src, norm.weight, norm.bias))


def ffn_residual_norm(x, linear1, linear2, norm):
    """norm(x + linear2(relu(linear1(x)))) of the post-norm encoder layer (`forward_ffn`,
    dab_deformable/deformable_transformer.py:1285-1289, dropout 0).  With the fused FFN node and the fused add + LayerNorm
    the two backward nodes are linked: the LayerNorm returns ONE gradient tensor for both of its addends, the FFN's
    input-gradient GEMM accumulates into that tensor (beta = 1) and returns nothing of its own -- the sum autograd would
    form (d_x = d_residual + d_branch, 136 MB of traffic per layer) is the GEMM's epilogue."""
    from . import norm as N
    if residual_gradient_in_gemm and _ffn_block_ok(x, linear1, linear2, norm):
        link = N.GradLink()
        xa = _Alias.apply(x, link)
        y = FusedFFNFunction.apply(xa, linear1.weight, linear1.bias, linear2.weight, linear2.bias, link)
        return N.AddLayerNormFunction.apply(xa, y, norm.weight, norm.bias, norm.eps, link)
    return N.add_layer_norm(x, fused_ffn(x, linear1, linear2), norm)


def fused_ffn(x, linear1, linear2):
    """linear2(relu(linear1(x))) -- fused node when the shapes are the encoder's (token-major, 256 -> N -> 256,
    bf16 parameters, training), the two `token_linear` calls otherwise."""
    w1, b1, w2, b2 = linear1.weight, linear1.bias, linear2.weight, linear2.bias
    T = x.numel() // x.shape[-1]
    if (not torch.is_autocast_enabled() and torch.is_grad_enabled() and b1 is not None and b2 is not None
            and w1.requires_grad and w2.requires_grad and T >= EXPAND_MIN_ROWS and x.dtype == w1.dtype == w2.dtype
            and expand_supported(x, w1.shape[0]) and supported(x, w1)
            and w2.shape[0] % 128 == 0 and w1.shape[0] % 128 == 0 and fused_ffn_enabled):
        return FusedFFNFunction.apply(x, w1, b1, w2, b2, None)
    return token_linear(token_linear(x, w1, b1, relu=True), w2, b2)


def _linear_forward(x, weight, bias, relu):
    if relu and bias is not None:
        # bias + ReLU in the GEMM epilogue (hipBLASLt): no separate activation pass over the output
        y = torch._addmm_activation(bias, x.reshape(-1, x.shape[-1]), weight.t())
        return y.view(*x.shape[:-1], weight.shape[0])
    y = F.linear(x, weight, bias)
    return F.relu(y) if relu else y


class TokenLinearFunction(torch.autograd.Function):
    """`link` (norm.GradLink, attention block of the encoder layer -- encoder.py): the input `x` is the residual input of
    the block's LayerNorm, whose backward has already left the residual's gradient tensor in the link; the input gradient
    dy W is then accumulated INTO that tensor (one GEMM with beta = 1) and nothing is returned for `x`."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu=False, link=None):
        y = _linear_forward(x, weight, bias, relu)
        ctx.relu = relu
        ctx.link = link
        ctx.save_for_backward(x, weight, y if relu else None)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        if ctx.relu:
            dy = torch.ops.aten.threshold_backward(dy, y, 0)
        dx = None
        if ctx.needs_input_grad[0]:
            link = ctx.link
            acc = link.dx if link is not None else None
            if acc is not None and acc.shape == x.shape and acc.dtype == dy.dtype and acc.is_contiguous():
                # once a link has an accumulator EVERY contribution goes into it: a gradient returned the normal way would be
                # summed by autograd out of place, and what later nodes add into the accumulator would be lost
                dyc = dy if dy.is_contiguous() else dy.contiguous()
                acc.view(-1, acc.shape[-1]).addmm_(dyc.reshape(-1, dyc.shape[-1]), weight)
            else:
                dx = dy.matmul(weight)
                if acc is not None:
                    link.dx, link.broken = None, True            # unusable accumulator: the remaining nodes return normally
                elif link is not None and link.first_creates and not link.broken:
                    link.dx = dx                                 # the first consumer's gradient becomes the accumulator
        dw = db = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            dw, db = linear_wgrad(dy, x, with_bias=ctx.needs_input_grad[2], out_dtype=weight.dtype)
        return dx, dw, db, None, None


_ones = {}


def _ones_row(n, like):
    key = (like.device, like.dtype, n)
    t = _ones.get(key)
    if t is None:
        t = _ones[key] = torch.ones(1, n, dtype=like.dtype, device=like.device)
    return t


class SmallLinearFunction(torch.autograd.Function):
    """F.linear for the few-hundred-row inputs of the decoders / text layers.  Same three GEMMs as autograd's
    own backward, except the bias gradient: `dy.sum(0)` runs PyTorch's generic reduction, 19-31 us for a
    [320, 768] input (one thread walks a whole column), ~280 of them per step; as ones[1, R] @ dy it is one
    more small GEMM."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu=False):
        y = _linear_forward(x, weight, bias, relu)
        ctx.relu = relu
        ctx.save_for_backward(x, weight, y if relu else None)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        if ctx.relu:
            dy = torch.ops.aten.threshold_backward(dy, y, 0)
        dy2 = dy.reshape(-1, dy.shape[-1])
        dx = dy.matmul(weight) if ctx.needs_input_grad[0] else None
        dw = dy2.t().mm(x.reshape(-1, x.shape[-1])) if ctx.needs_input_grad[1] else None
        db = _ones_row(dy2.shape[0], dy2).mm(dy2).view(-1) if ctx.needs_input_grad[2] else None
        return dx, dw, db, None


class AddRowVectorFunction(torch.autograd.Function):
    """x[N, T, C] + row[C] whose backward sums the gradient over (N, T) as ones[N, 1, T] @ g (a batched GEMM) instead
    of PyTorch's generic reduction: the level-embedding gradients of the encoder's positional input
    (`lvl_pos_embed = pos_embed + level_embed[lvl]`, dab_deformable/deformable_transformer.py:396-399) cost 150 us per
    level as `sum` ([4, 16700, 256] -> [256]) and ~10 us this way.  The gradient may be a strided slice (it is a
    `narrow` of the concatenated levels): bmm takes the batch stride as it is, no copy."""

    @staticmethod
    def forward(ctx, x, row):
        ctx.x_needs = x.requires_grad
        return x + row.view(1, 1, -1)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        N, T, C = g.shape
        g_row = torch.bmm(_ones_row(T, g).expand(N, 1, T), g).sum(0).view(C) if ctx.needs_input_grad[1] else None
        return (g if ctx.needs_input_grad[0] else None), g_row


def add_row_vector(x, row):
    if x.is_cuda and x.dim() == 3 and torch.is_grad_enabled() and row.requires_grad and x.dtype == row.dtype \
            and not torch.is_autocast_enabled() and row_vector_gemm:
        return AddRowVectorFunction.apply(x, row)
    return x + row.view(1, 1, -1)


def token_linear(x, weight, bias=None, relu=False, grad_link=None):
    """relu?(F.linear(x, weight, bias)) with the MFMA weight-gradient kernel behind it when the shape
    qualifies; `relu=True` puts the activation into the GEMM epilogue.  `grad_link`: see TokenLinearFunction (only
    honoured on that route; every other route returns the input gradient the usual way)."""
    if torch.is_autocast_enabled():
        # autocast runs (float32 parameters, per-op casts): the custom backward paths assume one dtype throughout
        y = F.linear(x, weight, bias)
        return F.relu(y) if relu else y
    if torch.is_grad_enabled() and (weight.requires_grad or x.requires_grad):
        if supported(x, weight):
            return TokenLinearFunction.apply(x, weight, bias, relu, grad_link)
        if (enabled and x.is_cuda and bias is not None and x.dtype == weight.dtype
                and x.dtype in (torch.bfloat16, torch.float32) and (bias.requires_grad or relu)):
            return SmallLinearFunction.apply(x, weight, bias, relu)
    if x.is_cuda:
        return _linear_forward(x, weight, bias, relu)
    y = F.linear(x, weight, bias)
    return F.relu(y) if relu else y


class FastLinear(torch.nn.Linear):
    """nn.Linear (same parameters, same state_dict) whose forward goes through `token_linear`."""

    def forward(self, x):
        return token_linear(x, self.weight, self.bias)


def swap_linears(model):
    """Re-class every plain nn.Linear of `model` to FastLinear (no parameter is touched)."""
    n = 0
    for m in model.modules():
        if type(m) is torch.nn.Linear:
            m.__class__ = FastLinear
            n += 1
    return n
