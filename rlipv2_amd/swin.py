"""Swin Transformer backbone (BASELINE configs 4-5: RLIP_ParSeDA_v2 Swin-L) as the
input producer of the hot path, with the reference's parameter names so that its checkpoints load.

Reference: models/swin/swin_transformer.py (WindowAttention :221-301, SwinTransformerBlock :304-403,
PatchMerging :406-446, BasicLayer :449-550, PatchEmbed :553-593, SwinTransformer :596-763) and
models/swin/backbone.py (BackboneBase :62-97: position tables and every norm are frozen; Backbone :100-169:
the small / base / large presets; features of stages 1-3 at strides 8/16/32).

Re-designed rather than transcribed:
  * activations stay [B, H, W, C] (channels-last token maps) through a stage -- no (B, L, C) <-> (B, H, W, C)
    view ping-pong; the stage outputs are handed on as channels-last NCHW views (no .contiguous() copy);
  * window attention is ONE scaled_dot_product_attention call per block over [B, windows, heads, tokens, d]
    with an additive mask [1, windows, heads, tokens, tokens] = relative-position bias (+ the shift mask);
    since BackboneBase freezes the bias tables, that mask is cached per (stage geometry, table version)
    instead of being re-gathered from the table in every block of every step;
  * the shift masks depend only on the padded stage size: cached per (Hp, Wp).
Pad / roll / window partition are kept as data movement (one copy in, one copy out per block).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import nn

from .blocks import NestedTensor, PositionEmbeddingSine
from .linear import token_linear
from . import norm as _norm
from .norm import residual_pre_norm


class DropPath(nn.Module):
    """stochastic depth: drop the whole residual branch of a sample with probability p (training only)"""

    def __init__(self, p=0.0):
        super().__init__()
        self.p = float(p)

    def forward(self, x):
        if self.p == 0.0 or not self.training:
            return x
        keep = 1.0 - self.p
        # (timm's drop_path, which the reference imports -- models/swin/swin_transformer.py:21: the MASK is scaled by 1 / keep,
        #  then one multiply; `x * mask / keep` was one more pass over the activations forward and backward per residual branch)
        mask = x.new_empty((x.shape[0],) + (1,) * (x.dim() - 1)).bernoulli_(keep).div_(keep)
        return x * mask


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features, drop=0.0):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.fc2 = nn.Linear(hidden_features, in_features)
        self.drop = nn.Dropout(drop)

    def forward(self, x):
        return self.drop(token_linear(self.drop(F.gelu(token_linear(x, self.fc1.weight, self.fc1.bias))),
                                      self.fc2.weight, self.fc2.bias))


def relative_position_index(ws):
    """[ws*ws, ws*ws] index into the (2ws-1)^2 bias table (reference :246-257)"""
    coords = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


def shift_mask(Hp, Wp, ws, shift, device):
    """[windows, ws*ws, ws*ws] additive mask of the cyclically shifted layout: 0 within a region, -100 across
    (reference :517-533)"""
    img = torch.zeros((Hp, Wp), device=device)
    cnt = 0
    for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[hs, wsl] = cnt
            cnt += 1
    win = img.view(Hp // ws, ws, Wp // ws, ws).permute(0, 2, 1, 3).reshape(-1, ws * ws)
    diff = win[:, None, :] - win[:, :, None]
    return torch.where(diff != 0, torch.full_like(diff, -100.0), torch.zeros_like(diff))


# GPU-only route (csrc/window_attention.hip; never run on hardware): OFF until rlipv2_amd/routes.validate() has compared the
# Swin train step with it against the PyTorch op sequence on the caller's own model and batch.
fused_window_attention = False


def _padded(t, n, pad_keys):
    """[..., n, n] (query, key) -> float32 [..., 64, 64]: `pad_keys` in the key columns >= n, 0 elsewhere -- the layout
    include/rlipv2_swin.h asks for (a lane of the kernel owns a query and reads 4 consecutive keys per load)"""
    out = t.new_zeros(*t.shape[:-2], 64, 64, dtype=torch.float32)
    out[..., :, n:] = pad_keys
    out[..., :n, :n] = t.float()
    return out.contiguous()


def compact_masks(mask):
    """[nW, N, N] additive shift masks -> (mask_t [K, 64, 64] float32 of the K distinct non-zero masks, mask_id [nW] int32,
    -1 = no mask): all interior windows of a shifted layout share the zero mask, the right / bottom / corner ones a few more"""
    nW, n = mask.shape[0], mask.shape[-1]
    flat = mask.reshape(nW, -1)
    distinct, inverse = torch.unique(flat, dim=0, return_inverse=True)
    nonzero = distinct.abs().sum(1) != 0
    remap = torch.cumsum(nonzero.to(torch.int32), 0) - 1
    ids = torch.where(nonzero[inverse], remap[inverse], torch.full_like(remap[inverse], -1)).to(torch.int32)
    kept = distinct[nonzero].view(-1, n, n)
    if kept.shape[0] == 0:
        return None, None
    return _padded(kept, n, 0.0), ids.contiguous()


class WindowAttentionFunction(torch.autograd.Function):
    """softmax(scale q k^T + bias (+ mask)) v over all windows and heads of a block: one launch per direction on the packed qkv
    tensor (include/rlipv2_swin.h); the backward recomputes the probabilities, nothing N x N is saved.  The bias table is frozen
    (the reference freezes it, models/swin/backbone.py:66-69) -- no gradient for it here."""

    @staticmethod
    def forward(ctx, qkv, bias_t, mask_t, mask_id, heads, scale):
        from . import _lib
        Bw, nW, N = qkv.shape[0], qkv.shape[1], qkv.shape[2]
        qkv = qkv.contiguous()
        out = torch.empty(Bw, nW, N, heads * 32, dtype=qkv.dtype, device=qkv.device)
        st = _lib.lib().window_attention_forward_bf16(
            qkv.data_ptr(), bias_t.data_ptr(), None if mask_t is None else mask_t.data_ptr(),
            None if mask_id is None else mask_id.data_ptr(), Bw * nW, nW, heads, N, float(scale), out.data_ptr(),
            _norm._stream(qkv))
        if st:
            raise RuntimeError("window_attention_forward: " + _lib.strerror(st))
        ctx.save_for_backward(qkv, bias_t, mask_t, mask_id)
        ctx.heads, ctx.scale = heads, float(scale)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_out):
        from . import _lib
        qkv, bias_t, mask_t, mask_id = ctx.saved_tensors
        Bw, nW, N = qkv.shape[0], qkv.shape[1], qkv.shape[2]
        d_out = d_out.contiguous()
        d_qkv = torch.empty_like(qkv)
        st = _lib.lib().window_attention_backward_bf16(
            qkv.data_ptr(), d_out.data_ptr(), bias_t.data_ptr(), None if mask_t is None else mask_t.data_ptr(),
            None if mask_id is None else mask_id.data_ptr(), Bw * nW, nW, ctx.heads, N, ctx.scale, d_qkv.data_ptr(),
            _norm._stream(qkv))
        if st:
            raise RuntimeError("window_attention_backward: " + _lib.strerror(st))
        return d_qkv, None, None, None, None, None


def window_row_map(H, W, ws, shift, device):
    """(rowmap [nW * ws * ws] int32, pads): the image-order row of every token of every window as the reference's pad -> roll ->
    window_partition lays them out (models/swin/swin_transformer.py:362-379); padding positions are -(slot + 1)."""
    Hp, Wp = -(-H // ws) * ws, -(-W // ws) * ws
    real = torch.zeros(Hp, Wp, dtype=torch.bool)
    real[:H, :W] = True
    idx = torch.zeros(Hp, Wp, dtype=torch.int64)
    idx[real] = torch.arange(H * W)
    pads = int((~real).sum())
    idx[~real] = -(torch.arange(pads) + 1)
    if shift:
        idx = torch.roll(idx, shifts=(-shift, -shift), dims=(0, 1))
    m = idx.view(Hp // ws, ws, Wp // ws, ws).permute(0, 2, 1, 3).reshape(-1).to(torch.int32).contiguous()
    return m.to(device), pads


class WindowAttentionRowsFunction(torch.autograd.Function):
    """WindowAttentionFunction with pad / cyclic shift / window partition / reverse / crop folded into the kernels' addressing:
    qkv [B, H W, 3 C] and the result [B, H W, C] stay in image order, `rowmap` names every window token's row, padding tokens read
    `pad_row` (the qkv bias: the projection of the zero padding) and their gradient rows are summed into its gradient."""

    @staticmethod
    def forward(ctx, qkv, pad_row, rowmap, pads, bias_t, mask_t, mask_id, heads, scale, tokens):
        from . import _lib
        B, T = qkv.shape[0], qkv.shape[1]
        nW = rowmap.numel() // tokens
        qkv = qkv.contiguous()
        pr = None if pad_row is None else pad_row.contiguous()
        out = torch.empty(B, T, heads * 32, dtype=qkv.dtype, device=qkv.device)
        st = _lib.lib().window_attention_rows_forward_bf16(
            qkv.data_ptr(), None if pr is None else pr.data_ptr(), rowmap.data_ptr(), T, pads, bias_t.data_ptr(),
            None if mask_t is None else mask_t.data_ptr(), None if mask_id is None else mask_id.data_ptr(), B * nW, nW, heads,
            tokens, float(scale), out.data_ptr(), _norm._stream(qkv))
        if st:
            raise RuntimeError("window_attention_rows_forward: " + _lib.strerror(st))
        ctx.save_for_backward(qkv, pr, rowmap, bias_t, mask_t, mask_id)
        ctx.meta = (pads, heads, float(scale), tokens)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_out):
        from . import _lib
        qkv, pr, rowmap, bias_t, mask_t, mask_id = ctx.saved_tensors
        pads, heads, scale, tokens = ctx.meta
        B, T = qkv.shape[0], qkv.shape[1]
        nW = rowmap.numel() // tokens
        d_out = d_out.contiguous()
        d_qkv = torch.empty_like(qkv)
        d_pad = torch.empty(B * pads, qkv.shape[2], dtype=qkv.dtype, device=qkv.device) if (pads and pr is not None) else None
        st = _lib.lib().window_attention_rows_backward_bf16(
            qkv.data_ptr(), None if pr is None else pr.data_ptr(), rowmap.data_ptr(), T, pads, d_out.data_ptr(), bias_t.data_ptr(),
            None if mask_t is None else mask_t.data_ptr(), None if mask_id is None else mask_id.data_ptr(), B * nW, nW, heads, tokens,
            scale, d_qkv.data_ptr(), None if d_pad is None else d_pad.data_ptr(), _norm._stream(qkv))
        if st:
            raise RuntimeError("window_attention_rows_backward: " + _lib.strerror(st))
        g_pad = None
        if pr is not None and ctx.needs_input_grad[1]:
            g_pad = d_pad.float().sum(0).to(pr.dtype) if d_pad is not None else torch.zeros_like(pr)
        return d_qkv, g_pad, None, None, None, None, None, None, None, None


class WindowAttention(nn.Module):
    def __init__(self, dim, window_size, num_heads, qkv_bias=True, attn_drop=0.0, proj_drop=0.0):
        super().__init__()
        self.dim, self.ws, self.num_heads = dim, window_size, num_heads
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * window_size - 1) ** 2, num_heads))
        self.register_buffer("relative_position_index", relative_position_index(window_size))
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = attn_drop
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)

    def bias(self, dtype):
        """[heads, N, N] relative-position bias.  Cached ONLY for a frozen table (requires_grad False): a trainable
        table is rewritten by the fused optimiser through raw pointers and by HIP-graph replays, neither of which
        bumps `_version`, so a version-keyed cache would serve a stale bias to a later eval pass."""
        t = self.relative_position_bias_table
        n = self.ws * self.ws
        if t.requires_grad:
            return t[self.relative_position_index.view(-1)].view(n, n, -1).permute(2, 0, 1).to(dtype)
        key = (t._version, t.data_ptr(), dtype)
        if getattr(self, "_bias_key", None) != key:
            with torch.no_grad():
                self._bias = t[self.relative_position_index.view(-1)].view(n, n, -1).permute(2, 0, 1).contiguous().to(dtype)
            self._bias_key = key
        return self._bias

    def fused_supported(self, x):
        t = self.relative_position_bias_table
        return (fused_window_attention and _norm._on_device(x) and x.dtype == torch.bfloat16 and not t.requires_grad
                and self.dim // self.num_heads == 32 and self.ws * self.ws <= 64 and not torch.is_autocast_enabled()
                and not (self.training and self.attn_drop > 0) and self.qkv.weight.dtype == torch.bfloat16)

    def bias_table_t(self):
        """the frozen relative-position bias as the kernel reads it: float32 [heads, 64 queries, 64 keys], -30000 in the padded
        key columns (include/rlipv2_swin.h); cached like `bias` (frozen tables only)"""
        t = self.relative_position_bias_table
        key = (t._version, t.data_ptr(), "t")
        if getattr(self, "_bias_t_key", None) != key:
            with torch.no_grad():
                n = self.ws * self.ws
                b = t[self.relative_position_index.view(-1)].view(n, n, -1).permute(2, 0, 1)
                self._bias_t = _padded(b, n, -30000.0)
            self._bias_t_key = key
        return self._bias_t

    def forward_image_order(self, y, mask, rows):
        """y [B, H, W, C] (normalised, image order) -> [B, H, W, C]: the fused kernel with the row map `rows` = (rowmap, pads) of
        this block's shift (window_row_map): no pad / roll / partition / reverse copies.  Only when `fused_supported`."""
        B, H, W, C = y.shape
        h = self.num_heads
        packed = token_linear(y.reshape(B, H * W, C), self.qkv.weight, self.qkv.bias)
        mask_t, mask_id = (None, None) if mask is None else mask.compact
        out = WindowAttentionRowsFunction.apply(packed, self.qkv.bias, rows[0], rows[1], self.bias_table_t(), mask_t, mask_id, h,
                                                (C // h) ** -0.5, self.ws * self.ws)
        return self.proj_drop(token_linear(out, self.proj.weight, self.proj.bias)).view(B, H, W, C)

    def forward(self, x, mask):
        """x [B, nW, N, C]; mask None or [nW, N, N] -> [B, nW, N, C]"""
        B, nW, N, C = x.shape
        h = self.num_heads
        packed = token_linear(x, self.qkv.weight, self.qkv.bias)
        if self.fused_supported(x):
            # one kernel per direction on the packed projection (csrc/window_attention.hip); `mask` arrives with its compact
            # table attached by BasicLayer (mask.compact = (mask_t, mask_id))
            # (compact == (None, None): every window's mask is zero; a mask without the attribute keeps the op sequence)
            compact = (None, None) if mask is None else getattr(mask, "compact", None)
            if compact is not None:
                mask_t, mask_id = compact
                out = WindowAttentionFunction.apply(packed.view(B, nW, N, 3, h, C // h), self.bias_table_t(), mask_t, mask_id, h,
                                                    (C // h) ** -0.5)
                return self.proj_drop(token_linear(out, self.proj.weight, self.proj.bias))
        qkv = packed.view(B, nW, N, 3, h, C // h).permute(3, 0, 1, 4, 2, 5)
        add = self.bias(x.dtype)[None, None]                      # [1, 1, h, N, N]
        if mask is not None:
            add = add + mask.to(x.dtype)[None, :, None]           # [1, nW, h, N, N]
        if x.dtype == torch.float32:
            out = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2], attn_mask=add,
                                                 dropout_p=self.attn_drop if self.training else 0.0)
        else:
            # bf16: with this additive mask scaled_dot_product_attention takes its "math" route, which up-casts
            # q, k, v to float32 (measured: ~14 ms of float32 GEMMs per Swin-L step); the same three steps in
            # bf16 with a float32 softmax:
            attn = torch.matmul(qkv[0] * (C // h) ** -0.5, qkv[1].transpose(-2, -1)) + add
            attn = torch.softmax(attn, dim=-1, dtype=torch.float32).to(x.dtype)
            if self.training and self.attn_drop > 0:
                attn = F.dropout(attn, self.attn_drop)
            out = torch.matmul(attn, qkv[2])
        out = out.permute(0, 1, 3, 2, 4).reshape(B, nW, N, C)
        return self.proj_drop(token_linear(out, self.proj.weight, self.proj.bias))


class SwinTransformerBlock(nn.Module):
    def __init__(self, dim, num_heads, window_size=7, shift_size=0, mlp_ratio=4.0, qkv_bias=True, drop=0.0,
                 attn_drop=0.0, drop_path=0.0):
        super().__init__()
        assert 0 <= shift_size < window_size, "shift_size must in 0-window_size"
        self.ws, self.shift = window_size, shift_size
        self.norm1 = nn.LayerNorm(dim)
        self.attn = WindowAttention(dim, window_size, num_heads, qkv_bias, attn_drop, drop)
        self.drop_path = DropPath(drop_path)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio), drop)

    def attention(self, y, mask, rows=None):
        """the attention branch on the NORMALISED input y [B, H, W, C] (pad, shift, window partition, attention, reverse),
        drop-path applied: what is added to the residual stream.  `rows`: the stage's row maps ((unshifted), (shifted)) for the
        fused kernel, which then does all of that by addressing"""
        B, H, W, C = y.shape
        ws = self.ws
        m = mask if self.shift else None
        if rows is not None and self.attn.fused_supported(y) and (m is None or getattr(m, "compact", None) is not None):
            return self.drop_path(self.attn.forward_image_order(y, m, rows[1 if self.shift else 0]))
        pad_r, pad_b = (ws - W % ws) % ws, (ws - H % ws) % ws
        if pad_r or pad_b:
            y = F.pad(y, (0, 0, 0, pad_r, 0, pad_b))
        Hp, Wp = H + pad_b, W + pad_r
        if self.shift:
            y = torch.roll(y, shifts=(-self.shift, -self.shift), dims=(1, 2))
        y = y.view(B, Hp // ws, ws, Wp // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, -1, ws * ws, C)
        y = self.attn(y, mask if self.shift else None)
        y = y.view(B, Hp // ws, Wp // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, C)
        if self.shift:
            y = torch.roll(y, shifts=(self.shift, self.shift), dims=(1, 2))
        if pad_r or pad_b:
            y = y[:, :H, :W]
        return self.drop_path(y)

    def branches(self, x, pending, mask, rows=None):
        """One block on the residual stream `x + pending` (`pending`: the previous block's MLP branch, not yet added; None for
        the first block).  Returns (x, pending'): the stream after this block's attention branch and this block's MLP branch,
        again not yet added -- so that every residual add runs together with the LayerNorm that reads its result
        (norm.residual_pre_norm: one pass on the GPU at the Swin widths, `add` + `layer_norm` otherwise; same values)."""
        x, y = residual_pre_norm(x, pending, self.norm1)
        x, y = residual_pre_norm(x, self.attention(y, mask, rows), self.norm2)
        return x, self.drop_path(self.mlp(y))

    def forward(self, x, mask):
        """x [B, H, W, C] (reference SwinTransformerBlock.forward, models/swin/swin_transformer.py:355-403)"""
        x, pending = self.branches(x, None, mask)
        return x + pending


class PatchMerging(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = nn.LayerNorm(4 * dim)

    def forward(self, x):
        """[B, H, W, C] -> [B, ceil(H/2), ceil(W/2), 2C]"""
        B, H, W, C = x.shape
        if H % 2 or W % 2:
            x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
        x = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1)
        _, y = residual_pre_norm(x, None, self.norm)                # (the fused LayerNorm at 768 / 1 536 channels when its route is on)
        return token_linear(y, self.reduction.weight, None)


class BasicLayer(nn.Module):
    def __init__(self, dim, depth, num_heads, window_size=7, mlp_ratio=4.0, qkv_bias=True, drop=0.0, attn_drop=0.0,
                 drop_path=0.0, downsample=True):
        super().__init__()
        self.ws, self.shift = window_size, window_size // 2
        dp = drop_path if isinstance(drop_path, (list, tuple)) else [drop_path] * depth
        self.blocks = nn.ModuleList([
            SwinTransformerBlock(dim, num_heads, window_size, 0 if i % 2 == 0 else window_size // 2, mlp_ratio, qkv_bias,
                                 drop, attn_drop, dp[i]) for i in range(depth)])
        self.downsample = PatchMerging(dim) if downsample else None
        self._masks = {}
        self._rows = {}

    def forward(self, x, out_norm=None):
        """-> (stage output, input of the next stage, out_norm(stage output) or None)"""
        B, H, W, C = x.shape
        ws = self.ws
        Hp, Wp = -(-H // ws) * ws, -(-W // ws) * ws
        key = (Hp, Wp, str(x.device))
        if key not in self._masks:
            m = shift_mask(Hp, Wp, ws, self.shift, x.device)
            m.compact = compact_masks(m) if ws * ws <= 64 else (None, None)      # (for the fused attention kernel)
            self._masks[key] = m
        mask = self._masks[key]
        rows = None
        if fused_window_attention and ws * ws <= 64:
            rkey = (H, W, str(x.device))
            if rkey not in self._rows:
                self._rows[rkey] = (window_row_map(H, W, ws, 0, x.device), window_row_map(H, W, ws, self.shift, x.device))
            rows = self._rows[rkey]
        pending = None
        for blk in self.blocks:
            x, pending = blk.branches(x, pending, mask, rows)
        # the last block's MLP branch is added together with the stage's output norm when there is one
        if out_norm is not None:
            x, normed = residual_pre_norm(x, pending, out_norm)
        else:
            x, normed = (x if pending is None else x + pending), None
        return x, (self.downsample(x) if self.downsample is not None else x), normed


class PatchEmbed(nn.Module):
    def __init__(self, patch_size=4, in_chans=3, embed_dim=96, patch_norm=True):
        super().__init__()
        self.patch_size = patch_size
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.norm = nn.LayerNorm(embed_dim) if patch_norm else None

    def forward(self, x):
        """[B, 3, H, W] -> [B, H/4, W/4, C]"""
        p = self.patch_size
        H, W = x.shape[-2:]
        if W % p or H % p:
            x = F.pad(x, (0, (p - W % p) % p, 0, (p - H % p) % p))
        x = self.proj(x).permute(0, 2, 3, 1)
        return self.norm(x) if self.norm is not None else x


class SwinTransformer(nn.Module):
    def __init__(self, pretrain_img_size=224, patch_size=4, in_chans=3, embed_dim=96, depths=(2, 2, 6, 2),
                 num_heads=(3, 6, 12, 24), window_size=7, mlp_ratio=4.0, qkv_bias=True, drop_rate=0.0,
                 attn_drop_rate=0.0, drop_path_rate=0.2, ape=False, patch_norm=True, out_indices=(0, 1, 2, 3)):
        super().__init__()
        self.num_layers, self.embed_dim, self.ape, self.out_indices = len(depths), embed_dim, ape, tuple(out_indices)
        self.patch_embed = PatchEmbed(patch_size, in_chans, embed_dim, patch_norm)
        if ape:
            r = pretrain_img_size // patch_size
            self.absolute_pos_embed = nn.Parameter(torch.zeros(1, embed_dim, r, r))
            nn.init.trunc_normal_(self.absolute_pos_embed, std=0.02)
        self.pos_drop = nn.Dropout(drop_rate)
        dpr = [float(v) for v in torch.linspace(0, drop_path_rate, sum(depths))]
        self.layers = nn.ModuleList([
            BasicLayer(embed_dim * 2 ** i, depths[i], num_heads[i], window_size, mlp_ratio, qkv_bias, drop_rate,
                       attn_drop_rate, dpr[sum(depths[:i]):sum(depths[:i + 1])], downsample=i < len(depths) - 1)
            for i in range(len(depths))])
        self.num_features = [embed_dim * 2 ** i for i in range(len(depths))]
        for i in self.out_indices:
            self.add_module(f"norm{i}", nn.LayerNorm(self.num_features[i]))
        self.apply(self._init_weights)

    @staticmethod
    def _init_weights(m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def forward(self, x):
        """[B, 3, H, W] -> {"layer{i}": [B, C_i, H_i, W_i]} (channels-last memory) for i in out_indices"""
        x = self.patch_embed(x)
        if self.ape:
            pos = F.interpolate(self.absolute_pos_embed, size=x.shape[1:3], mode="bicubic")
            x = x + pos.permute(0, 2, 3, 1)
        x = self.pos_drop(x)
        outs = {}
        for i, layer in enumerate(self.layers):
            _, x, normed = layer(x, getattr(self, f"norm{i}") if i in self.out_indices else None)
            if i in self.out_indices:
                outs[f"layer{i}"] = normed.permute(0, 3, 1, 2)
        return outs


SWIN_PRESETS = {
    "swin_tiny": dict(depths=(2, 2, 6, 2), embed_dim=96, num_heads=(3, 6, 12, 24), channels=(192, 384, 768)),
    "swin_small": dict(depths=(2, 2, 18, 2), embed_dim=96, num_heads=(3, 6, 12, 24), channels=(192, 384, 768)),
    "swin_base": dict(depths=(2, 2, 18, 2), embed_dim=128, num_heads=(4, 8, 16, 32), channels=(256, 512, 1024)),
    "swin_large": dict(depths=(2, 2, 18, 2), embed_dim=192, num_heads=(6, 12, 24, 48), channels=(384, 768, 1536)),
}


class SwinBackbone(nn.Module):
    """The reference's `Backbone` for swin names (models/swin/backbone.py:100-169): stages 1-3 at strides
    8/16/32; absolute / relative position tables and all norm layers frozen (BackboneBase :66-69)."""

    def __init__(self, name="swin_large", num_feature_levels=3, drop_path_rate=0.2):
        super().__init__()
        key = next(k for k in ("swin_large", "swin_base", "swin_small", "swin_tiny") if k.split("_")[1] in name)
        cfg = SWIN_PRESETS[key]
        big_window = "384" in name
        self.body = SwinTransformer(pretrain_img_size=384 if big_window else 224, depths=cfg["depths"],
                                    embed_dim=cfg["embed_dim"], num_heads=cfg["num_heads"],
                                    window_size=12 if big_window else 7, drop_path_rate=drop_path_rate,
                                    out_indices=(1, 2, 3)[-num_feature_levels:])
        for n, p in self.body.named_parameters():
            if "absolute_pos_embed" in n or "relative_position_bias_table" in n or "norm" in n:
                p.requires_grad_(False)
        self.strides = [8, 16, 32][-num_feature_levels:]
        self.num_channels = list(cfg["channels"])[-num_feature_levels:]

    def forward(self, tensor_list: NestedTensor):
        m = tensor_list.mask
        assert m is not None
        out = []
        for _, x in sorted(self.body(tensor_list.tensors).items()):
            mask = F.interpolate(m[None].float(), size=x.shape[-2:]).to(torch.bool)[0]
            out.append(NestedTensor(x, mask, getattr(tensor_list, "no_padding", False)))
        return out


def build_swin_backbone(name="swin_large", hidden_dim=256, num_feature_levels=3, drop_path_rate=0.2):
    from .backbone import Joiner
    return Joiner(SwinBackbone(name, num_feature_levels, drop_path_rate),
                  PositionEmbeddingSine(hidden_dim // 2, normalize=True))
