"""The ALIF-fused deformable encoder of RLIPv2.

Reference: DeformableTransformerEncoderLayer (models/dab_deformable/deformable_transformer.py:1261-1300,
twin models/deformable_transformer.py:719-758) and RLIPv2_DeformableTransformerEncoder
(models/deformable_transformer.py:791-884).  Parameter names match the reference.
"""
from __future__ import annotations

import copy

import torch
import torch.nn.functional as F
from torch import nn

from .deform_attn import MSDeformAttn
from .linear import _AddInto, _Alias, attention_block_link_ok, ffn_residual_norm, fused_ffn, token_linear
from .norm import AddLayerNormFunction, GradLink, add_layer_norm


inplace_tail = True       # (tests flip it to compare with the concatenating form)
cache_reference_points = True


def _clones(module, n):
    return nn.ModuleList([copy.deepcopy(module) for _ in range(n)])


def _activation(name):
    if name == "relu":
        return F.relu
    if name == "gelu":
        return F.gelu
    if name == "glu":
        return F.glu
    raise RuntimeError(f"activation should be relu/gelu, not {name}.")


class DeformableTransformerEncoderLayer(nn.Module):
    """Post-norm block: MSDeformAttn(src + pos) -> +res -> LN -> FFN -> +res -> LN."""

    def __init__(self, d_model=256, d_ffn=1024, dropout=0.1, activation="relu", n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        self.self_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.linear1 = nn.Linear(d_model, d_ffn)
        self.activation = _activation(activation)
        self.dropout2 = nn.Dropout(dropout)
        self.linear2 = nn.Linear(d_ffn, d_model)
        self.dropout3 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(d_model)

    def forward(self, src, pos, reference_points, spatial_shapes, level_start_index, padding_mask=None):
        if (not (self.training and self.dropout1.p > 0)
                and attention_block_link_ok(src, pos, self.self_attn.value_proj.weight, self.norm1)):
            # Linked block: d_src = d_residual + d_value W_v + d_query has three producers (the LayerNorm's backward, the value
            # projection's input-gradient GEMM, the query's add); left to autograd that is two sums of 45 MB tensors.  Here the
            # LayerNorm's tensor is the accumulator: the GEMM adds into it (beta = 1), the query's gradient is added in place.
            # `xa` is consumed by exactly these three nodes, so its gradient buffer holds that tensor by reference.
            link = GradLink()
            xa = _Alias.apply(src, link)
            q = _AddInto.apply(xa, pos, link)
            attn = self.self_attn(q, reference_points, xa, spatial_shapes, level_start_index, padding_mask,
                                  value_grad_link=link)
            src = AddLayerNormFunction.apply(xa, attn, self.norm1.weight, self.norm1.bias, self.norm1.eps, link)
        else:
            q = src if pos is None else src + pos
            attn = self.self_attn(q, reference_points, src, spatial_shapes, level_start_index, padding_mask)
            src = add_layer_norm(src, self.dropout1(attn), self.norm1)
        if self.activation is F.relu and not (self.training and self.dropout2.p > 0):
            # one autograd node for linear2(relu(linear1(.))): the ReLU mask rides in the input-gradient GEMM, and the
            # residual's gradient in the last GEMM's accumulator (dropout3 is the identity here: --dropout 0.0; with a live
            # dropout between branch and norm the plain nodes run)
            if not (self.training and self.dropout3.p > 0):
                return ffn_residual_norm(src, self.linear1, self.linear2, self.norm2)
            return add_layer_norm(src, self.dropout3(fused_ffn(src, self.linear1, self.linear2)), self.norm2)
        if self.activation is F.relu:
            hidden = self.dropout2(token_linear(src, self.linear1.weight, self.linear1.bias, relu=True))
        else:
            hidden = self.dropout2(self.activation(token_linear(src, self.linear1.weight, self.linear1.bias)))
        ffn = token_linear(hidden, self.linear2.weight, self.linear2.bias)
        return add_layer_norm(src, self.dropout3(ffn), self.norm2)


def encoder_reference_points(spatial_shapes_list, valid_ratios, device):
    """Pixel-centre reference points of every pyramid cell, per level scaled by the valid ratios:
    [N, S, L, 2] (reference: models/deformable_transformer.py:803-815).  `spatial_shapes_list` is
    the host copy [(H, W), ...] so no device->host sync is needed."""
    refs = []
    for lvl, (H, W) in enumerate(spatial_shapes_list):
        ys = torch.linspace(0.5, H - 0.5, H, dtype=torch.float32, device=device)
        xs = torch.linspace(0.5, W - 0.5, W, dtype=torch.float32, device=device)
        ref_y, ref_x = torch.meshgrid(ys, xs, indexing="ij")
        ref_y = ref_y.reshape(-1)[None] / (valid_ratios[:, None, lvl, 1] * H)
        ref_x = ref_x.reshape(-1)[None] / (valid_ratios[:, None, lvl, 0] * W)
        refs.append(torch.stack((ref_x, ref_y), -1))
    ref = torch.cat(refs, 1)
    return ref[:, :, None] * valid_ratios[:, None]


class _TailGradient:
    """what _TakeTail and _PutTail share in the backward pass: the full-size gradient buffer"""
    full = None


class _TakeTail(torch.autograd.Function):
    """tail = x[:, start:].clone() -- the last pyramid level, handed to the fusion (reference
    deformable_transformer.py:844-846 clones it too).  Its gradient is NOT returned as a zero-padded full-size tensor for
    autograd to add to _PutTail's: it is copied into the tail of the gradient buffer _PutTail has already handed over for
    the same `x` (see there)."""

    @staticmethod
    def forward(ctx, x, start, shared):
        ctx.start, ctx.shared = start, shared
        return x[:, start:].clone()

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_tail):
        full = ctx.shared.full
        ctx.shared.full = None
        if full is None:                 # _PutTail's output took no gradient: the plain zero-padded form
            out = g_tail.new_zeros((g_tail.shape[0], ctx.start + g_tail.shape[1]) + tuple(g_tail.shape[2:]))
            out[:, ctx.start:] = g_tail
            return out, None, None
        full[:, ctx.start:].copy_(g_tail)
        return None, None, None


class _PutTail(torch.autograd.Function):
    """x[:, start:] = fused tail, in place (Q4: the reference writes the fused last-level slice back into the full tensor,
    deformable_transformer.py:855-859).  `torch.cat([x[:, :start], tail], 1)` is the same values through a 90 MB copy, and its
    backward through autograd is two zero-filled full-size gradients (one per slice) plus their sum: ~310 MB of traffic per
    fusion for a 0.5 MB slice.  Here the forward copies the slice; the backward hands the incoming full-size gradient on AS
    the gradient of `x` and lets _TakeTail overwrite its tail with the fusion's input gradient: `x` is consumed by exactly
    these two nodes (the encoder below), so nothing reads the buffer between the two."""

    @staticmethod
    def forward(ctx, x, tail, start, shared):
        ctx.start, ctx.shared = start, shared
        x[:, start:] = tail
        ctx.mark_dirty(x)
        return x

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        g_tail = g[:, ctx.start:].clone()
        # (the buffer must be this node's own: a gradient that is an expanded / shared view is materialised first)
        if not g.is_contiguous() or g._base is not None:
            g = g.contiguous().clone() if g._base is not None else g.contiguous()
        ctx.shared.full = g
        return g, g_tail, None, None


class RLIPv2_DeformableTransformerEncoder(nn.Module):
    """Every `fusion_interval`-th layer is preceded by an ALIF fusion (on the last pyramid level
    only when `fusion_last_vis`) and a language layer on the fused text states."""

    def __init__(self, encoder_layer, roberta_layer, VLFuse_layer, num_layers, fusion_interval=2,
                 fusion_last_vis=False, lang_aux_loss=False):
        super().__init__()
        self.layers = _clones(encoder_layer, num_layers)
        self.num_layers = num_layers
        self.fusion_interval = fusion_interval
        self.roberta_layers = _clones(roberta_layer, num_layers // fusion_interval)
        self.VLFuse_layers = _clones(VLFuse_layer, num_layers // fusion_interval)
        self.fusion_last_vis = fusion_last_vis
        self.lang_aux_loss = lang_aux_loss

    def forward(self, src, spatial_shapes, level_start_index, valid_ratios, pos=None, padding_mask=None,
                lang_hidden=None, lang_masks=None, spatial_shapes_list=None, skip_value_mask=False):
        """`skip_value_mask`: the caller knows the batch has NO padding (all masks False, hence all valid ratios 1): the value
        masking is skipped and the reference points come from the shape alone (cached)."""
        if spatial_shapes_list is None:                      # reference behaviour: read them back
            spatial_shapes_list = [(int(h), int(w)) for h, w in spatial_shapes.tolist()]
        last_start = sum(h * w for h, w in spatial_shapes_list[:-1])
        if skip_value_mask and cache_reference_points:
            # (no padding: the valid ratios are all 1 and the reference points a function of the pyramid's shape alone --
            #  ~45 launches per step; same function, same values)
            rkey = (tuple(spatial_shapes_list), valid_ratios.shape[0], str(src.device))
            rcache = self.__dict__.setdefault("_reference_cache", {})
            if rkey not in rcache:
                capturing = src.is_cuda and torch.cuda.is_current_stream_capturing()
                with torch.no_grad():
                    points = encoder_reference_points(spatial_shapes_list, torch.ones_like(valid_ratios), src.device)
                # (never evicted: captured graphs keep reading a cached tensor's memory; a full table stops caching, and a
                #  capture's own pool memory is not kept)
                if capturing or len(rcache) >= 16:
                    rcache = {rkey: points}
                else:
                    rcache[rkey] = points
            reference_points = rcache[rkey]
        else:
            reference_points = encoder_reference_points(spatial_shapes_list, valid_ratios, src.device)
        # the fusion is handed inverted (True = valid) bool masks; see alif.py Q1 for what they do
        vis_mask = ~padding_mask
        lang_mask = ~lang_masks
        if self.fusion_last_vis:
            vis_mask, vis_pos = vis_mask[:, last_start:], pos[:, last_start:]
        else:
            vis_pos = pos
        output, hidden = src, lang_hidden
        collected = []
        for idx, layer in enumerate(self.layers):
            if idx % self.fusion_interval == 0:
                k = idx // self.fusion_interval
                # (in place only into a tensor this loop produced -- the previous layer's LayerNorm output, which no
                #  backward has saved; layer 0's input belongs to the caller and keeps the out-of-place form)
                in_place = self.fusion_last_vis and idx > 0 and inplace_tail and torch.is_grad_enabled() and output.requires_grad
                if in_place:
                    shared = _TailGradient()
                    part = _TakeTail.apply(output, last_start, shared)
                else:
                    part = output[:, last_start:] if self.fusion_last_vis else output
                fused = self.VLFuse_layers[k]({"visual": {"src": part, "padding_mask": vis_mask, "pos": vis_pos},
                                               "lang": {"hidden": hidden, "masks": lang_mask}})
                part, hidden = fused["visual"]["src"], fused["lang"]["hidden"]
                # Q4: the fused last-level slice replaces that slice of the full sequence
                if in_place:
                    output = _PutTail.apply(output, part, last_start, shared)
                else:
                    output = torch.cat([output[:, :last_start], part], 1) if self.fusion_last_vis else part
                hidden = self.roberta_layers[k](hidden_states=hidden, attention_mask=lang_mask)
                collected.append(hidden)
            output = layer(output, pos, reference_points, spatial_shapes, level_start_index,
                           None if skip_value_mask else padding_mask)
        if self.lang_aux_loss:
            lang = torch.stack(collected if self.fusion_interval == 2 else collected[::2], dim=0)
        else:
            lang = collected[-1]
        return output, lang
