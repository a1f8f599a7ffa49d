"""DAB deformable decoders of RLIPv2-ParSeDA (human-object pair decoder and verb decoder).

Reference: DeformableTransformerDecoderLayer (models/dab_deformable/deformable_transformer.py:1346-1401)
and DABDeformableTransformerDecoderHOI (:1404-1552).  Parameter names match the reference.
"""
from __future__ import annotations


import torch
import torch.nn.functional as F
from torch import nn

from . import _lib
from .blocks import MLP, _sine_dim_t, inverse_sigmoid, sine_embed_for_position
from .deform_attn import MSDeformAttn
from .encoder import _activation, _clones
from .linear import token_linear
from .norm import add_layer_norm


class _SplitRows(torch.autograd.Function):
    """(w[:n], w[n:]) as views; the backward is ONE cat of the two gradients (plain slicing costs a zero-fill, a copy
    and an accumulation per slice)."""

    @staticmethod
    def forward(ctx, w, n):
        ctx.n, ctx.shape = n, w.shape
        return w[:n], w[n:]

    @staticmethod
    def backward(ctx, ga, gb):
        if ga is None or gb is None:
            g = (ga if ga is not None else gb).new_zeros(ctx.shape)
            (g[:ctx.n] if ga is not None else g[ctx.n:]).copy_(ga if ga is not None else gb)
            return g, None
        return torch.cat((ga, gb), 0), None


direct_self_attention = True
fused_glue = True


share_box_deltas = True       # the heads' MLPs run once per layer and are shared with the refinement (device-agnostic)
# GPU-only route, OFF until rlipv2_amd/routes.validate() has compared it with the op sequence on the caller's own step
one_launch_box_head = False


def _glue_ok(*tensors):
    return fused_glue and all(t.is_cuda and t.dtype == torch.float32 for t in tensors)


def refine_boxes(delta, ref, eps=1e-5):
    """sigmoid(delta + inverse_sigmoid(ref)) without autograd, float32 (reference deformable_transformer.py:1519-1541):
    one launch of csrc/decoder_glue.hip on the GPU."""
    with torch.no_grad():
        if not (_glue_ok(ref) and delta.is_cuda and delta.dtype in (torch.bfloat16, torch.float32)
                and delta.shape == ref.shape and ref.shape[-1] == 4):
            return (delta.float() + inverse_sigmoid(ref, eps)).sigmoid()
        delta, ref = delta.contiguous(), ref.contiguous()
        out = torch.empty_like(ref)
        st = _lib.lib().dab_refine_boxes(delta.data_ptr(), int(delta.dtype == torch.bfloat16), ref.data_ptr(),
                                         out.data_ptr(), ref.numel() // 4, float(eps),
                                         torch.cuda.current_stream(ref.device).cuda_stream)
        if st:
            raise RuntimeError("dab_refine_boxes: " + _lib.strerror(st))
        return out


class BoxHeadFunction(torch.autograd.Function):
    """boxes = sigmoid(delta + inverse_sigmoid(ref)) of a prediction head (reference hoi.py:2122-2138) for a reference that
    carries no gradient (the refined, detached anchors of decoder layers >= 1): forward = the one launch of `refine_boxes`
    (bit-identical to the op sequence: inverse_sigmoid's 6 launches + cast + add + sigmoid), backward = sigmoid's own backward
    + the cast back to the head's dtype."""

    @staticmethod
    def forward(ctx, delta, ref):
        y = refine_boxes(delta, ref)
        ctx.save_for_backward(y)
        ctx.delta_dtype = delta.dtype
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        return torch.ops.aten.sigmoid_backward(g, y).to(ctx.delta_dtype), None


def box_head(delta, ref):
    """sigmoid(delta + inverse_sigmoid(ref)), differentiable in `delta` (and in `ref` when it requires grad: the
    learnable anchors of layer 0 keep the op sequence)."""
    if (one_launch_box_head and not ref.requires_grad and ref.shape[-1] == 4 and delta.shape == ref.shape and delta.is_cuda
            and _glue_ok(ref)
            and delta.dtype in (torch.bfloat16, torch.float32)):
        return BoxHeadFunction.apply(delta, ref)
    inv = inverse_sigmoid(ref)
    delta = delta.to(inv.dtype)              # no mixed-dtype elementwise ops (see below)
    if ref.shape[-1] == 4:
        return (delta + inv).sigmoid()
    assert ref.shape[-1] == 2
    return torch.cat([delta[..., :2] + inv, delta[..., 2:]], dim=-1).sigmoid()


def reference_embed(sub_ref, obj_ref, valid_ratios, parse, out_dtype):
    """(ref_in [N, nq, L, 4] float32, sine features of its level 0 [N, nq, 512] in `out_dtype`) from the anchor boxes:
    one launch of csrc/decoder_glue.hip (the boxes are detached, nothing here carries a gradient)."""
    N, n, _ = sub_ref.shape
    L = valid_ratios.shape[1]
    nq = 2 * n if parse else n
    sub_ref, obj_ref, valid_ratios = sub_ref.contiguous(), obj_ref.contiguous(), valid_ratios.contiguous()
    ref_in = torch.empty(N, nq, L, 4, dtype=torch.float32, device=sub_ref.device)
    embed = torch.empty(N, nq, 512, dtype=out_dtype, device=sub_ref.device)
    st = _lib.lib().dab_reference_embed(sub_ref.data_ptr(), obj_ref.data_ptr(), valid_ratios.data_ptr(),
                                        _sine_dim_t(sub_ref.device).data_ptr(), N, n, L, int(bool(parse)),
                                        ref_in.data_ptr(), embed.data_ptr(), int(out_dtype == torch.bfloat16),
                                        torch.cuda.current_stream(sub_ref.device).cuda_stream)
    if st:
        raise RuntimeError("dab_reference_embed: " + _lib.strerror(st))
    return ref_in, embed


class DeformableTransformerDecoderLayer(nn.Module):
    """self-attention over the queries (+LN) -> deformable cross-attention into the image memory
    (+LN) -> FFN (+LN), post-norm."""

    def __init__(self, d_model=256, d_ffn=1024, dropout=0.1, activation="relu", n_levels=4, n_heads=8, n_points=4,
                 do_self_attn=True):
        super().__init__()
        self.cross_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.do_self_attn = do_self_attn
        if do_self_attn:
            self.self_attn = nn.MultiheadAttention(d_model, n_heads, dropout=dropout)
            self.dropout2 = nn.Dropout(dropout)
            self.norm2 = nn.LayerNorm(d_model)
        self.linear1 = nn.Linear(d_model, d_ffn)
        self.activation = _activation(activation)
        self.dropout3 = nn.Dropout(dropout)
        self.linear2 = nn.Linear(d_ffn, d_model)
        self.dropout4 = nn.Dropout(dropout)
        self.norm3 = nn.LayerNorm(d_model)

    def forward(self, tgt, query_pos, reference_points, src, src_spatial_shapes, level_start_index,
                src_padding_mask=None):
        if self.do_self_attn:
            tgt = add_layer_norm(tgt, self.dropout2(self._self_attention(tgt, query_pos)), self.norm2)
        ca = self.cross_attn(tgt if query_pos is None else tgt + query_pos, reference_points, src,
                             src_spatial_shapes, level_start_index, src_padding_mask)
        tgt = add_layer_norm(tgt, self.dropout1(ca), self.norm1)
        if self.activation is F.relu:
            hidden = token_linear(tgt, self.linear1.weight, self.linear1.bias, relu=True)
        else:
            hidden = self.activation(self.linear1(tgt))
        ffn = self.linear2(self.dropout3(hidden))
        return add_layer_norm(tgt, self.dropout4(ffn), self.norm3)


def _decoder_self_attention(self, tgt, query_pos):
    """nn.MultiheadAttention(q = k = tgt + pos, v = tgt) of the decoder layer (reference deformable_transformer.py:
    1377-1381) on the module's own parameters, batch-first throughout: one GEMM for Q and K, one for V, fused attention,
    output projection -- no seq-first round trip (the module transposes [N, nq, C] -> [nq, N, C] and back, 8 copies per
    layer and direction) and one `cat` instead of slice gradients for the packed projection."""
    mha = self.self_attn
    if not (direct_self_attention and tgt.is_cuda and mha._qkv_same_embed_dim and mha.in_proj_bias is not None
            and mha.bias_k is None and not mha.add_zero_attn and not torch.is_autocast_enabled()):
        qk = (tgt if query_pos is None else tgt + query_pos).transpose(0, 1)
        return mha(qk, qk, tgt.transpose(0, 1), need_weights=False)[0].transpose(0, 1)
    B, L, E = tgt.shape
    H = mha.num_heads
    w_qk, w_v = _SplitRows.apply(mha.in_proj_weight, 2 * E)
    b_qk, b_v = _SplitRows.apply(mha.in_proj_bias, 2 * E)
    qk = token_linear(tgt if query_pos is None else tgt + query_pos, w_qk, b_qk).view(B, L, 2, H, E // H)
    v = token_linear(tgt, w_v, b_v).view(B, L, H, E // H)
    out = F.scaled_dot_product_attention(qk[:, :, 0].transpose(1, 2), qk[:, :, 1].transpose(1, 2), v.transpose(1, 2),
                                         dropout_p=mha.dropout if self.training else 0.0)
    return token_linear(out.transpose(1, 2).reshape(B, L, E), mha.out_proj.weight, mha.out_proj.bias)


DeformableTransformerDecoderLayer._self_attention = _decoder_self_attention


class DABDeformableTransformerDecoderHOI(nn.Module):
    """ParSe=True : queries are [subjects | objects]; each half refines its own anchor boxes.
    ParSe=False: verb queries; the cross-attention reference is the mean of the (sub, obj) boxes.
    Refined boxes are detached between layers (reference :1525, :1541)."""

    def __init__(self, decoder_layer, num_layers, return_intermediate=False, use_dab=False, d_model=256,
                 high_dim_query_update=False, no_sine_embed=False, ParSe=False):
        super().__init__()
        self.layers = _clones(decoder_layer, num_layers)
        self.num_layers = num_layers
        self.return_intermediate = return_intermediate
        self.sub_bbox_embed = None          # set by the model (aliases of its box heads)
        self.obj_bbox_embed = None
        # set by a model whose prediction heads apply the SAME bbox_embed[lid] to the same layer outputs (RLIP_ParSeDA,
        # hoi.py:2122-2138 against deformable_transformer.py:1511-1541): the head MLPs then run once, with autograd, and
        # the refinement uses their detached result -- `hs.deltas[lid] = (sub, obj)` goes to the heads
        self.keep_box_deltas = False
        self.class_embed = None
        self.use_dab = use_dab
        self.d_model = d_model
        self.no_sine_embed = no_sine_embed
        if use_dab:
            self.query_scale = MLP(d_model, d_model, d_model, 2)
            self.ref_point_head = MLP(4, d_model, d_model, 3) if no_sine_embed else MLP(2 * d_model, d_model, d_model, 2)
        self.high_dim_query_update = high_dim_query_update
        if high_dim_query_update:
            self.high_dim_query_proj = MLP(d_model, d_model, d_model, 2)
        self.ParSe = ParSe

    def forward(self, tgt, reference_points, src, src_spatial_shapes, src_level_start_index, src_valid_ratios,
                query_pos=None, src_padding_mask=None):
        output = tgt
        if self.use_dab:
            assert query_pos is None
        bs = src.shape[0]
        sub_ref, obj_ref = reference_points
        if self.ParSe:
            sub_ref = sub_ref[None].repeat(bs, 1, 1)
            obj_ref = obj_ref[None].repeat(bs, 1, 1)
        assert sub_ref.shape[-1] == 4 and obj_ref.shape[-1] == 4
        n_pair = obj_ref.shape[1]
        ratios4 = torch.cat([src_valid_ratios, src_valid_ratios], -1)[:, None]          # [N,1,L,4]

        # (explicit .float() before mixing with the float32 box chain: PyTorch-ROCm's mixed-dtype elementwise
        #  kernel costs ~40 us even on a [4, 150, 4] tensor, against ~2 us for a cast + a same-dtype op)
        inter, inter_sub, inter_obj, deltas = [], [], [], []
        glue = (self.use_dab and not self.no_sine_embed and _glue_ok(sub_ref, obj_ref, src_valid_ratios)
                and output.dtype in (torch.bfloat16, torch.float32) and src_valid_ratios.shape[1] <= 8)
        for lid, layer in enumerate(self.layers):
            feat = None
            # (layer 0's anchors are learnable parameters and keep the differentiable form; the refined boxes of the
            #  later layers are detached)
            if glue and not (sub_ref.requires_grad or obj_ref.requires_grad):
                ref_in, feat = reference_embed(sub_ref, obj_ref, src_valid_ratios, self.ParSe, output.dtype)
            elif self.ParSe:
                ref_in = torch.cat((sub_ref[:, :, None] * ratios4, obj_ref[:, :, None] * ratios4), dim=1)
            else:
                ref_in = 0.5 * (sub_ref + obj_ref)[:, :, None] * ratios4
            if self.use_dab:
                if feat is None:
                    feat = ref_in if self.no_sine_embed else sine_embed_for_position(ref_in[:, :, 0, :])
                raw = self.ref_point_head(feat.to(output.dtype))
                query_pos = raw if lid == 0 else self.query_scale(output) * raw
            if self.high_dim_query_update and lid != 0:
                query_pos = query_pos + self.high_dim_query_proj(output)

            output = layer(output, query_pos, ref_in, src, src_spatial_shapes, src_level_start_index,
                           src_padding_mask)

            # (the refined boxes are detached, reference :1525 / :1541: heads and chain run without autograd)
            # (the heads see a DETACHED tensor: a module called under no_grad on an input that requires grad trips
            #  torch's ModuleTracker -- FlopCounterMode, i.e. bench.py's step-roofline probe -- with "Expected gradient
            #  function to be set"; that is what removed `step_roofline` from round 2's bench line)
            out_d = output.detach()
            share = (share_box_deltas and self.keep_box_deltas and self.ParSe and self.return_intermediate and torch.is_grad_enabled()
                     and self.sub_bbox_embed is not None and self.obj_bbox_embed is not None)
            if share:
                # (one split node per layer output, shared with the heads: two slices here and another split there would
                #  be three nodes whose gradients autograd has to zero-fill, copy and add up)
                out_h, out_o = output.split(n_pair, dim=1)
                d_sub = self.sub_bbox_embed[lid](out_h)
                d_obj = self.obj_bbox_embed[lid](out_o)
                deltas.append((d_sub, d_obj, out_h, out_o))
                sub_ref = refine_boxes(d_sub.detach(), sub_ref)
                obj_ref = refine_boxes(d_obj.detach(), obj_ref)
            else:
                if self.sub_bbox_embed is not None:
                    with torch.no_grad():
                        delta = self.sub_bbox_embed[lid](out_d[:, :n_pair] if self.ParSe else out_d)
                    sub_ref = refine_boxes(delta, sub_ref)
                if self.obj_bbox_embed is not None:
                    with torch.no_grad():
                        delta = self.obj_bbox_embed[lid](out_d[:, n_pair:] if self.ParSe else out_d)
                    obj_ref = refine_boxes(delta, obj_ref)
            if self.return_intermediate:
                inter.append(output)
                inter_sub.append(sub_ref)
                inter_obj.append(obj_ref)

        if self.return_intermediate:
            refs = torch.stack((torch.stack(inter_sub), torch.stack(inter_obj)), dim=0).transpose(0, 1)
            hs = torch.stack(inter)
            # consumers that walk the layers take the per-layer tensors themselves: hs[l] would put a select (and its
            # zero-fill + copy + accumulate backward) between every head and the layer that feeds it
            hs.layers = tuple(inter)
            if len(deltas) == len(inter):
                hs.deltas = tuple(deltas)
            return hs, refs
        return output, reference_points
