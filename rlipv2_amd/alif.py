"""ALIF: the gated language-image cross-attention of the RLIPv2 encoder, and the sparse language
layer that follows it.

Reference: models/fuse_helper.py -- RLIPv2_VLFuse (:983-1096), RLIPv2_BiAttentionBlockForCheckpoint
(:591-752), RLIPv2_BiMultiHeadAttention (:314-466); models/modeling_roberta.py -- RobertaLayer
(:340-408) with its attention / output sub-modules (:117-338).  Module and parameter names match
the reference so its checkpoints load (state_dict layout: SURVEY.md section 8b, B3).

Parity-critical behaviours reproduced on purpose:
  Q1  the masks handed to the fusion arrive as bool tensors; `mask.masked_fill(mask == 0, -9e15)`
      on a bool tensor is all-True, i.e. the "mask" adds the constant 1.0 to every logit and masks
      nothing (fuse_helper.py:410-412, :425-427).  We add the same 1.0.
  Q3  the gated residual is taken on the LayerNorm'd inputs (fuse_helper.py:685-693).
  Q5  the language-side logits subtract their row maximum before the softmax (:399-400).
  Q6  attention-probability dropout 0.1 in the fusion, hidden / attention dropout 0.1 in the
      language layer: active in train(), whatever the model-level --dropout says.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F
from torch import nn

from .blocks import MultiBranchFusion

GATINGS_SCALAR = ("VXAc", "Vtanh")

# the fused HIP kernel of the attention core (csrc/alif_attention.hip); tests switch it off to compare both routes
fused_attention = True            # (attribute: tests and `bench.py --set alif.fused_attention=0` flip it for A/B runs)


class AlifAttentionFunction(torch.autograd.Function):
    """(q, k, values_l_t, values_v_t) -> (out_v, out_l): the bi-directional attention core of
    RLIPv2_BiMultiHeadAttention (fuse_helper.py:395-462) as one HIP launch (csrc/alif_attention.hip, C ABI
    include/rlipv2_alif.h).  q [B, Tv, E] (scaled), k [B, Tl, E], values_l_t [B, E, 64], values_v_t [B, E, Tvp]
    (value projections transposed, see the header).  The backward pass is written out with PyTorch matrix products on
    the probabilities the kernel saved: 8 small batched GEMMs + 2 softmax-backward expressions."""

    @staticmethod
    def forward(ctx, q, k, vlt, vvt, H, p_drop, training):
        from . import _lib
        if not q.is_cuda:
            raise RuntimeError("Not implemented on the CPU")
        B, Tv, E = q.shape
        Tl = k.shape[1]
        L = _lib.lib()
        q, k, vlt, vvt = q.contiguous(), k.contiguous(), vlt.contiguous(), vvt.contiguous()
        out_v = torch.empty_like(q)
        out_l = torch.empty_like(k)
        probs_v = torch.empty((B, H, Tv, Tl), dtype=torch.bfloat16, device=q.device)
        probs_l = torch.empty((B, H, Tl, Tv), dtype=torch.bfloat16, device=q.device)
        drop = training and p_drop > 0
        keep_v = keep_l = None
        if drop:                                            # Q6: attention-probability dropout, both directions
            keep = torch.rand((2, B, H, Tv * Tl), device=q.device) >= p_drop       # bool = one byte per entry
            keep_v, keep_l = keep[0].view(B, H, Tv, Tl), keep[1].view(B, H, Tl, Tv)
        scale = 1.0 / (1.0 - p_drop) if drop else 1.0
        st = L.alif_attention_forward_bf16(q.data_ptr(), k.data_ptr(), vlt.data_ptr(), vvt.data_ptr(),
                                           keep_v.data_ptr() if drop else None, keep_l.data_ptr() if drop else None,
                                           scale, B, H, Tv, Tl, out_v.data_ptr(), out_l.data_ptr(), probs_v.data_ptr(),
                                           probs_l.data_ptr(), torch.cuda.current_stream(q.device).cuda_stream)
        if st:
            raise RuntimeError("alif_attention_forward: " + _lib.strerror(st))
        ctx.save_for_backward(q, k, vlt, vvt, probs_v, probs_l, keep_v, keep_l)
        ctx.H, ctx.scale = H, scale
        return out_v, out_l

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_out_v, g_out_l):
        from . import _lib
        q, k, vlt, vvt, pv, pl, keep_v, keep_l = ctx.saved_tensors
        H = ctx.H
        B, Tv, E = q.shape
        Tl, hd = k.shape[1], E // H
        heads = lambda t, T: t.reshape(B, T, H, hd).transpose(1, 2)                # [B, H, T, hd] views
        g_ov, g_ol = heads(g_out_v, Tv), heads(g_out_l, Tl)
        vl_t = vlt.view(B, H, hd, -1)[..., :Tl]                                    # [B, H, hd, Tl]
        vv_t = vvt.view(B, H, hd, -1)[..., :Tv]                                    # [B, H, hd, Tv]
        # gradients of the (dropped) probabilities, then ONE kernel for both softmax backward passes + dropout
        d_pv = torch.matmul(g_ov, vl_t)                                            # [B, H, Tv, Tl]
        d_pl = torch.matmul(g_ol, vv_t)                                            # [B, H, Tl, Tv]
        d_s = torch.empty_like(pv)
        drop = keep_v is not None
        pv_d = torch.empty_like(pv) if drop else pv
        pl_d = torch.empty_like(pl) if drop else pl
        st = _lib.lib().alif_attention_softmax_backward_bf16(
            pv.data_ptr(), pl.data_ptr(), d_pv.contiguous().data_ptr(), d_pl.contiguous().data_ptr(),
            keep_v.data_ptr() if drop else None, keep_l.data_ptr() if drop else None, ctx.scale, B, H, Tv, Tl,
            d_s.data_ptr(), pv_d.data_ptr() if drop else None, pl_d.data_ptr() if drop else None,
            torch.cuda.current_stream(q.device).cuda_stream)
        if st:
            raise RuntimeError("alif_attention_softmax_backward: " + _lib.strerror(st))
        # value projections (transposed layout): d V^T = g_out^T P_dropped, zero in the padding columns
        g_vlt = torch.matmul(g_ov.transpose(-1, -2), pv_d)
        g_vvt = torch.matmul(g_ol.transpose(-1, -2), pl_d)
        if g_vlt.shape[-1] != vlt.shape[-1]:
            g_vlt = F.pad(g_vlt, (0, vlt.shape[-1] - Tl))
        if g_vvt.shape[-1] != vvt.shape[-1]:
            g_vvt = F.pad(g_vvt, (0, vvt.shape[-1] - Tv))
        g_q = torch.matmul(d_s, heads(k, Tl)).transpose(1, 2).reshape(B, Tv, E)
        g_k = torch.matmul(d_s.transpose(-1, -2), heads(q, Tv)).transpose(1, 2).reshape(B, Tl, E)
        return g_q, g_k, g_vlt.reshape(vlt.shape), g_vvt.reshape(vvt.shape), None, None, None


class RLIPv2_BiMultiHeadAttention(nn.Module):
    """One set of logits q(v+pos) . k(l)^T feeds both directions: vision attends over language
    tokens, language attends over vision tokens."""

    def __init__(self, v_dim, l_dim, embed_dim, num_heads, dropout=0.1, args=None):
        super().__init__()
        self.embed_dim, self.num_heads = embed_dim, num_heads
        self.head_dim = embed_dim // num_heads
        self.v_dim, self.l_dim = v_dim, l_dim
        assert self.head_dim * num_heads == embed_dim, \
            f"embed_dim must be divisible by num_heads (got `embed_dim`: {embed_dim} and `num_heads`: {num_heads})."
        self.scale = self.head_dim ** (-0.5)
        self.dropout = dropout
        self.v_proj = nn.Linear(v_dim, embed_dim)
        self.l_proj = nn.Linear(l_dim, embed_dim)
        self.values_v_proj = nn.Linear(v_dim, embed_dim)
        self.values_l_proj = nn.Linear(l_dim, embed_dim)
        self.out_v_proj = nn.Linear(embed_dim, v_dim)
        self.out_l_proj = nn.Linear(embed_dim, l_dim)
        self.stable_softmax_2d = getattr(args, "stable_softmax_2d", False)
        self.clamp_min_for_underflow = getattr(args, "clamp_min_for_underflow", False)
        self.clamp_max_for_overflow = getattr(args, "clamp_max_for_overflow", False)
        self._reset_parameters()

    def _reset_parameters(self):
        for lin in (self.v_proj, self.l_proj, self.values_v_proj, self.values_l_proj, self.out_v_proj,
                    self.out_l_proj):
            nn.init.xavier_uniform_(lin.weight)
            lin.bias.data.fill_(0)

    def _heads(self, x):                      # [B, T, E] -> [B, H, T, hd]
        B, T, _ = x.shape
        return x.view(B, T, self.num_heads, self.head_dim).transpose(1, 2)

    def _clamp(self, x):
        if self.clamp_min_for_underflow:
            x = torch.clamp(x, min=-50000)
        if self.clamp_max_for_overflow:
            x = torch.clamp(x, max=50000)
        return x

    def _fused_ok(self, v, l):
        from . import _lib
        return (fused_attention and v.is_cuda and v.dtype == torch.bfloat16 and l.dtype == torch.bfloat16
                and not (self.stable_softmax_2d or self.clamp_min_for_underflow or self.clamp_max_for_overflow)
                and not torch.is_autocast_enabled()
                and bool(_lib.lib().alif_attention_supported(v.shape[0], self.num_heads, v.shape[1], l.shape[1],
                                                             self.head_dim)))

    def _forward_fused(self, v, l, v_pos):
        """The attention core on the HIP kernel.  Host side: the four projections as library GEMMs -- the two value
        projections with swapped operands, i.e. written channel-major, which is the layout the kernel's P V products
        read straight from memory -- one launch for the core, the two output projections."""
        from . import _lib
        B, Tv, _ = v.shape
        Tl = l.shape[1]
        q = self.v_proj(v if v_pos is None else v + v_pos) * self.scale
        k = self.l_proj(l)
        Tvp = _lib.lib().alif_attention_padded_tv(Tv)
        l_pad = F.pad(l, (0, 0, 0, 64 - Tl)) if Tl < 64 else l          # padded tokens project to the bias: finite,
        v_pad = F.pad(v, (0, 0, 0, Tvp - Tv)) if Tvp > Tv else v        # and their probabilities are zero
        wl, wv = self.values_l_proj, self.values_v_proj
        vlt = torch.baddbmm(wl.bias.view(1, -1, 1), wl.weight.unsqueeze(0).expand(B, -1, -1), l_pad.transpose(1, 2))
        vvt = torch.baddbmm(wv.bias.view(1, -1, 1), wv.weight.unsqueeze(0).expand(B, -1, -1), v_pad.transpose(1, 2))
        out_v, out_l = AlifAttentionFunction.apply(q, k, vlt, vvt, self.num_heads, self.dropout, self.training)
        return self.out_v_proj(out_v), self.out_l_proj(out_l)

    def forward(self, v, l, v_pos=None, attention_mask_l=None, attention_mask_v=None):
        # (Q1: bool masks only add a constant to every logit, which no softmax sees; other mask dtypes do mask)
        if ((attention_mask_l is None or attention_mask_l.dtype == torch.bool)
                and (attention_mask_v is None or attention_mask_v.dtype == torch.bool) and self._fused_ok(v, l)):
            return self._forward_fused(v, l, v_pos)
        B, Tv, _ = v.shape
        Tl = l.shape[1]
        q = self._heads(self.v_proj(v if v_pos is None else v + v_pos) * self.scale)    # [B,H,Tv,hd]
        k = self._heads(self.l_proj(l))                                                  # [B,H,Tl,hd]
        val_v = self._heads(self.values_v_proj(v))
        val_l = self._heads(self.values_l_proj(l))

        logits = torch.matmul(q, k.transpose(-1, -2))                                    # [B,H,Tv,Tl]
        if self.stable_softmax_2d:
            logits = logits - logits.max()
        logits = self._clamp(logits)

        logits_t = logits.transpose(-1, -2)                                              # [B,H,Tl,Tv]
        logits_l = self._clamp(logits_t - logits_t.max(dim=-1, keepdim=True)[0])         # Q5
        if attention_mask_v is not None:
            logits_l = logits_l + self._mask_term(attention_mask_v, logits_l)            # Q1
        probs_l = logits_l.softmax(dim=-1)

        logits_v = logits
        if attention_mask_l is not None:
            logits_v = logits_v + self._mask_term(attention_mask_l, logits_v)            # Q1
        probs_v = logits_v.softmax(dim=-1)

        probs_v = F.dropout(probs_v, p=self.dropout, training=self.training)             # Q6
        probs_l = F.dropout(probs_l, p=self.dropout, training=self.training)
        out_v = torch.matmul(probs_v, val_l).transpose(1, 2).reshape(B, Tv, self.embed_dim)
        out_l = torch.matmul(probs_l, val_v).transpose(1, 2).reshape(B, Tl, self.embed_dim)
        return self.out_v_proj(out_v), self.out_l_proj(out_l)

    @staticmethod
    def _mask_term(mask, like):
        """What the reference adds to the logits for a [B, T] mask (fuse_helper.py:405-416)."""
        assert mask.dim() == 2
        m = mask[:, None, None, :]
        if m.dtype == torch.bool:
            # bool.masked_fill(mask == 0, -9e15) is True everywhere -> +1.0 on every logit (Q1)
            return torch.ones((), dtype=like.dtype, device=like.device)
        return m.masked_fill(m == 0, -9e15).to(like.dtype)


class RLIPv2_BiAttentionBlockForCheckpoint(nn.Module):
    def __init__(self, v_dim, l_dim, embed_dim, num_heads, hidden_dim=None, dropout=0.1, drop_path=.0,
                 init_values=1e-4, args=None):
        super().__init__()
        self.layer_norm_v = nn.LayerNorm(v_dim)
        self.layer_norm_l = nn.LayerNorm(l_dim)
        self.attn = RLIPv2_BiMultiHeadAttention(v_dim=v_dim, l_dim=l_dim, embed_dim=embed_dim, num_heads=num_heads,
                                                dropout=dropout, args=args)
        assert drop_path == 0.0, "DropPath is the identity in every RLIPv2 script (fuse_helper.py:1012)"
        self.gamma_v = nn.Parameter(init_values * torch.ones((v_dim)), requires_grad=True)
        self.gamma_l = nn.Parameter(init_values * torch.ones((l_dim)), requires_grad=True)
        g = self.gating_mechanism = args.gating_mechanism
        if g in ("Stanh", "SDFtanh", "SFtanh", "SDFXAc", "SXAc", "SXAcLN", "SDFXAcLN"):
            self.gamma_v_down = nn.Linear(v_dim, v_dim // 4)
            self.gamma_v_up = nn.Linear(v_dim // 4, v_dim)
            self.gamma_l_down = nn.Linear(l_dim, l_dim // 4)
            self.gamma_l_up = nn.Linear(l_dim // 4, l_dim)
        if g in ("SXAcLN", "SDFXAcLN"):
            self.layer_norm_gating_v = nn.LayerNorm(v_dim // 4)
            self.layer_norm_gating_l = nn.LayerNorm(l_dim // 4)
        if g in ("SOtanh", "SDFOXAcLN"):
            self.gamma_v_down = nn.Linear(v_dim, v_dim // 2)
            self.gamma_v_one = nn.Linear(v_dim // 2, 1)
            self.gamma_l_down = nn.Linear(l_dim, l_dim // 2)
            self.gamma_l_one = nn.Linear(l_dim // 2, 1)
        if g == "SDFOXAcLN":
            self.layer_norm_gating_v = nn.LayerNorm(v_dim // 2)
            self.layer_norm_gating_l = nn.LayerNorm(l_dim // 2)
        if g == "MBF":
            self.MBF_v = MultiBranchFusion(v_dim, v_dim, v_dim, 16)
            self.MBF_l = MultiBranchFusion(l_dim, l_dim, l_dim, 16)

    def _gates(self, delta_v, delta_l):
        """(gate_v, gate_l) multiplying the attention deltas; fuse_helper.py:695-750."""
        g = self.gating_mechanism
        gv, gl = self.gamma_v, self.gamma_l
        relu, tanh = torch.relu, torch.tanh
        if g == "GLIP":
            return gv, gl
        if g == "Vtanh":
            return tanh(gv[0]), tanh(gl[0])
        if g == "Etanh":
            return tanh(gv), tanh(gl)
        if g == "Stanh":
            return tanh(self.gamma_v_up(relu(self.gamma_v_down(gv)))), tanh(self.gamma_l_up(relu(self.gamma_l_down(gl))))
        if g == "SDFtanh":
            return (tanh(self.gamma_v_up(relu(self.gamma_v_down(delta_v)))),
                    tanh(self.gamma_l_up(relu(self.gamma_l_down(delta_l)))))
        if g == "SOtanh":
            return (tanh(self.gamma_v_one(relu(self.gamma_v_down(gv)))),
                    tanh(self.gamma_l_one(relu(self.gamma_l_down(gl)))))
        if g == "VXAc":
            return gv[0], gl[0]
        if g == "SXAc":
            return self.gamma_v_up(relu(self.gamma_v_down(gv))), self.gamma_l_up(relu(self.gamma_l_down(gl)))
        if g == "SDFXAc":
            return self.gamma_v_up(relu(self.gamma_v_down(delta_v))), self.gamma_l_up(relu(self.gamma_l_down(delta_l)))
        if g == "SXAcLN":
            return (self.gamma_v_up(relu(self.layer_norm_gating_v(self.gamma_v_down(gv)))),
                    self.gamma_l_up(relu(self.layer_norm_gating_l(self.gamma_l_down(gl)))))
        if g == "SDFXAcLN":
            return (self.gamma_v_up(relu(self.layer_norm_gating_v(self.gamma_v_down(delta_v)))),
                    self.gamma_l_up(relu(self.layer_norm_gating_l(self.gamma_l_down(delta_l)))))
        if g == "SDFOXAcLN":
            return (self.gamma_v_one(relu(self.layer_norm_gating_v(self.gamma_v_down(delta_v)))),
                    self.gamma_l_one(relu(self.layer_norm_gating_l(self.gamma_l_down(delta_l)))))
        if g == "XGating":
            return None, None
        raise AssertionError(f"unknown gating mechanism {g}")

    def forward(self, q, l, q_pos=None, attention_mask_l=None, attention_mask_v=None, dummy_tensor=None):
        v = self.layer_norm_v(q)
        l = self.layer_norm_l(l)
        delta_v, delta_l = self.attn(v, l, q_pos, attention_mask_l=attention_mask_l,
                                     attention_mask_v=attention_mask_v)
        if self.gating_mechanism == "MBF":
            return self.MBF_v(v, delta_v), self.MBF_l(l, delta_l), None, None, None
        gate_v, gate_l = self._gates(delta_v, delta_l)
        if gate_v is None:
            v, l = v + delta_v, l + delta_l                      # Q3: residual on the normalised input
        else:
            v, l = v + gate_v * delta_v, l + gate_l * delta_l
        return v, l, None, None, None


class RLIPv2_VLFuse(nn.Module):
    """Early-fusion wrapper: dict in, dict out, as the encoder calls it (fuse_helper.py:1047-1096)."""

    def __init__(self, args):
        super().__init__()
        self.args = args
        self.lang_model = args.text_encoder_type
        self.joint_embedding_size = 256
        self.n_head = 8
        self.embed_dim = 2048
        self.i2t_hidden_dim = 3072
        self.lang_dim = 768 if self.lang_model in ("bert-base-uncased", "roberta-base", "clip") else 1024
        self.use_checkpoint_fusion = bool(getattr(args, "use_checkpoint_fusion", False))
        if args.fusion_type == "GLIP_attn":
            self.b_attn = RLIPv2_BiAttentionBlockForCheckpoint(
                v_dim=self.joint_embedding_size, l_dim=self.lang_dim, embed_dim=self.embed_dim,
                num_heads=self.n_head, hidden_dim=self.i2t_hidden_dim, dropout=0.1, drop_path=.0,
                init_values=1.0 / args.num_feature_levels, args=args)

    def forward(self, x):
        vis, lang = x["visual"], x["lang"]
        if self.args.fusion_type == "GLIP_attn":
            call = self.b_attn
            fn_args = (vis['src'], lang['hidden'], vis['pos'], lang['masks'], vis['padding_mask'])
            if self.use_checkpoint_fusion and self.training:
                from torch.utils import checkpoint
                q, l0, _, _, _ = checkpoint.checkpoint(call, *fn_args, use_reentrant=False)
            else:
                q, l0, _, _, _ = call(*fn_args)
            vis['src'] = q
            lang['hidden'] = l0
        return {"visual": vis, "lang": lang}


# ------------------------------------------------------------------------------------------------
# sparse language layer (RoBERTa-base encoder layer, post-LN)
# ------------------------------------------------------------------------------------------------
fused_text_attention = True       # (attribute, see fused_attention)


class _RobertaSelfAttention(nn.Module):
    def __init__(self, hidden, heads, attn_dropout):
        super().__init__()
        self.num_attention_heads = heads
        self.attention_head_size = hidden // heads
        self.query = nn.Linear(hidden, hidden)
        self.key = nn.Linear(hidden, hidden)
        self.value = nn.Linear(hidden, hidden)
        self.dropout = nn.Dropout(attn_dropout)

    def forward(self, x, additive_mask):
        B, T, C = x.shape
        if fused_text_attention and x.is_cuda and not torch.is_autocast_enabled():
            # one GEMM for the three projections (their parameters share one buffer: deform_attn.adjacent_cat_n) and the
            # library's fused attention on strided views of its output: the matmul / scale / mask / softmax / dropout /
            # matmul chain and the head permutes' copies were ~25 launch-bound launches per layer and direction
            from .deform_attn import adjacent_cat_n
            from .linear import token_linear
            w = adjacent_cat_n(self, "_qkv_w", (self.query.weight, self.key.weight, self.value.weight))
            b = adjacent_cat_n(self, "_qkv_b", (self.query.bias, self.key.bias, self.value.bias))
            qkv = token_linear(x, w, b).view(B, T, 3, self.num_attention_heads, self.attention_head_size)
            q, k, v = (qkv[:, :, i].transpose(1, 2) for i in range(3))
            out = F.scaled_dot_product_attention(q, k, v, attn_mask=additive_mask,
                                                 dropout_p=self.dropout.p if self.training else 0.0)
            return out.transpose(1, 2).reshape(B, T, C)
        split = lambda t: t.view(B, T, self.num_attention_heads, self.attention_head_size).permute(0, 2, 1, 3)
        q, k, v = split(self.query(x)), split(self.key(x)), split(self.value(x))
        scores = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(self.attention_head_size)
        if additive_mask is not None:
            scores = scores + additive_mask
        probs = self.dropout(F.softmax(scores, dim=-1))
        return torch.matmul(probs, v).permute(0, 2, 1, 3).reshape(B, T, C)


class _RobertaSelfOutput(nn.Module):
    def __init__(self, hidden_in, hidden, eps, dropout):
        super().__init__()
        self.dense = nn.Linear(hidden_in, hidden)
        self.LayerNorm = nn.LayerNorm(hidden, eps=eps)
        self.dropout = nn.Dropout(dropout)

    def forward(self, h, residual):
        return self.LayerNorm(self.dropout(self.dense(h)) + residual)


class _RobertaAttention(nn.Module):
    def __init__(self, hidden, heads, eps, dropout, attn_dropout):
        super().__init__()
        self.self = _RobertaSelfAttention(hidden, heads, attn_dropout)
        self.output = _RobertaSelfOutput(hidden, hidden, eps, dropout)

    def forward(self, x, additive_mask):
        return self.output(self.self(x, additive_mask), x)


class _RobertaIntermediate(nn.Module):
    def __init__(self, hidden, inner):
        super().__init__()
        self.dense = nn.Linear(hidden, inner)

    def forward(self, x):
        return F.gelu(self.dense(x))            # exact (erf) GELU: HF ACT2FN["gelu"]


class RobertaLayer(nn.Module):
    """Self-attention (12 x 64) + GELU FFN, post-LayerNorm.  `attention_mask` is [B, T] with 1/True
    for tokens to attend to; it enters as the additive mask (1 - mask) * -10000 of transformers
    4.5.1's get_extended_attention_mask (the version the reference pins)."""

    def __init__(self, hidden_size=768, num_attention_heads=12, intermediate_size=3072, layer_norm_eps=1e-5,
                 hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1):
        super().__init__()
        self.attention = _RobertaAttention(hidden_size, num_attention_heads, layer_norm_eps, hidden_dropout_prob,
                                           attention_probs_dropout_prob)
        self.intermediate = _RobertaIntermediate(hidden_size, intermediate_size)
        self.output = _RobertaSelfOutput(intermediate_size, hidden_size, layer_norm_eps, hidden_dropout_prob)

    def forward(self, hidden_states, attention_mask=None):
        additive = None
        if attention_mask is not None:
            additive = (1.0 - attention_mask[:, None, None, :].to(torch.float32)) * -10000.0
            additive = additive.to(hidden_states.dtype)
        a = self.attention(hidden_states, additive)
        return self.output(self.intermediate(a), a)
