"""RLIPv2-ParSeDA: the two-phase transformer and the top-level model.

Reference: RLIP_ParSeDABDeformableTransformer_v2 (models/dab_deformable/deformable_transformer.py:234-744)
and RLIP_ParSeDA (models/hoi.py:1871-2256).  Constructor arguments, the two-phase
``model(samples, encode_and_save=True, text=...) -> memory_cache`` /
``model(samples, encode_and_save=False, memory_cache=...) -> outputs`` protocol (engine.py:99-100),
output keys and state_dict names follow the reference.

Text: the reference tokenises label strings on the CPU and runs RoBERTa-base every step
(deformable_transformer.py:486-522).  Here ``text`` may be
  * the pre-encoded tuple the reference's evaluation path uses,
    ``(text_attention_mask [n_text, N] bool, text_memory [n_text, N, 768], obj_pred_names_sums [[n_obj, n_verb]])``
    (deformable_transformer.py:569-595), or
  * ``{"input_ids": [n_text, T], "attention_mask": [n_text, T], "obj_pred_names_sums": [[n_obj, n_verb]]}``
    -- token ids for ``self.text_encoder`` (any module returning an object with `.pooler_output`);
    one embedding per label, shared by the whole batch (Q12), masked by the sign of its feature sum (Q2).
"""
from __future__ import annotations


import math
from types import SimpleNamespace

import torch
import torch.nn.functional as F
from torch import nn
from torch.nn.init import normal_

from .msda import attach_host_shapes
from .alif import RLIPv2_VLFuse, RobertaLayer
from .blocks import (MLP, FeatureResizer, MultiBranchFusion, NestedTensor, inverse_sigmoid,
                     nested_tensor_from_tensor_list)
from .decoder import DABDeformableTransformerDecoderHOI, DeformableTransformerDecoderLayer, box_head
from .deform_attn import MSDeformAttn
from .linear import add_row_vector, shared_input
from .encoder import DeformableTransformerEncoderLayer, RLIPv2_DeformableTransformerEncoder, _clones


# encode the label texts on a second HIP stream, concurrently with the backbone (RLIP_ParSeDA._encode)
overlap_text_encoder = True
text_encoder_first = False


def default_args(**overrides):
    """The flags of scripts/RLIP_ParSeDA/train_RLIP_ParSeDA_v2_mixed_vgcoco_resnet.sh that reach the model."""
    a = SimpleNamespace(
        hidden_dim=256, nheads=8, enc_layers=6, dec_layers=3, dim_feedforward=2048, dropout=0.0,
        num_feature_levels=4, enc_n_points=4, dec_n_points=4, num_queries=200, with_box_refine=True,
        subject_class=True, fusion_type="GLIP_attn", gating_mechanism="VXAc", verb_query_tgt_type="vanilla_MBF",
        fusion_interval=2, fusion_last_vis=True, lang_aux_loss=True, pseudo_verb=True, use_dab=True,
        text_encoder_type="roberta-base", freeze_text_encoder=False, use_checkpoint_fusion=False,
        stable_softmax_2d=False, clamp_min_for_underflow=False, clamp_max_for_overflow=False,
        separate_bidirectional=False, do_lang_proj_outside_checkpoint=False, aux_loss=True)
    for k, v in overrides.items():
        setattr(a, k, v)
    return a



def _strided_conv_as_gemm(x, conv):
    """The extra pyramid level's convolution (3x3, stride 2, 2048 -> 256 on the last backbone map: reference
    models/hoi.py:1946-1951) as im2col + one batched GEMM, token-major result [N, H_out * W_out, C_out].

    Why not MIOpen: its forward solver for this shape is the ONE kernel of the whole train step whose output is not
    repeatable bit for bit (tools/nondet_modules.py, profiles/r03_nondeterminism.txt) -- it made the loss and with it
    every gradient differ by 2-5 % between two runs of the same step -- and MIOpen's deterministic mode costs 1.5 s per
    step.  im2col (`unfold`) + GEMM + col2im (`fold`, a gather) are all repeatable; 1 092 output pixels x 18 432."""
    N, C, H, W = x.shape
    kh, kw = conv.kernel_size
    Ho = (H + 2 * conv.padding[0] - kh) // conv.stride[0] + 1
    Wo = (W + 2 * conv.padding[1] - kw) // conv.stride[1] + 1
    cols = F.unfold(x, (kh, kw), padding=conv.padding, stride=conv.stride)          # [N, C * kh * kw, Ho * Wo]
    w2d = conv.weight.reshape(conv.out_channels, C * kh * kw)
    y = torch.matmul(w2d, cols)                                                      # [N, C_out, Ho * Wo]
    if conv.bias is not None:
        y = y + conv.bias.view(1, -1, 1)
    return y.transpose(1, 2).contiguous(), (Ho, Wo)


class LevelViews(list):
    """Per-level [N, 256, H, W] views of one flattened [N, S, 256] tensor (`.flat`)."""
    flat = None


class RLIP_ParSeDABDeformableTransformer_v2(nn.Module):
    def __init__(self, d_model=256, nhead=8, num_encoder_layers=6, num_decoder_layers=6, dim_feedforward=1024,
                 dropout=0.1, activation="relu", return_intermediate_dec=False, num_feature_levels=4,
                 dec_n_points=4, enc_n_points=4, two_stage=False, two_stage_num_proposals=300, use_dab=False,
                 high_dim_query_update=False, no_sine_embed=False, pass_pos_and_query=True,
                 text_encoder_type="roberta-base", freeze_text_encoder=False, args=None, text_encoder=None):
        super().__init__()
        assert not two_stage, "two-stage proposals are not part of the RLIPv2 ParSeDA path"
        assert use_dab, "RLIP_ParSeDA uses dynamic anchor boxes"
        assert args.fusion_type == "GLIP_attn", "the ParSeDA scripts use ALIF fusion (GLIP_attn)"
        self.d_model, self.nhead = d_model, nhead
        self.two_stage, self.use_dab = two_stage, use_dab
        self.fusion_type = args.fusion_type

        dec_layer = DeformableTransformerDecoderLayer(d_model, dim_feedforward, dropout, activation,
                                                      num_feature_levels, nhead, dec_n_points)
        self.ho_decoder = DABDeformableTransformerDecoderHOI(
            dec_layer, num_decoder_layers, return_intermediate_dec, use_dab=use_dab, d_model=d_model,
            high_dim_query_update=high_dim_query_update, no_sine_embed=no_sine_embed, ParSe=True)
        self.verb_decoder = DABDeformableTransformerDecoderHOI(
            dec_layer, num_decoder_layers, return_intermediate_dec, use_dab=use_dab, d_model=d_model,
            high_dim_query_update=high_dim_query_update, no_sine_embed=no_sine_embed, ParSe=False)
        self.verb_tgt_generator = MultiBranchFusion(256, 256, 256, 16)
        self.level_embed = nn.Parameter(torch.Tensor(num_feature_levels, d_model))

        enc_layer = DeformableTransformerEncoderLayer(d_model, dim_feedforward, dropout, activation,
                                                      num_feature_levels, nhead, enc_n_points)
        self.encoder = RLIPv2_DeformableTransformerEncoder(
            enc_layer, RobertaLayer(), RLIPv2_VLFuse(args), num_encoder_layers,
            fusion_interval=args.fusion_interval, fusion_last_vis=args.fusion_last_vis,
            lang_aux_loss=args.lang_aux_loss)
        self.high_dim_query_update = high_dim_query_update
        self._reset_parameters()

        # text encoder: supplied by the caller (RoBERTa-base shaped); hidden size 768
        self.text_encoder = text_encoder
        if text_encoder is not None and freeze_text_encoder:
            for p in self.text_encoder.parameters():
                p.requires_grad_(False)
        self.resizer = FeatureResizer(input_feat_size=768, output_feat_size=d_model, dropout=0.1)
        self.verb_query_tgt_type = args.verb_query_tgt_type
        if "MBF" in self.verb_query_tgt_type:
            self.verb_tgt_generator = MultiBranchFusion(256, 256, 256, 16)

    def _reset_parameters(self):
        # reference :364-374 -- xavier on every matrix, then the MSDeformAttn-specific init, then N(0,1) levels
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        for m in self.modules():
            if isinstance(m, MSDeformAttn):
                m._reset_parameters()
        normal_(self.level_embed)

    @staticmethod
    def get_valid_ratio(mask):
        _, H, W = mask.shape
        valid_h = torch.sum(~mask[:, :, 0], 1).float() / H
        valid_w = torch.sum(~mask[:, 0, :], 1).float() / W
        return torch.stack([valid_w, valid_h], -1)

    # ---- phase A ---------------------------------------------------------------------------------
    def _encode_text(self, text, bs, device):
        if isinstance(text, tuple):                       # pre-encoded (reference :569-571)
            text_attention_mask, text_memory, sums = text
            return text_attention_mask, text_memory, sums
        cache = getattr(self, "label_cache", None)
        if cache is not None and not torch.cuda.is_current_stream_capturing():
            # opt-in (`transformer.label_cache = text_batch.LabelEmbeddingCache()`): identical labels encoded once, and with
            # a frozen encoder in eval mode served from the table across calls (host-side keys: not inside a capture)
            pooled = cache.encode(self.text_encoder, text["input_ids"], text["attention_mask"])
        else:
            ids, am = text["input_ids"].to(device), text["attention_mask"].to(device)
            pooled = self.text_encoder(input_ids=ids, attention_mask=am).pooler_output      # [n_text, 768]
        text_memory = pooled[:, None, :]                                                    # [n_text, 1, 768]
        text_attention_mask = ~(text_memory.sum(dim=-1) > 0)                                # Q2
        if text_memory.shape[1] != bs:                                                      # Q12
            text_memory = text_memory.repeat(1, bs, 1)
            text_attention_mask = text_attention_mask.repeat(1, bs)
        return text_attention_mask, text_memory, text["obj_pred_names_sums"]

    def forward(self, srcs=None, masks=None, pos_embeds=None, query_embed=None, text=None, encode_and_save=True,
                text_memory=None, img_memory=None, text_attention_mask=None, obj_pred_names_sums=None,
                spatial_shapes=None, level_start_index=None, valid_ratios=None, spatial_shapes_list=None,
                encoded_text=None, no_padding=False):
        assert query_embed is not None
        if encode_and_save:
            shapes_list = [tuple(s.shape[-2:]) for s in srcs]
            bs = srcs[0].shape[0]
            src_flatten = getattr(srcs, "flat", None)        # (levels already flattened by the model's projection)
            if src_flatten is None:
                src_flatten = torch.cat([s.flatten(2).transpose(1, 2) for s in srcs], 1)
            mask_flatten = torch.cat([m.flatten(1) for m in masks], 1)
            level_rows = self.level_embed.unbind(0)          # (one backward node instead of a select per level)
            lvl_pos = torch.cat([add_row_vector(p.flatten(2).transpose(1, 2), level_rows[l])
                                 for l, p in enumerate(pos_embeds)], 1)
            # device-resident int64 metadata, built once per pyramid shape (a host->device copy per step
            # would also be illegal inside a HIP-graph capture)
            key = (tuple(shapes_list), str(src_flatten.device))
            cache = self.__dict__.setdefault("_shape_cache", {})
            if key not in cache:
                sp = torch.as_tensor(shapes_list, dtype=torch.long, device=src_flatten.device)
                attach_host_shapes(sp, shapes_list)      # the MSDA op sizes grids / checks sum(H*W) from the host copy
                cache[key] = (sp, torch.cat((sp.new_zeros((1,)), sp.prod(1).cumsum(0)[:-1])))
            spatial_shapes, level_start_index = cache[key]
            if no_padding:
                # (all-False masks, the caller's host-side hint: every ratio is exactly 1 -- ~40 tiny launches per step for
                #  a constant)
                vkey = ("valid_ratios", bs, len(masks), str(src_flatten.device))
                if vkey not in cache:
                    cache[vkey] = torch.ones(bs, len(masks), 2, dtype=torch.float32, device=src_flatten.device)
                valid_ratios = cache[vkey]
            else:
                valid_ratios = torch.stack([self.get_valid_ratio(m) for m in masks], 1)

            if encoded_text is None:
                encoded_text = self._encode_text(text, bs, src_flatten.device)
            text_attention_mask, text_memory, obj_pred_names_sums = encoded_text
            img_memory, lang = self.encoder(src_flatten, spatial_shapes, level_start_index, valid_ratios, lvl_pos,
                                            mask_flatten, lang_hidden=text_memory.transpose(0, 1),
                                            lang_masks=text_attention_mask.transpose(0, 1),
                                            spatial_shapes_list=shapes_list, skip_value_mask=no_padding)
            # [N, n_text, 768] or, with lang_aux_loss, [n_fusions, N, n_text, 768] -> text-major + resize
            text_memory_resized = self.resizer(lang.transpose(0, 1) if lang.dim() == 3 else lang.transpose(1, 2))
            return {
                # raw label features on the token path; on the pre-encoded path the reference's variable
                # has been overwritten by the encoder's language output by now (:571 vs :599)
                "text_memory_bf_resize": lang if isinstance(text, tuple) else text_memory,
                "text_memory_resized": text_memory_resized,
                "text_memory": text_memory_resized,
                "img_memory": img_memory,
                # (only consumed as MSDeformAttn's value mask in the decoders: an all-False mask is dropped)
                "masks": None if no_padding else mask_flatten,
                "text_attention_mask": text_attention_mask,
                "pos_embed": lvl_pos,
                "ho_query_embed": query_embed,
                "obj_pred_names_sums": obj_pred_names_sums,
                "spatial_shapes": spatial_shapes,
                "level_start_index": level_start_index,
                "valid_ratios": valid_ratios,
                "spatial_shapes_list": shapes_list,
            }

        # ---- phase B --------------------------------------------------------------------------------
        bs = img_memory.shape[0]
        d = self.d_model
        anchors = query_embed[..., 2 * d:].float().sigmoid()        # the box chain stays in float32
        nq = query_embed.shape[0]
        ref_sub, ref_obj = anchors[:nq // 2], anchors[nq // 2:]
        tgt = query_embed[..., :d].unsqueeze(0).expand(bs, -1, -1)
        verb_tgt = query_embed[..., d:2 * d].unsqueeze(0).expand(bs, -1, -1)
        init_reference = (ref_sub, ref_obj)

        # (the image memory is read by the value projections of all decoder layers and by nothing else below)
        img_memory = shared_input(img_memory)
        hs_ho, inter_refs = self.ho_decoder(tgt, init_reference, img_memory, spatial_shapes, level_start_index,
                                            valid_ratios, query_pos=None, src_padding_mask=masks)
        if getattr(hs_ho, "deltas", None) is not None:          # (the decoder's own split of its last layer output)
            last_sub, last_obj = hs_ho.deltas[-1][2], hs_ho.deltas[-1][3]
        else:
            last_sub, last_obj = getattr(hs_ho, "layers", hs_ho)[-1].split(nq // 2, dim=1)
        kind = self.verb_query_tgt_type
        verb_a, verb_b = verb_tgt.split(nq // 2, dim=1)         # (one node; two slices are two zero-padded gradients + a sum)
        if kind == "vanilla":
            verb_in = verb_a + verb_b
        elif kind == "MBF":
            verb_in = self.verb_tgt_generator(last_sub, last_obj)
        elif kind == "vanilla_MBF":
            verb_in = self.verb_tgt_generator(last_sub, last_obj) + verb_a + verb_b
        else:
            raise AssertionError(kind)
        hs_verb, _ = self.verb_decoder(verb_in, inter_refs[-1], img_memory, spatial_shapes, level_start_index,
                                       valid_ratios, query_pos=None, src_padding_mask=masks)
        n_layer = hs_ho.shape[0]
        if text_memory.dim() == 4 and text_memory.shape[0] == n_layer:
            text_dec = text_memory
        else:
            text_dec = text_memory.unsqueeze(0).repeat(n_layer, 1, 1, 1)
        return hs_ho, hs_verb, text_dec, init_reference, inter_refs, hs_ho, hs_verb, None, None


batched_heads = True          # (tools/r04_host_ab.py flips the attribute for its A/B)


def _add_reference(delta, ref):
    """box head output + inverse_sigmoid(reference): all 4 coordinates for a reference box, only (x, y)
    for a 2-d reference point (reference hoi.py:2122-2138 / :3040-3056)."""
    inv = inverse_sigmoid(ref)
    delta = delta.to(inv.dtype)              # no mixed-dtype elementwise ops (see decoder.py)
    if ref.shape[-1] == 4:
        return delta + inv
    assert ref.shape[-1] == 2
    return torch.cat([delta[..., :2] + inv, delta[..., 2:]], dim=-1)


class RLIP_ParSeDA(nn.Module):
    def __init__(self, backbone, transformer, num_queries, num_feature_levels, aux_loss=True, with_box_refine=True,
                 two_stage=False, use_dab=True, num_patterns=0, random_refpoints_xy=False, subject_class=False,
                 pseudo_verb=False, args=None):
        super().__init__()
        assert use_dab and not two_stage and num_patterns == 0
        self.num_queries = num_queries
        self.transformer = transformer
        hidden = transformer.d_model
        self.num_feature_levels = num_feature_levels
        self.use_dab, self.num_patterns, self.random_refpoints_xy = use_dab, num_patterns, random_refpoints_xy
        self.projection_text = nn.Linear(hidden, hidden)
        self.bias_c = -math.log((1 - 0.01) / 0.01)
        self.bias_obj_a = nn.Parameter(torch.zeros((256,), dtype=torch.float32), requires_grad=True)
        self.bias_pred_a = nn.Parameter(torch.zeros((256,), dtype=torch.float32), requires_grad=True)
        self.tgt_embed = nn.Embedding(num_queries, hidden)
        self.verb_tgt_embed = nn.Embedding(num_queries, hidden)
        self.refpoint_embed = nn.Embedding(num_queries, 4)
        if random_refpoints_xy:
            self.refpoint_embed.weight.data[:, :2].uniform_(0, 1)
            self.refpoint_embed.weight.data[:, :2] = inverse_sigmoid(self.refpoint_embed.weight.data[:, :2])

        n_out = len(backbone.strides)
        projs = []
        for i in range(n_out):
            projs.append(nn.Sequential(nn.Conv2d(backbone.num_channels[i], hidden, kernel_size=1),
                                       nn.GroupNorm(32, hidden)))
        in_ch = backbone.num_channels[n_out - 1]
        for _ in range(num_feature_levels - n_out):
            projs.append(nn.Sequential(nn.Conv2d(in_ch, hidden, kernel_size=3, stride=2, padding=1),
                                       nn.GroupNorm(32, hidden)))
            in_ch = hidden
        self.input_proj = nn.ModuleList(projs)
        self.backbone = backbone
        self.aux_loss, self.with_box_refine, self.two_stage = aux_loss, with_box_refine, two_stage
        for proj in self.input_proj:
            nn.init.xavier_uniform_(proj[0].weight, gain=1)
            nn.init.constant_(proj[0].bias, 0)

        sub = MLP(hidden, hidden, 4, 3)
        obj = MLP(hidden, hidden, 4, 3)
        for head in (sub, obj):
            nn.init.constant_(head.layers[-1].weight.data, 0)
            nn.init.constant_(head.layers[-1].bias.data, 0)
        n_pred = transformer.ho_decoder.num_layers
        if with_box_refine:
            # 2*n_pred clones each: the first half refines inside the ho decoder and predicts the
            # boxes, the second half is handed to the verb decoder (reference hoi.py:1980-1990)
            self.sub_bbox_embed = _clones(sub, n_pred * 2)
            self.obj_bbox_embed = _clones(obj, n_pred * 2)
            nn.init.constant_(self.sub_bbox_embed[0].layers[-1].bias.data[2:], -2.0)
            nn.init.constant_(self.obj_bbox_embed[0].layers[-1].bias.data[2:], -2.0)
            transformer.ho_decoder.sub_bbox_embed = self.sub_bbox_embed[:n_pred]
            transformer.verb_decoder.sub_bbox_embed = self.sub_bbox_embed[n_pred:]
            transformer.ho_decoder.obj_bbox_embed = self.obj_bbox_embed[:n_pred]
            transformer.verb_decoder.obj_bbox_embed = self.obj_bbox_embed[n_pred:]
            transformer.ho_decoder.keep_box_deltas = True      # the heads below reuse the refinement's MLP outputs
        else:
            nn.init.constant_(sub.layers[-1].bias.data[2:], -2.0)
            nn.init.constant_(obj.layers[-1].bias.data[2:], -2.0)
            self.sub_bbox_embed = nn.ModuleList([sub for _ in range(n_pred)])
            self.obj_bbox_embed = nn.ModuleList([obj for _ in range(n_pred)])
        self.subject_class = subject_class
        self.pseudo_verb = pseudo_verb
        self.pseudo_verb_mode = "online"

    def _query_embeds(self):
        return torch.cat((self.tgt_embed.weight, self.verb_tgt_embed.weight, self.refpoint_embed.weight), 1)

    def _transformer_phase_b(self, mc):
        return self.transformer(
            masks=mc["masks"], query_embed=mc["ho_query_embed"], encode_and_save=False,
            text_memory=mc["text_memory_resized"], img_memory=mc["img_memory"],
            text_attention_mask=mc["text_attention_mask"], obj_pred_names_sums=mc["obj_pred_names_sums"],
            spatial_shapes=mc["spatial_shapes"], level_start_index=mc["level_start_index"],
            valid_ratios=mc["valid_ratios"])

    # ---- phase A: backbone + input projections + ALIF encoder -------------------------------------------
    def _encode(self, samples, text):
        # The label texts do not depend on the images: encode them on a second HIP stream while the
        # backbone runs.  The text encoder is a chain of ~400 tiny kernels on [n_text, 5] tokens (4 ms of
        # launch latency, ~0 compute) and the backbone is compute-bound, so the two overlap almost
        # perfectly; autograd replays each backward on the stream of its forward, so the backward passes
        # overlap the same way.
        encoded_text = None
        tr = self.transformer
        if (isinstance(text, dict) and samples.tensors.is_cuda and getattr(tr, "text_encoder", None) is not None
                and overlap_text_encoder):
            cur = torch.cuda.current_stream()
            side = self.__dict__.setdefault("_text_stream", torch.cuda.Stream(device=samples.tensors.device))
            fork = cur.record_event()
        if 'fork' in locals() and text_encoder_first:      # (the round-1 issue order; attribute for A/B runs)
            side.wait_event(fork)
            with torch.cuda.stream(side):
                encoded_text = tr._encode_text(text, samples.tensors.shape[0], samples.tensors.device)
        features, pos = self.backbone(samples)
        if encoded_text is None and 'fork' in locals():
            # Issued AFTER the backbone, forked from BEFORE it (the event): the two still run side by side, but the
            # text encoder's autograd nodes are now the younger ones, so the backward pass runs the text encoder
            # BEFORE the backbone -- its gradients (60 % of the all-reduce bytes) are complete while the backbone's
            # backward still computes, and the data-parallel step reduces them underneath it (train.GradientSynchronizer)
            side.wait_event(fork)
            with torch.cuda.stream(side):
                encoded_text = tr._encode_text(text, samples.tensors.shape[0], samples.tensors.device)
        srcs, masks = [], []
        fast = self._project_levels_token_major(features)
        if fast is not None:
            # (every level projected + normalised token-major, straight into the flattened [N, S, 256] tensor; the
            #  per-level NCHW tensors handed on are views of it, `srcs.flat` lets the transformer skip its cat)
            srcs = fast
            masks = [feat.decompose()[1] for feat in features]
            for l in range(len(features), self.num_feature_levels):
                mask = F.interpolate(samples.mask[None].float(), size=srcs[l].shape[-2:]).to(torch.bool)[0]   # Q13
                pos.append(self.backbone[1](NestedTensor(srcs[l], mask, getattr(samples, "no_padding", False)),
                                            out_dtype=srcs[l].dtype))
                masks.append(mask)
        else:
            for l, feat in enumerate(features):
                src, mask = feat.decompose()
                assert mask is not None
                srcs.append(self.input_proj[l](src))
                masks.append(mask)
            for l in range(len(srcs), self.num_feature_levels):
                src = self.input_proj[l](features[-1].tensors if l == len(features) else srcs[-1])
                mask = F.interpolate(samples.mask[None].float(), size=src.shape[-2:]).to(torch.bool)[0]   # Q13
                pos.append(self.backbone[1](NestedTensor(src, mask, getattr(samples, "no_padding", False)),
                                            out_dtype=src.dtype))
                srcs.append(src)
                masks.append(mask)
        query_embeds = self._query_embeds()
        if encoded_text is not None:
            torch.cuda.current_stream().wait_stream(side)
            for t in encoded_text[:2]:
                t.record_stream(torch.cuda.current_stream())
        return self.transformer(srcs=srcs, masks=masks, pos_embeds=pos, query_embed=query_embeds, text=text,
                                encode_and_save=True, encoded_text=encoded_text,
                                no_padding=bool(getattr(samples, "no_padding", False)))

    def _project_levels_token_major(self, features):
        """input_proj (reference models/hoi.py:1936-1957) on channels-last bf16 features without leaving the token-major
        layout: the 1x1 convolutions are GEMMs over [N*H*W, C_in] views, the extra level's 3x3 stride-2 convolution stays
        on MIOpen, and one GroupNorm launch pair writes all levels into the flattened tensor (`norm.level_group_norm`).
        Returns None when the fast path does not apply (float32 / autocast runs, more than one extra level, other
        widths): the caller falls back to the module-by-module form."""
        from .linear import token_linear
        from .norm import level_group_norm, level_group_norm_supported
        n_out = len(features)
        if self.num_feature_levels - n_out > 1 or self.num_feature_levels > 4 or torch.is_autocast_enabled():
            return None
        xs, shapes = [], []
        for l in range(self.num_feature_levels):
            conv = self.input_proj[l][0]
            x = features[min(l, n_out - 1)].tensors
            if not (x.is_cuda and x.dtype == torch.bfloat16 and conv.weight.dtype == torch.bfloat16
                    and x.is_contiguous(memory_format=torch.channels_last)):
                return None
            if l < n_out:
                if conv.kernel_size != (1, 1) or conv.stride != (1, 1):
                    return None
                N, Cin, H, W = x.shape
                y = token_linear(x.permute(0, 2, 3, 1).reshape(N, H * W, Cin), conv.weight.view(conv.out_channels, Cin),
                                 conv.bias)
            else:
                y, (H, W) = _strided_conv_as_gemm(x, conv)              # [N, H*W, C_out], deterministic
                N = x.shape[0]
            xs.append(y)
            shapes.append((H, W))
        norms = [proj[1] for proj in self.input_proj]
        if not level_group_norm_supported(xs, norms):
            return None
        flat = level_group_norm(xs, norms)                              # [N, S, 256]
        srcs, start = LevelViews(), 0
        for (H, W), y in zip(shapes, xs):
            srcs.append(flat[:, start:start + H * W].view(flat.shape[0], H, W, flat.shape[2]).permute(0, 3, 1, 2))
            start += H * W
        srcs.flat = flat
        return srcs

    def forward(self, samples, encode_and_save=True, memory_cache=None, **kwargs):
        if not isinstance(samples, NestedTensor):
            samples = nested_tensor_from_tensor_list(samples)
        if encode_and_save:
            return self._encode(samples, kwargs['text'])

        mc = memory_cache
        hs_ho, hs_verb, text_dec, init_reference, inter_references, _, _, _, _ = self._transformer_phase_b(mc)
        half = self.num_queries // 2
        # per-layer decoder outputs (the tensors the layers produced, not selects of their stack), split once into the
        # subject / object halves: one cat in the backward pass instead of slice gradients
        ho_layers, verb_layers = getattr(hs_ho, "layers", hs_ho), getattr(hs_verb, "layers", hs_verb)
        sums = mc["obj_pred_names_sums"]
        n_obj, n_verb = int(sums[:, 0].max()), int(sums[:, 1].max())

        sub_cls, obj_cls, verb_cls, sub_box, obj_box = [], [], [], [], []
        hs_h, hs_o = [None] * len(ho_layers), [None] * len(ho_layers)
        deltas = getattr(hs_ho, "deltas", None)
        dec = self.transformer.ho_decoder
        if deltas is not None and not (len(deltas) == len(ho_layers) and all(
                dec.sub_bbox_embed[k] is self.sub_bbox_embed[k] and dec.obj_bbox_embed[k] is self.obj_bbox_embed[k]
                for k in range(len(ho_layers)))):
            deltas = None                                   # (not the heads' own modules: compute them here)
        # Text projections and logits of ALL decoder layers in one pass (hoi.py:2140-2157 runs them per layer): one normalise,
        # one projection GEMM, one batched product per decoder over (layer, image) -- the per-layer loop was ~19 launch-bound
        # kernels per layer forward and twice that backward.  The stacked decoder outputs enter whole (their backward is the
        # stack's unbind: views), the per-layer logits leave as views of the batched result.
        n_lay = len(ho_layers)
        batched = (batched_heads and torch.is_tensor(hs_ho) and hs_ho.dim() == 4 and torch.is_tensor(hs_verb) and hs_verb.dim() == 4
                   and hs_ho.shape[0] == n_lay == hs_verb.shape[0] and text_dec.shape[0] >= n_lay)
        if batched:
            text_all = F.normalize(text_dec[:n_lay].transpose(1, 2).float(), p=2, dim=-1)            # [L, N, n_text, C]
            proj_all = self.projection_text((text_all / 2.0).to(self.projection_text.weight.dtype))
            assert n_obj + n_verb == proj_all.shape[2]
            obj_text_t = proj_all[:, :, :n_obj].transpose(2, 3)
            verb_text_t = proj_all[:, :, n_obj:n_obj + n_verb].transpose(2, 3)
            ho_cls_all = torch.matmul(hs_ho + self.bias_obj_a, obj_text_t) + self.bias_c           # [L, N, nq, n_obj]
            verb_cls_all = torch.matmul(hs_verb + self.bias_pred_a, verb_text_t) + self.bias_c     # [L, N, nq/2, n_verb]
            # (per-layer views through ONE split + unbinds: a select / slice per layer and head would each be a zero-filled
            #  full-size gradient + copy + accumulation in the backward)
            sub_cls_all, obj_cls_all = ho_cls_all.split(half, dim=2)
            sub_cls_lay, obj_cls_lay, verb_cls_lay = sub_cls_all.unbind(0), obj_cls_all.unbind(0), verb_cls_all.unbind(0)
        for lvl in range(len(ho_layers)):
            if deltas is not None:
                hs_h[lvl], hs_o[lvl] = deltas[lvl][2], deltas[lvl][3]
            else:
                hs_h[lvl], hs_o[lvl] = ho_layers[lvl].split(half, dim=1)
            ref_s, ref_o = init_reference if lvl == 0 else inter_references[lvl - 1]
            # the decoder has already applied these heads to these layer outputs for its box refinement (same modules,
            # same inputs): their results arrive with the layers, the MLPs run once per step
            if deltas is not None:
                d_sub, d_obj = deltas[lvl][:2]
            else:
                d_sub, d_obj = self.sub_bbox_embed[lvl](hs_h[lvl]), self.obj_bbox_embed[lvl](hs_o[lvl])
            sub_box.append(box_head(d_sub, ref_s))
            obj_box.append(box_head(d_obj, ref_o))
            if batched:
                obj_cls.append(obj_cls_lay[lvl])
                verb_cls.append(verb_cls_lay[lvl])
                if self.subject_class:
                    sub_cls.append(sub_cls_lay[lvl])
                continue
            text = F.normalize(text_dec[lvl].transpose(0, 1).float(), p=2, dim=-1)       # float32 norm
            proj = self.projection_text((text / 2.0).to(self.projection_text.weight.dtype))
            assert n_obj + n_verb == proj.shape[1]
            obj_text, verb_text = proj[:, :n_obj], proj[:, n_obj:n_obj + n_verb]
            obj_cls.append(torch.matmul(hs_o[lvl] + self.bias_obj_a, obj_text.transpose(1, 2)) + self.bias_c)
            verb_cls.append(torch.matmul(verb_layers[lvl] + self.bias_pred_a, verb_text.transpose(1, 2)) + self.bias_c)
            if self.subject_class:
                sub_cls.append(torch.matmul(hs_h[lvl] + self.bias_obj_a, obj_text.transpose(1, 2)) + self.bias_c)

        keys = ["pred_obj_logits", "pred_verb_logits", "pred_sub_boxes", "pred_obj_boxes"]
        stacks = [obj_cls, verb_cls, sub_box, obj_box]
        if self.subject_class:
            keys, stacks = ["pred_sub_logits"] + keys, [sub_cls] + stacks
        out = {k: v[-1] for k, v in zip(keys, stacks)}
        if self.aux_loss:
            out["aux_outputs"] = [{k: v[i] for k, v in zip(keys, stacks)} for i in range(len(obj_cls) - 1)]

        if self.pseudo_verb:
            # soft verb targets from pairwise distances of the raw verb label features (reference :2197-2239)
            verb_feat = mc["text_memory_bf_resize"][:, 0][n_obj:n_obj + n_verb]
            # F.pairwise_distance adds eps=1e-6 to the difference before the norm (the reference uses it)
            dist = F.pairwise_distance(verb_feat.repeat(1, n_verb).view(-1, verb_feat.shape[1]),
                                       verb_feat.repeat(n_verb, 1).view(-1, verb_feat.shape[1]), p=2).view(n_verb, n_verb)
            sim = dist.max(-1)[0].unsqueeze(-1) - dist
            tgt_verbs = torch.cat([t['verb_labels'] for t in kwargs['targets']])
            tvs = (tgt_verbs.unsqueeze(-1).repeat(1, 1, sim.shape[-1]) * sim).sum(dim=1)
            if tgt_verbs.shape[0] > 0:
                tvs = tvs / tvs.max(-1)[0].unsqueeze(-1)
            tvs[tgt_verbs.bool()] = 0
            tvs = tvs * (tvs > 0.3)
            out["target_verb_sim"] = tvs
            if self.aux_loss:
                for aux in out["aux_outputs"]:
                    aux["target_verb_sim"] = tvs
        return out


def build_parseda(backbone, args=None, text_encoder=None):
    """Assemble the model the way models/detr.py:552-567 + models/transformer.py:1344 do for
    --RLIP_ParSeDA_v2 (return_intermediate_dec=True, two_stage=False, use_dab=True)."""
    args = default_args() if args is None else args
    transformer = RLIP_ParSeDABDeformableTransformer_v2(
        d_model=args.hidden_dim, nhead=args.nheads, num_encoder_layers=args.enc_layers,
        num_decoder_layers=args.dec_layers, dim_feedforward=args.dim_feedforward, dropout=args.dropout,
        activation="relu", return_intermediate_dec=True, num_feature_levels=args.num_feature_levels,
        dec_n_points=args.dec_n_points, enc_n_points=args.enc_n_points, two_stage=False,
        two_stage_num_proposals=args.num_queries, use_dab=True, text_encoder_type=args.text_encoder_type,
        freeze_text_encoder=args.freeze_text_encoder, args=args, text_encoder=text_encoder)
    return RLIP_ParSeDA(backbone, transformer, num_queries=args.num_queries,
                        num_feature_levels=args.num_feature_levels, aux_loss=args.aux_loss,
                        with_box_refine=args.with_box_refine, two_stage=False, use_dab=True,
                        subject_class=args.subject_class, pseudo_verb=args.pseudo_verb, args=args)
