"""RLIPv2-ParSeD (v2): the non-DAB sibling of ParSeDA used by BASELINE config 1 -- same ALIF-fused
encoder and heads, learned query position embeddings and 2-d reference points instead of dynamic
anchor boxes.

Reference: TransformerDecoderHOI (models/deformable_transformer.py:390-481), its decoder layer
(:887-941, identical to the DAB file's layer), RLIP_ParSeDTransformer_v2
(models/ParSetransformer.py:404-911) and RLIP_ParSeD (models/hoi.py:2840-3315).  Parameter names
match the reference (`ho_encoder`, `reference_points_sub/obj`, `verb_query_embed`, `query_embed`).
"""
from __future__ import annotations

import torch
from torch import nn
from torch.nn.init import constant_, normal_, xavier_uniform_

from .alif import RLIPv2_VLFuse, RobertaLayer
from .blocks import FeatureResizer, MultiBranchFusion, inverse_sigmoid
from .decoder import DeformableTransformerDecoderLayer
from .deform_attn import MSDeformAttn
from .encoder import DeformableTransformerEncoderLayer, RLIPv2_DeformableTransformerEncoder, _clones
from .parseda import RLIP_ParSeDA, RLIP_ParSeDABDeformableTransformer_v2, _add_reference, default_args


class TransformerDecoderHOI(nn.Module):
    """Learned query positions; reference points are 2-d in the first layer and become refined 4-d boxes
    (detached between layers) when box heads are attached."""

    def __init__(self, decoder_layer, num_layers, return_intermediate=False, ParSe=False):
        super().__init__()
        self.layers = _clones(decoder_layer, num_layers)
        self.num_layers = num_layers
        self.return_intermediate = return_intermediate
        self.ParSe = ParSe
        self.sub_bbox_embed = None
        self.obj_bbox_embed = None
        self.class_embed = None

    def forward(self, tgt, reference_points, src, src_spatial_shapes, src_level_start_index, src_valid_ratios,
                query_pos=None, src_padding_mask=None, attn_mask=None, key_padding_mask=None):
        assert attn_mask is None and key_padding_mask is None, "verb-tagger masks are outside the RLIPv2 path"
        output = tgt
        sub_ref, obj_ref = reference_points
        assert sub_ref.shape[1] == obj_ref.shape[1]
        n_pair = obj_ref.shape[1]
        inter, inter_sub, inter_obj = [], [], []
        for lid, layer in enumerate(self.layers):
            ratios = src_valid_ratios if sub_ref.shape[-1] == 2 else torch.cat([src_valid_ratios, src_valid_ratios], -1)
            if self.ParSe:
                ref_in = torch.cat((sub_ref[:, :, None] * ratios[:, None], obj_ref[:, :, None] * ratios[:, None]), dim=1)
            else:
                ref_in = 0.5 * (sub_ref + obj_ref)[:, :, None] * ratios[:, None]
            output = layer(output, query_pos, ref_in, src, src_spatial_shapes, src_level_start_index, src_padding_mask)
            if self.sub_bbox_embed is not None:
                h = output[:, :n_pair] if self.ParSe else output
                sub_ref = _add_reference(self.sub_bbox_embed[lid](h), sub_ref).sigmoid().detach()
            if self.obj_bbox_embed is not None:
                h = output[:, n_pair:] if self.ParSe else output
                obj_ref = _add_reference(self.obj_bbox_embed[lid](h), obj_ref).sigmoid().detach()
            if self.return_intermediate:
                inter.append(output)
                inter_sub.append(sub_ref)
                inter_obj.append(obj_ref)
        if self.return_intermediate:
            refs = torch.stack((torch.stack(inter_sub), torch.stack(inter_obj)), dim=0).transpose(0, 1)
            return torch.stack(inter), refs
        return output, reference_points


class RLIP_ParSeDTransformer_v2(RLIP_ParSeDABDeformableTransformer_v2):
    """Shares phase A (feature flattening, text, ALIF encoder, cache) with the DAB transformer; the
    encoder is registered as `ho_encoder` and phase B uses learned query embeddings."""

    def __init__(self, d_model=256, nhead=8, num_encoder_layers=6, num_decoder_layers=6, dim_feedforward=1024,
                 dropout=0.1, activation="relu", return_intermediate_dec=False, num_feature_levels=4, dec_n_points=4,
                 enc_n_points=4, two_stage=False, two_stage_num_proposals=300, pass_pos_and_query=True,
                 text_encoder_type="roberta-base", freeze_text_encoder=False, args=None, text_encoder=None):
        nn.Module.__init__(self)
        assert not two_stage and args.fusion_type == "GLIP_attn"
        self.d_model, self.nhead, self.two_stage = d_model, nhead, two_stage
        self.fusion_type = args.fusion_type
        dec_layer = DeformableTransformerDecoderLayer(d_model, dim_feedforward, dropout, activation,
                                                      num_feature_levels, nhead, dec_n_points)
        self.ho_decoder = TransformerDecoderHOI(dec_layer, num_decoder_layers, return_intermediate_dec, ParSe=True)
        self.verb_decoder = TransformerDecoderHOI(dec_layer, num_decoder_layers, return_intermediate_dec, ParSe=False)
        self.level_embed = nn.Parameter(torch.Tensor(num_feature_levels, d_model))
        enc_layer = DeformableTransformerEncoderLayer(d_model, dim_feedforward, dropout, activation,
                                                      num_feature_levels, nhead, enc_n_points)
        self.ho_encoder = RLIPv2_DeformableTransformerEncoder(
            enc_layer, RobertaLayer(), RLIPv2_VLFuse(args), num_encoder_layers, fusion_interval=args.fusion_interval,
            fusion_last_vis=args.fusion_last_vis, lang_aux_loss=args.lang_aux_loss)
        self.reference_points_sub = nn.Linear(d_model, 2)
        self.reference_points_obj = nn.Linear(d_model, 2)
        # reference _reset_parameters (:527-539)
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        for m in self.modules():
            if isinstance(m, MSDeformAttn):
                m._reset_parameters()
        xavier_uniform_(self.reference_points_sub.weight.data, gain=1.0)
        constant_(self.reference_points_sub.bias.data, 0.)
        xavier_uniform_(self.reference_points_obj.weight.data, gain=1.0)
        constant_(self.reference_points_obj.bias.data, 0.)
        normal_(self.level_embed)
        self.text_encoder = text_encoder
        if text_encoder is not None and freeze_text_encoder:
            for p in self.text_encoder.parameters():
                p.requires_grad_(False)
        self.resizer = FeatureResizer(input_feat_size=768, output_feat_size=d_model, dropout=0.1)
        self.verb_query_tgt_type = args.verb_query_tgt_type
        if "MBF" in self.verb_query_tgt_type:
            self.verb_tgt_generator = MultiBranchFusion(256, 256, 256, 16)
            self.verb_query_embed = nn.Embedding(args.num_queries // 2, d_model)

    @property
    def encoder(self):          # phase A of the parent class calls self.encoder
        return self.ho_encoder

    def forward(self, srcs=None, masks=None, pos_embeds=None, query_embed=None, text=None, encode_and_save=True,
                text_memory=None, img_memory=None, text_attention_mask=None, obj_pred_names_sums=None,
                spatial_shapes=None, level_start_index=None, valid_ratios=None, encoded_text=None, **unused):
        if encode_and_save:
            mc = super().forward(srcs=srcs, masks=masks, pos_embeds=pos_embeds, query_embed=query_embed, text=text,
                                 encode_and_save=True, encoded_text=encoded_text)
            mc["key_padding_mask"] = None
            mc["attn_mask"] = None
            return mc
        bs, _, c = img_memory.shape
        q_pos, q_tgt, verb_tgt = torch.split(query_embed, c, dim=1)
        q_pos = q_pos.unsqueeze(0).expand(bs, -1, -1)
        q_tgt = q_tgt.unsqueeze(0).expand(bs, -1, -1)
        nq = q_pos.shape[1]
        ref_sub = self.reference_points_sub(q_pos[:, :nq // 2]).float().sigmoid()
        ref_obj = self.reference_points_obj(q_pos[:, nq // 2:]).float().sigmoid()
        init_reference = (ref_sub, ref_obj)
        hs_ho, inter_refs = self.ho_decoder(q_tgt, init_reference, img_memory, spatial_shapes, level_start_index,
                                            valid_ratios, q_pos, masks)
        last_sub, last_obj = hs_ho[-1][:, :nq // 2], hs_ho[-1][:, nq // 2:]
        merged = (verb_tgt[:nq // 2] + verb_tgt[nq // 2:]).unsqueeze(0).expand(bs, -1, -1)
        kind = self.verb_query_tgt_type
        if kind == "vanilla":
            verb_pos, verb_in = last_sub + last_obj, merged
        elif kind == "MBF":
            verb_pos = self.verb_query_embed.weight.unsqueeze(0).expand(bs, -1, -1)
            verb_in = self.verb_tgt_generator(last_sub, last_obj)
        elif kind == "vanilla_MBF":
            verb_pos = self.verb_query_embed.weight.unsqueeze(0).expand(bs, -1, -1)
            verb_in = self.verb_tgt_generator(last_sub, last_obj) + merged
        else:
            raise AssertionError(kind)
        hs_verb, _ = self.verb_decoder(verb_in, inter_refs[-1], img_memory, spatial_shapes, level_start_index,
                                       valid_ratios, verb_pos, masks)
        n_layer = hs_ho.shape[0]
        if text_memory.dim() == 4 and text_memory.shape[0] == n_layer:
            text_dec = text_memory
        else:
            text_dec = text_memory.unsqueeze(0).repeat(n_layer, 1, 1, 1)
        return hs_ho, hs_verb, text_dec, init_reference, inter_refs, hs_ho, hs_verb, None, None


class RLIP_ParSeD(RLIP_ParSeDA):
    """Same input projections, heads and two-phase protocol as RLIP_ParSeDA; one `query_embed` table of
    [position | target | verb target] rows (reference hoi.py:2862)."""

    def __init__(self, backbone, transformer, num_queries, num_feature_levels, aux_loss=True, with_box_refine=False,
                 two_stage=False, subject_class=False, verb_curing=False, masked_entity_modeling=None,
                 pseudo_verb=False, matcher=None, verb_tagger=False, args=None):
        assert not two_stage and not verb_curing and not masked_entity_modeling and not verb_tagger
        super().__init__(backbone, transformer, num_queries, num_feature_levels, aux_loss=aux_loss,
                         with_box_refine=with_box_refine, two_stage=False, use_dab=True, subject_class=subject_class,
                         pseudo_verb=pseudo_verb, args=args)
        del self.tgt_embed, self.verb_tgt_embed, self.refpoint_embed
        self.query_embed = nn.Embedding(num_queries, transformer.d_model * 3)

    def _query_embeds(self):
        return self.query_embed.weight


def build_parsed(backbone, args=None, text_encoder=None):
    """--RLIP_ParSeD_v2 assembly (scripts/RLIP_ParSeD/train_RLIP_ParSeD_v2_vg_resnet.sh: FFN 1024, XGating)."""
    if args is None:
        args = default_args(dim_feedforward=1024, gating_mechanism="XGating")
    transformer = RLIP_ParSeDTransformer_v2(
        d_model=args.hidden_dim, nhead=args.nheads, num_encoder_layers=args.enc_layers,
        num_decoder_layers=args.dec_layers, dim_feedforward=args.dim_feedforward, dropout=args.dropout,
        activation="relu", return_intermediate_dec=True, num_feature_levels=args.num_feature_levels,
        dec_n_points=args.dec_n_points, enc_n_points=args.enc_n_points, two_stage=False,
        text_encoder_type=args.text_encoder_type, freeze_text_encoder=args.freeze_text_encoder, args=args,
        text_encoder=text_encoder)
    return RLIP_ParSeD(backbone, transformer, num_queries=args.num_queries, num_feature_levels=args.num_feature_levels,
                       aux_loss=args.aux_loss, with_box_refine=args.with_box_refine, subject_class=args.subject_class,
                       pseudo_verb=args.pseudo_verb, args=args)
