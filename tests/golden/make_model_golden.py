"""Generate module- and model-level golden vectors from the imported reference (build container
only).  Weights are filled in closed form from the parameter NAMES (ref_harness.fill_closed_form),
inputs from seeded generators recorded in the file, so only inputs + outputs are stored.

Goldens (tests/golden/model_*.npz), all in eval() mode and float32 unless noted:
  msdeformattn_2d / _4d   MSDeformAttn module, 2-d and 4-d reference points, padding mask (float64)
  vlfuse_VXAc / _XGating  RLIPv2_VLFuse block with masks (exhibits Q1)
  roberta_layer           RobertaLayer with a mask containing masked tokens (Q2)
  encoder                 RLIPv2_DeformableTransformerEncoder, 2 layers / 1 fusion, fusion_last_vis on and off
  decoder_ho / _verb      DABDeformableTransformerDecoderHOI ParSe=True / False, 2 layers, box refine
  mbf                     MultiBranchFusion
  parsed                  full RLIP_ParSeD v2 (config-1 family: non-DAB decoders, XGating), outputs + aux + gradients
  criterion               SetCriterionHOI + HungarianMatcherHOI loss dict and gradients for fixed predictions
  parseda                 full RLIP_ParSeDA (enc 4 / dec 2, nq 20, 12 texts, two images of different
                          size): all outputs + aux, and gradients w.r.t. the input features and a few
                          sentinel parameters

usage: python tests/golden/make_model_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as R  # noqa: E402

PYR = [(8, 10), (4, 5), (2, 3), (1, 2)]


def rng_tensor(seed, *shape, scale=1.0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g, dtype=torch.float64) * scale).to(dtype)


def level_meta():
    shapes = torch.tensor(PYR, dtype=torch.long)
    starts = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    return shapes, starts, int(shapes.prod(1).sum())


def padding_mask(N, frac=(1.0, 0.75)):
    """[N, S] bool, image n valid on the top-left frac[n] part of every level."""
    masks = []
    for (H, W) in PYR:
        m = torch.ones(N, H, W, dtype=torch.bool)
        for n in range(N):
            m[n, : max(1, int(round(H * frac[n]))), : max(1, int(round(W * frac[n])))] = False
        masks.append(m)
    return masks


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        out[k] = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
    path = os.path.join(HERE, f"model_{name}.npz")
    np.savez_compressed(path, **out)
    print(f"{name:20s} -> {os.path.getsize(path)/1024:.0f} KiB")


def gold_msdeformattn():
    from models.ops.modules.ms_deform_attn import MSDeformAttn
    shapes, starts, S = level_meta()
    for nd in (2, 4):
        torch.manual_seed(0)
        m = MSDeformAttn(256, 4, 8, 4).double()
        R.fill_closed_form(m)
        with torch.no_grad():           # keep the compass-direction bias, shrink the random offset weights
            m.sampling_offsets.weight.mul_(0.3)
        N, Lq = 2, 9
        query = rng_tensor(1, N, Lq, 256, dtype=torch.float64).requires_grad_(True)
        inp = rng_tensor(2, N, S, 256, dtype=torch.float64).requires_grad_(True)
        g = torch.Generator().manual_seed(3)
        ref = torch.rand(N, Lq, 4, nd, generator=g, dtype=torch.float64)
        if nd == 4:
            ref[..., 2:] = ref[..., 2:] * 0.4 + 0.05
        mask = torch.cat([x.flatten(1) for x in padding_mask(N)], 1)
        out = m(query, ref, inp, shapes, starts, mask)
        go = rng_tensor(4, *out.shape, dtype=torch.float64)
        out.backward(go)
        save(f"msdeformattn_{nd}d", query=query, ref=ref, inp=inp, mask=mask, out=out, go=go, g_query=query.grad,
             g_inp=inp.grad, g_value_proj_b=m.value_proj.bias.grad, g_off_b=m.sampling_offsets.bias.grad)


def gold_vlfuse():
    from models.fuse_helper import RLIPv2_VLFuse
    for gating in ("VXAc", "XGating"):
        args = R.reference_args(gating_mechanism=gating)
        delattr(args, "use_checkpoint_fusion") if hasattr(args, "use_checkpoint_fusion") else None
        args.use_checkpoint_fusion = False
        m = RLIPv2_VLFuse(args).eval()
        R.fill_closed_form(m)
        N, Tv, Tl = 2, 12, 7
        v = rng_tensor(10, N, Tv, 256).requires_grad_(True)
        pos = rng_tensor(11, N, Tv, 256)
        l = rng_tensor(12, N, Tl, 768).requires_grad_(True)
        vmask = torch.ones(N, Tv, dtype=torch.bool); vmask[1, 8:] = False      # True = valid (already inverted)
        lmask = torch.ones(N, Tl, dtype=torch.bool); lmask[:, 5:] = False
        out = m({"visual": {"src": v, "padding_mask": vmask, "pos": pos}, "lang": {"hidden": l, "masks": lmask}})
        ov, ol = out["visual"]["src"], out["lang"]["hidden"]
        gv, gl = rng_tensor(13, *ov.shape), rng_tensor(14, *ol.shape)
        (ov * gv).sum().add((ol * gl).sum()).backward()
        save(f"vlfuse_{gating}", v=v, pos=pos, l=l, vmask=vmask, lmask=lmask, out_v=ov, out_l=ol, gv=gv, gl=gl,
             g_v=v.grad, g_l=l.grad)


def gold_roberta():
    from models.modeling_roberta import RobertaLayer
    from transformers import RobertaConfig
    m = RobertaLayer(RobertaConfig.from_pretrained("roberta-base")).eval()
    R.fill_closed_form(m)
    N, T = 2, 9
    x = rng_tensor(20, N, T, 768).requires_grad_(True)
    mask = torch.ones(N, T, dtype=torch.bool); mask[0, 6:] = False; mask[1, 2] = False
    out = m(hidden_states=x, attention_mask=mask)
    g = rng_tensor(21, *out.shape)
    out.backward(g)
    save("roberta_layer", x=x, mask=mask, out=out, g=g, g_x=x.grad)


def _encoder(last_vis):
    from models.dab_deformable.deformable_transformer import DeformableTransformerEncoderLayer
    from models.deformable_transformer import RLIPv2_DeformableTransformerEncoder
    from models.fuse_helper import RLIPv2_VLFuse
    from models.modeling_roberta import RobertaLayer
    from transformers import RobertaConfig
    args = R.reference_args()
    args.use_checkpoint_fusion = False
    enc = RLIPv2_DeformableTransformerEncoder(
        DeformableTransformerEncoderLayer(256, 512, 0.0, "relu", 4, 8, 4),
        RobertaLayer(RobertaConfig.from_pretrained("roberta-base")), RLIPv2_VLFuse(args), 2, fusion_interval=2,
        fusion_last_vis=last_vis, lang_aux_loss=True).eval()
    R.fill_closed_form(enc)
    with torch.no_grad():
        for layer in enc.layers:
            layer.self_attn.sampling_offsets.weight.mul_(0.3)
    return enc


def gold_encoder():
    shapes, starts, S = level_meta()
    for last_vis in (True, False):
        enc = _encoder(last_vis)
        N, Tl = 2, 6
        src = rng_tensor(30, N, S, 256).requires_grad_(True)
        pos = rng_tensor(31, N, S, 256)
        masks = padding_mask(N)
        mask = torch.cat([x.flatten(1) for x in masks], 1)
        vr = torch.stack([torch.stack([(~m[:, 0, :]).sum(1).float() / m.shape[2],
                                       (~m[:, :, 0]).sum(1).float() / m.shape[1]], -1) for m in masks], 1)
        lang = rng_tensor(32, N, Tl, 768).requires_grad_(True)
        lmask = torch.zeros(N, Tl, dtype=torch.bool); lmask[:, 4:] = True      # True = padding (not inverted)
        img, lng = enc(src.clone(), shapes, starts, vr, pos, mask, lang_hidden=lang, lang_masks=lmask)
        gi, gl = rng_tensor(33, *img.shape), rng_tensor(34, *lng.shape)
        (img * gi).sum().add((lng * gl).sum()).backward()
        save(f"encoder_lastvis{int(last_vis)}", src=src, pos=pos, mask=mask, valid_ratios=vr, lang=lang,
             lmask=lmask, img=img, lng=lng, gi=gi, gl=gl, g_src=src.grad, g_lang=lang.grad)


def gold_decoder():
    from models.dab_deformable.deformable_transformer import (DABDeformableTransformerDecoderHOI,
                                                              DeformableTransformerDecoderLayer, MLP, _get_clones)
    shapes, starts, S = level_meta()
    for parse in (True, False):
        layer = DeformableTransformerDecoderLayer(256, 512, 0.0, "relu", 4, 8, 4)
        dec = DABDeformableTransformerDecoderHOI(layer, 2, True, use_dab=True, d_model=256, ParSe=parse).eval()
        dec.sub_bbox_embed = _get_clones(MLP(256, 256, 4, 3), 2)
        dec.obj_bbox_embed = _get_clones(MLP(256, 256, 4, 3), 2)
        R.fill_closed_form(dec)
        with torch.no_grad():
            for l in dec.layers:
                l.cross_attn.sampling_offsets.weight.mul_(0.3)
        N, nq = 2, 8
        npair = nq // 2
        tgt = rng_tensor(40, N, nq if parse else npair, 256).requires_grad_(True)
        src = rng_tensor(41, N, S, 256).requires_grad_(True)
        g = torch.Generator().manual_seed(42)
        if parse:
            refs = (torch.rand(npair, 4, generator=g) * 0.5 + 0.2, torch.rand(npair, 4, generator=g) * 0.5 + 0.2)
        else:
            refs = (torch.rand(N, npair, 4, generator=g) * 0.5 + 0.2, torch.rand(N, npair, 4, generator=g) * 0.5 + 0.2)
        masks = padding_mask(N)
        mask = torch.cat([x.flatten(1) for x in masks], 1)
        vr = torch.stack([torch.stack([(~m[:, 0, :]).sum(1).float() / m.shape[2],
                                       (~m[:, :, 0]).sum(1).float() / m.shape[1]], -1) for m in masks], 1)
        hs, inter = dec(tgt, refs, src, shapes, starts, vr, query_pos=None, src_padding_mask=mask)
        gh = rng_tensor(43, *hs.shape)
        (hs * gh).sum().backward()
        save(f"decoder_parse{int(parse)}", tgt=tgt, src=src, ref_sub=refs[0], ref_obj=refs[1], mask=mask,
             valid_ratios=vr, hs=hs, inter=inter, gh=gh, g_tgt=tgt.grad, g_src=src.grad)


def gold_mbf():
    from models.dab_deformable.deformable_transformer import MultiBranchFusion
    m = MultiBranchFusion(256, 256, 256, 16)
    R.fill_closed_form(m)
    a = rng_tensor(50, 2, 5, 256).requires_grad_(True)
    b = rng_tensor(51, 2, 5, 256).requires_grad_(True)
    out = m(a, b)
    g = rng_tensor(52, *out.shape)
    out.backward(g)
    save("mbf", a=a, b=b, out=out, g=g, g_a=a.grad, g_b=b.grad)


def gold_parseda():
    from models.dab_deformable.deformable_transformer import RLIP_ParSeDABDeformableTransformer_v2
    from models.hoi import RLIP_ParSeDA
    from util.misc import NestedTensor
    args = R.reference_args(num_queries=20, enc_layers=4, dec_layers=2, dim_feedforward=512, pseudo_verb=True)
    args.use_checkpoint_fusion = False
    tr = RLIP_ParSeDABDeformableTransformer_v2(
        d_model=256, nhead=8, num_encoder_layers=4, num_decoder_layers=2, dim_feedforward=512, dropout=0.0,
        activation="relu", return_intermediate_dec=True, num_feature_levels=4, dec_n_points=4, enc_n_points=4,
        two_stage=False, two_stage_num_proposals=20, use_dab=True, args=args)
    bb = R.StandInBackbone(num_channels=(32, 64, 128))
    model = RLIP_ParSeDA(bb, tr, num_queries=20, num_feature_levels=4, aux_loss=True, with_box_refine=True,
                         two_stage=False, use_dab=True, subject_class=True, pseudo_verb=True, args=args).eval()
    R.fill_closed_form(model)
    with torch.no_grad():
        for mod in model.modules():
            if mod.__class__.__name__ == "MSDeformAttn":
                mod.sampling_offsets.weight.mul_(0.3)
        model.refpoint_embed.weight.mul_(8.0)
    N = 2
    img_hw = [(64, 96), (56, 72)]                        # second image smaller -> padding / valid ratios
    H, W = 64, 96
    img_mask = torch.ones(N, H, W, dtype=torch.bool)
    for n, (h, w) in enumerate(img_hw):
        img_mask[n, :h, :w] = False
    feats = []
    for i, (c, s) in enumerate(zip((32, 64, 128), (8, 16, 32))):
        t = rng_tensor(60 + i, N, c, H // s, W // s).requires_grad_(True)
        m = torch.nn.functional.interpolate(img_mask[None].float(), size=t.shape[-2:]).to(torch.bool)[0]
        feats.append((t, m))
    bb.features = feats
    samples = NestedTensor(torch.zeros(N, 3, H, W), img_mask)
    n_obj, n_verb = 7, 5
    text_mem = torch.tanh(rng_tensor(70, n_obj + n_verb, 1, 768)).repeat(1, N, 1)
    text_mask = ~(text_mem.sum(-1) > 0)                                       # Q2
    sums = torch.tensor([[n_obj, n_verb]])
    targets = []
    g = torch.Generator().manual_seed(71)
    for n in range(N):
        vl = torch.zeros(3, n_verb); vl[torch.arange(3), torch.randint(0, n_verb, (3,), generator=g)] = 1
        targets.append({"verb_labels": vl})
    mc = model(samples, encode_and_save=True, text=(text_mask, text_mem, sums), targets=targets)
    # the pseudo-verb targets read the RAW label features, which only the training text path leaves in the
    # cache (deformable_transformer.py:599 vs :571); emulate that path's cache entry here
    rec_bf = mc["text_memory_bf_resize"]
    mc["text_memory_bf_resize"] = text_mem
    out = model(samples, encode_and_save=False, memory_cache=mc, text=(text_mask, text_mem, sums), targets=targets)
    keys = ["pred_sub_logits", "pred_obj_logits", "pred_verb_logits", "pred_sub_boxes", "pred_obj_boxes"]
    loss = 0
    rec = {}
    for i, k in enumerate(keys):
        gk = rng_tensor(80 + i, *out[k].shape)
        loss = loss + (out[k] * gk).sum() + (out["aux_outputs"][0][k] * gk).sum() * 0.5
        rec["g_" + k] = gk
        rec[k] = out[k]
        rec["aux0_" + k] = out["aux_outputs"][0][k]
    loss.backward()
    rec["target_verb_sim"] = out["target_verb_sim"]
    rec["text_memory_bf_resize_eval_path"] = rec_bf
    rec["img_memory"] = mc["img_memory"]
    rec["text_memory_resized"] = mc["text_memory_resized"]
    for i, (t, m) in enumerate(feats):
        rec[f"feat{i}"] = t
        rec[f"featmask{i}"] = m
        rec[f"g_feat{i}"] = t.grad
    rec["img_mask"] = img_mask
    rec["text_mem"] = text_mem
    rec["text_mask"] = text_mask
    for n in range(N):
        rec[f"verb_labels{n}"] = targets[n]["verb_labels"]
    sd = dict(model.named_parameters(remove_duplicate=False))
    for name in ("transformer.level_embed", "tgt_embed.weight", "transformer.encoder.VLFuse_layers.0.b_attn.gamma_v",
                 "transformer.ho_decoder.layers.1.cross_attn.sampling_offsets.bias", "projection_text.weight",
                 "input_proj.3.0.weight", "sub_bbox_embed.0.layers.2.bias", "sub_bbox_embed.3.layers.0.weight"):
        gname = "gparam_" + name.replace(".", "__")
        gr = sd[name].grad
        rec[gname] = gr if gr is not None else torch.zeros(0)
    rec["param_names"] = np.array(sorted(sd.keys()))
    save("parseda", **rec)


def criterion_case():
    """Fixed predictions (main + 1 aux layer) and targets for the criterion golden."""
    N, nq, n_obj, n_verb = 2, 10, 7, 5

    def outs(seed):
        g = torch.Generator().manual_seed(seed)
        return {'pred_sub_logits': torch.randn(N, nq, n_obj, generator=g),
                'pred_obj_logits': torch.randn(N, nq, n_obj, generator=g),
                'pred_verb_logits': torch.randn(N, nq, n_verb, generator=g),
                'pred_sub_boxes': torch.rand(N, nq, 4, generator=g) * 0.4 + 0.2,
                'pred_obj_boxes': torch.rand(N, nq, 4, generator=g) * 0.4 + 0.2}
    g = torch.Generator().manual_seed(5)
    main, aux = outs(1), [outs(2)]
    tvs = torch.rand(5, n_verb, generator=g) * 0.5
    targets = []
    for n, k in enumerate((3, 2)):
        vl = torch.zeros(k, n_verb)
        vl[torch.arange(k), torch.randint(0, n_verb, (k,), generator=g)] = 1
        ob = torch.rand(k, 4, generator=g) * 0.3 + 0.2
        if n == 1:
            ob[1] = 0                                   # a relation without an object box
        targets.append({'obj_labels': torch.randint(0, n_obj - 1, (k,), generator=g),
                        'sub_labels': torch.randint(0, n_obj - 1, (k,), generator=g), 'verb_labels': vl,
                        'sub_boxes': torch.rand(k, 4, generator=g) * 0.3 + 0.2, 'obj_boxes': ob})
    tvs[torch.cat([t['verb_labels'] for t in targets]).bool()] = 0
    for o in [main] + aux:
        o['target_verb_sim'] = tvs
    return main, aux, targets, (N, nq, n_obj, n_verb)


def gold_criterion():
    """SetCriterionHOI + HungarianMatcherHOI in the ParSeDA script configuration (models/hoi.py:3627,
    models/matcher.py:95; weights models/detr.py:571-620 with set_cost_bbox 2.5 / bbox_loss_coef 2.5)."""
    cwd = os.getcwd()
    os.chdir(R.REF)                                     # the ctor loads datasets/priors/hico_verb_samples.npz
    from models.hoi import SetCriterionHOI
    from models.matcher import HungarianMatcherHOI
    args = R.reference_args()
    args.verb_tagger = False
    main, aux, targets, (N, nq, n_obj, n_verb) = criterion_case()
    for o in [main] + aux:
        for k in o:
            if k.startswith('pred_'):
                o[k].requires_grad_(True)
    out = dict(main)
    out['aux_outputs'] = aux
    matcher = HungarianMatcherHOI(cost_obj_class=1, cost_verb_class=1, cost_bbox=2.5, cost_giou=1, subject_class=True)
    crit = SetCriterionHOI(n_obj - 1, nq, n_verb, matcher=matcher, weight_dict={}, eos_coef=0.1,
                           losses=['obj_labels', 'verb_labels', 'sub_obj_boxes', 'obj_cardinality'],
                           verb_loss_type='focal', obj_loss_type='cross_entropy', matching_symmetric=False,
                           RLIP_ParSe=False, subject_class=True, use_no_verb_token=False, giou_verb_label=True,
                           verb_curing=False, pseudo_verb=True, triplet_filtering=False, naive_obj_smooth=0,
                           naive_verb_smooth=0, args=args)
    ld = crit(out, [dict(t) for t in targets])
    os.chdir(cwd)
    w = {'loss_obj_ce': 1.0, 'loss_verb_ce': 1.0, 'loss_sub_bbox': 2.5, 'loss_obj_bbox': 2.5, 'loss_sub_giou': 1.0,
         'loss_obj_giou': 1.0}
    w.update({k + '_0': v for k, v in list(w.items())})
    total = sum(ld[k] * w[k] for k in ld if k in w)
    total.backward()
    rec = {"total": total}
    for li, o in enumerate([main] + aux):
        for k in o:
            if k.startswith('pred_'):
                rec[f"g_L{li}_{k}"] = o[k].grad
    for k, v in ld.items():
        rec["loss_" + k] = v if torch.is_tensor(v) else torch.tensor(float(v))
    save("criterion", **rec)


def gold_parsed():
    """RLIP_ParSeD (v2), the non-DAB sibling used by BASELINE config 1 (models/hoi.py:2840,
    models/ParSetransformer.py:404): XGating, learned query positions, 2-d reference points."""
    from models.ParSetransformer import RLIP_ParSeDTransformer_v2
    from models.hoi import RLIP_ParSeD
    from util.misc import NestedTensor
    args = R.reference_args(num_queries=20, enc_layers=4, dec_layers=2, dim_feedforward=512, pseudo_verb=False,
                            gating_mechanism="XGating", RLIP_ParSeDA_v2=False, RLIP_ParSeD_v2=True, use_dab=False)
    args.use_checkpoint_fusion = False
    args.verb_tagger = False
    tr = RLIP_ParSeDTransformer_v2(
        d_model=256, nhead=8, num_encoder_layers=4, num_decoder_layers=2, dim_feedforward=512, dropout=0.0,
        activation="relu", return_intermediate_dec=True, num_feature_levels=4, dec_n_points=4, enc_n_points=4,
        two_stage=False, two_stage_num_proposals=20, args=args)
    bb = R.StandInBackbone(num_channels=(32, 64, 128))
    model = RLIP_ParSeD(bb, tr, num_queries=20, num_feature_levels=4, aux_loss=True, with_box_refine=True,
                        two_stage=False, subject_class=True, pseudo_verb=False, args=args).eval()
    R.fill_closed_form(model)
    with torch.no_grad():
        for mod in model.modules():
            if mod.__class__.__name__ == "MSDeformAttn":
                mod.sampling_offsets.weight.mul_(0.3)
    N, H, W = 2, 64, 96
    img_hw = [(64, 96), (56, 72)]
    img_mask = torch.ones(N, H, W, dtype=torch.bool)
    for n, (h, w) in enumerate(img_hw):
        img_mask[n, :h, :w] = False
    feats = []
    for i, (c, s_) in enumerate(zip((32, 64, 128), (8, 16, 32))):
        t = rng_tensor(160 + i, N, c, H // s_, W // s_).requires_grad_(True)
        m = torch.nn.functional.interpolate(img_mask[None].float(), size=t.shape[-2:]).to(torch.bool)[0]
        feats.append((t, m))
    bb.features = feats
    samples = NestedTensor(torch.zeros(N, 3, H, W), img_mask)
    n_obj, n_verb = 7, 5
    text_mem = torch.tanh(rng_tensor(170, n_obj + n_verb, 1, 768)).repeat(1, N, 1)
    text_mask = ~(text_mem.sum(-1) > 0)
    sums = torch.tensor([[n_obj, n_verb]])
    mc = model(samples, encode_and_save=True, text=(text_mask, text_mem, sums), targets=None)
    out = model(samples, encode_and_save=False, memory_cache=mc, text=(text_mask, text_mem, sums), targets=None)
    keys = ["pred_sub_logits", "pred_obj_logits", "pred_verb_logits", "pred_sub_boxes", "pred_obj_boxes"]
    loss = 0
    rec = {}
    for i, k in enumerate(keys):
        gk = rng_tensor(180 + i, *out[k].shape)
        loss = loss + (out[k] * gk).sum() + (out["aux_outputs"][0][k] * gk).sum() * 0.5
        rec["g_" + k] = gk
        rec[k] = out[k]
        rec["aux0_" + k] = out["aux_outputs"][0][k]
    loss.backward()
    for i, (t, m) in enumerate(feats):
        rec[f"feat{i}"] = t
        rec[f"featmask{i}"] = m
        rec[f"g_feat{i}"] = t.grad
    rec["img_mask"], rec["text_mem"], rec["text_mask"] = img_mask, text_mem, text_mask
    sd = dict(model.named_parameters(remove_duplicate=False))
    for name in ("query_embed.weight", "transformer.reference_points_sub.weight", "transformer.verb_query_embed.weight",
                 "transformer.ho_encoder.layers.0.linear1.weight"):
        rec["gparam_" + name.replace(".", "__")] = sd[name].grad
    rec["param_names"] = np.array(sorted(sd.keys()))
    save("parsed", **rec)


def postprocess_case():
    """seeded model outputs for 3 images (43 object classes incl. "no object", 21 verbs) + image sizes"""
    bs, nq, n_obj, n_verb = 3, 7, 43, 21
    out = {"pred_obj_logits": rng_tensor(71, bs, nq, n_obj, scale=2.0),
           "pred_sub_logits": rng_tensor(72, bs, nq, n_obj, scale=2.0),
           "pred_verb_logits": rng_tensor(73, bs, nq, n_verb, scale=2.0),
           "pred_sub_boxes": rng_tensor(74, bs, nq, 4).sigmoid(),
           "pred_obj_boxes": rng_tensor(75, bs, nq, 4).sigmoid()}
    # make some subjects "person" (class 0) so that the zero-shot filter keeps a ragged subset per image
    out["pred_sub_logits"][:, ::2, 0] += 8.0
    sizes = torch.tensor([[480, 640], [800, 1333], [333, 500]])
    return out, sizes


def gold_postprocess():
    """PostProcessHOI (models/hoi.py:4769-4873): plain, temperature and zero-shot (subject-filtered) variants.
    The constructor reads datasets/priors/obj_verb_cooccurrence.npz for a buffer its forward never uses (the
    only use is commented out, :4862): the instance is built without running it."""
    from models.hoi import PostProcessHOI
    out, sizes = postprocess_case()
    rec = {}
    for name, kw in (("plain", dict(temperature=False, zero_shot_hoi_eval=False)),
                     ("temperature", dict(temperature=True, zero_shot_hoi_eval=False)),
                     ("zeroshot", dict(temperature=False, zero_shot_hoi_eval=True))):
        pp = PostProcessHOI.__new__(PostProcessHOI)
        torch.nn.Module.__init__(pp)
        pp.subject_category_id, pp.sigmoid, pp.verb_curing = 0, True, False
        pp.temperature, pp.zero_shot_hoi_eval, pp.tao = kw["temperature"], kw["zero_shot_hoi_eval"], 0.07
        res = pp(out, sizes)
        for i, r in enumerate(res):
            for k, v in r.items():
                rec[f"{name}_{i}_{k}"] = v
    save("postprocess", **rec)


SWIN_CFG = dict(embed_dim=24, depths=[2, 2, 2, 2], num_heads=[3, 6, 12, 24], window_size=7, out_indices=(1, 2, 3),
                drop_path_rate=0.0)


def swin_input():
    return rng_tensor(81, 2, 3, 70, 90)              # sizes that need patch / window / merge padding


def gold_swin():
    """SwinTransformer (models/swin/swin_transformer.py:596-763), a 4-stage toy configuration: the three
    output maps and gradients w.r.t. the image, a qkv weight, a merge weight and a relative-position table."""
    from models.swin.swin_transformer import SwinTransformer
    m = SwinTransformer(**SWIN_CFG).eval()
    R.fill_closed_form(m)
    x = swin_input().requires_grad_(True)
    outs = m(x)
    rec = {"keys": np.array(sorted(m.state_dict().keys()))}
    total = 0
    for i, (k, v) in enumerate(sorted(outs.items())):
        rec[k] = v
        total = total + (v * rng_tensor(90 + i, *v.shape)).sum()
    total.backward()
    rec["g_x"] = x.grad
    for name in ("layers.0.blocks.1.attn.qkv.weight", "layers.1.downsample.reduction.weight",
                 "layers.2.blocks.1.attn.relative_position_bias_table", "norm2.weight"):
        rec["g_" + name] = dict(m.named_parameters())[name].grad
    save("swin", **rec)


def main():
    R.install()
    torch.manual_seed(0)
    gold_msdeformattn()
    gold_vlfuse()
    gold_roberta()
    gold_encoder()
    gold_decoder()
    gold_mbf()
    gold_parseda()
    gold_criterion()
    gold_parsed()
    gold_postprocess()
    gold_swin()


if __name__ == "__main__":
    main()
