"""Checkpoint interop with the reference.

The build keeps the reference's state_dict layout (tests/test_modules_cpu.py::
test_state_dict_names_match_reference), so a reference checkpoint -- a dict with the weights under 'model'
(main.py:599-629) -- loads directly.  What main.py does around `load_state_dict` is restated here:

  * `--pretrained`: the learned queries are cut to the run's `num_queries` first
    (util/misc.py:466-490: `filter_ckpt_tgt_anchor` for RLIP_ParSeDA_v2 -- tgt_embed / verb_tgt_embed /
    refpoint_embed rows; `filter_ckpt_query_embed(share_verb_query=True)` for RLIP_ParSeD_v2 -- query_embed
    rows, verb_query_embed to half), then `load_state_dict(strict=False)` and the load report is returned;
  * `--resume`: strict load of the model, optimiser / epoch left to the caller.
"""
from __future__ import annotations

from collections import OrderedDict

import torch


def _weights(checkpoint):
    if isinstance(checkpoint, (str, bytes)) or hasattr(checkpoint, "__fspath__"):
        checkpoint = torch.load(checkpoint, map_location="cpu", weights_only=False)
    return checkpoint["model"] if "model" in checkpoint else checkpoint


def filter_queries(state_dict, num_queries, family="parseda"):
    """the reference's query filters: keep the first `num_queries` learned queries"""
    out = OrderedDict()
    for k, v in state_dict.items():
        if family == "parseda":
            if "tgt_embed" in k or "verb_tgt_embed" in k or "refpoint_embed" in k:
                v = v[:num_queries]
        elif family == "parsed":
            if "query_embed" in k:
                v = v[:num_queries // 2] if "verb_query_embed" in k else v[:num_queries]
        else:
            raise ValueError(f"unknown model family {family!r}")
        out[k] = v
    return out


def load_pretrained(model, checkpoint, num_queries=None, family="parseda"):
    """`--pretrained` semantics: filtered queries, non-strict load; returns (missing, unexpected) key lists"""
    sd = _weights(checkpoint)
    if num_queries is not None:
        sd = filter_queries(sd, num_queries, family)
    info = model.load_state_dict(sd, strict=False)
    return list(info.missing_keys), list(info.unexpected_keys)


def load_resume(model, checkpoint):
    """`--resume` semantics: strict load of checkpoint['model']"""
    model.load_state_dict(_weights(checkpoint))


def master_state_dict(model, optimizer):
    """model.state_dict() with the optimiser's float32 master weights in place of the bf16-rounded parameters
    (the reference's checkpoints hold float32 weights, main.py:599-629); buffers and frozen parameters are
    upcast copies."""
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    names = getattr(optimizer, "names", None)
    masters = getattr(optimizer, "master", None)
    if names is not None and masters is not None:
        for n, m in zip(names, masters):
            if n in sd:
                sd[n] = m.detach().clone().view_as(sd[n])
    return {k: (v.float() if v.is_floating_point() else v) for k, v in sd.items()}


def save_checkpoint(path, model, optimizer=None, epoch=None, extra=None):
    """the reference's layout: {'model', 'optimizer', 'epoch', ...} (engine / util.misc.save_on_master).  With a
    master-weight optimiser (bf16 parameters) 'model' holds the float32 masters, so that save -> resume continues
    from exactly the weights the optimiser was updating."""
    has_master = optimizer is not None and getattr(optimizer, "master", None) is not None
    ckpt = {"model": master_state_dict(model, optimizer) if has_master else model.state_dict()}
    if optimizer is not None:
        ckpt["optimizer"] = optimizer.state_dict()
    if epoch is not None:
        ckpt["epoch"] = epoch
    if extra:
        ckpt.update(extra)
    torch.save(ckpt, path)


# ---- DAB-Deformable-DETR -> RLIPv2-ParSeDA key conversion ----------------------------------------------------------------
# The 80 COCO category ids among the 91 output rows of a DETR-family class head (row index = category id).
COCO_OBJECT_IDS = (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 27, 28, 31, 32, 33, 34,
                   35, 36, 37, 38, 39, 40, 41, 42, 43, 44, 46, 47, 48, 49, 50, 51, 52, 53, 54, 55, 56, 57, 58, 59, 60, 61, 62,
                   63, 64, 65, 67, 70, 72, 73, 74, 75, 76, 77, 78, 79, 80, 81, 82, 84, 85, 86, 87, 88, 89, 90)


def convert_dab_ddetr(checkpoint, dataset="hico", with_box_refine=False, drop_class_embed=False, detreg=False,
                      parse=True, num_layers=6, generator=None):
    """A DAB-Deformable-DETR detection checkpoint -> the key layout an RLIPv2-ParSeDA model loads with
    `load_pretrained` (what the reference's convert_parameters/convert_parameters_DABDDETR.py writes to disk, :48-170;
    `parse` = its --ParSeDABDDETR switch).  Returns the converted checkpoint dict (weights under 'model'); the input
    dict is not modified.

      * an mmdetection checkpoint ('state_dict' instead of 'model', :67-78): its 'bbox_head.' prefixes are dropped and
        only the encoder / decoder duplication below applies;
      * `transformer.encoder.*` is duplicated as `transformer.ho_encoder.*`, `transformer.decoder.*` as
        `transformer.ho_decoder.*` AND `transformer.verb_decoder.*` (:86-92; the originals stay, they are ignored by
        the non-strict load);
      * the per-layer box heads `bbox_embed.i.layers.j` seed both `sub_bbox_embed.i` and `obj_bbox_embed.i` (:94-106);
        with iterative box refinement also the four copies inside the two decoders, from `transformer.decoder.bbox_embed`
        (from `bbox_embed` for a DETReg checkpoint, :108-132);
      * unless dropped, the 91-way class heads become `obj_class_embed.i`: the 80 COCO rows plus one freshly initialised
        "no pair" row (a Linear(256, 1), :60-63, :134-139); V-COCO gets one more fresh row in front of the last (:158-168);
      * `verb_tgt_embed` starts from `tgt_embed` (:146).
    `generator`: torch.Generator for the fresh rows (the reference draws them from the global RNG)."""
    if isinstance(checkpoint, (str, bytes)) or hasattr(checkpoint, "__fspath__"):
        checkpoint = torch.load(checkpoint, map_location="cpu", weights_only=False)      # a path: load first, then look
    mmdet = isinstance(checkpoint, dict) and "state_dict" in checkpoint and "model" not in checkpoint
    src = checkpoint["state_dict"] if mmdet else _weights(checkpoint)
    model = OrderedDict((k.replace("bbox_head.", "") if mmdet else k, v) for k, v in src.items())

    def fresh_row(width, like):
        lin = torch.nn.Linear(width, 1)
        if generator is not None:
            bound = 1.0 / width ** 0.5
            with torch.no_grad():
                lin.weight.copy_((torch.rand(1, width, generator=generator) * 2 - 1) * bound)
                lin.bias.copy_((torch.rand(1, generator=generator) * 2 - 1) * bound)
        return lin.weight.detach().to(like), lin.bias.detach().to(like)

    if parse:
        for k in list(model.keys()):
            if "transformer.encoder" in k:
                model[k.replace("transformer.encoder", "transformer.ho_encoder")] = model[k].clone()
            if "transformer.decoder" in k:
                model[k.replace("transformer.decoder", "transformer.ho_decoder")] = model[k].clone()
                model[k.replace("transformer.decoder", "transformer.verb_decoder")] = model[k].clone()
        if not mmdet:
            ids = list(COCO_OBJECT_IDS) + [91]                  # 91: the appended "no pair" row
            background = None                                   # ONE fresh row shared by all layers (:60-63)
            for i in range(num_layers):
                for j in range(3):
                    for part in ("weight", "bias"):
                        head = model[f"bbox_embed.{i}.layers.{j}.{part}"]
                        model[f"sub_bbox_embed.{i}.layers.{j}.{part}"] = head
                        model[f"obj_bbox_embed.{i}.layers.{j}.{part}"] = head
                        if with_box_refine:
                            inner = head if detreg else model[f"transformer.decoder.bbox_embed.{i}.layers.{j}.{part}"]
                            for dec in ("ho_decoder", "verb_decoder"):
                                model[f"transformer.{dec}.sub_bbox_embed.{i}.layers.{j}.{part}"] = inner
                                model[f"transformer.{dec}.obj_bbox_embed.{i}.layers.{j}.{part}"] = inner
                if not drop_class_embed:
                    w, b = model[f"class_embed.{i}.weight"], model[f"class_embed.{i}.bias"]
                    if background is None:
                        background = fresh_row(w.shape[1], w)
                    fw, fb = background
                    model[f"obj_class_embed.{i}.weight"] = torch.cat((w.clone(), fw), 0)[ids]
                    model[f"obj_class_embed.{i}.bias"] = torch.cat((b.clone(), fb), 0)[ids]
            model["verb_tgt_embed.weight"] = model["tgt_embed.weight"]
    if dataset == "vcoco":
        for i in range(num_layers):
            kw, kb = f"obj_class_embed.{i}.weight", f"obj_class_embed.{i}.bias"
            fw, fb = fresh_row(model[kw].shape[1], model[kw])
            model[kw] = torch.cat((model[kw][:-1], fw, model[kw][[-1]]))
            model[kb] = torch.cat((model[kb][:-1], fb, model[kb][[-1]]))
    out = {k: v for k, v in checkpoint.items() if k not in ("model", "state_dict")} if isinstance(checkpoint, dict) and (
        "model" in checkpoint or "state_dict" in checkpoint) else {}
    out["model"] = model
    return out
