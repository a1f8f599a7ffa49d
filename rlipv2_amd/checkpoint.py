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
