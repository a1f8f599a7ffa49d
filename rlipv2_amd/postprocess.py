"""PostProcessHOI: model outputs -> per-image HOI detections (reference
models/hoi.py:4769-4873): object class = argmax of softmax over the real classes (optionally with
temperature 0.07), verb scores = sigmoid(verb logits) x object score, boxes cxcywh -> xyxy in pixels, the
subject label fixed to `subject_category_id`, and with `zero_shot_hoi_eval` only the queries whose predicted
SUBJECT class is that category are kept.

Same result dicts as the reference ('labels', 'boxes', 'verb_scores', 'sub_ids', 'obj_ids', CPU tensors),
but everything is computed for the whole batch on the device and moved to the host in ONE transfer (the
reference does three `.to('cpu')` round trips per image).  The reference's constructor also loads an
object/verb co-occurrence prior that its forward never uses (:4862 is commented out); it is omitted."""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import nn

from .criterion import box_cxcywh_to_xyxy


class PostProcessHOI(nn.Module):
    def __init__(self, subject_category_id, sigmoid=True, temperature=False, zero_shot_hoi_eval=False,
                 verb_curing=False):
        super().__init__()
        if verb_curing and not sigmoid:
            raise AssertionError("verb_curing needs sigmoid verb scores")
        self.subject_category_id = subject_category_id
        self.sigmoid, self.temperature, self.tao = sigmoid, temperature, 0.07
        self.zero_shot_hoi_eval, self.verb_curing = zero_shot_hoi_eval, verb_curing

    @torch.no_grad()
    def forward(self, outputs, target_sizes):
        obj_logits, verb_logits = outputs['pred_obj_logits'].float(), outputs['pred_verb_logits'].float()
        sub_boxes, obj_boxes = outputs['pred_sub_boxes'].float(), outputs['pred_obj_boxes'].float()
        assert len(obj_logits) == len(target_sizes)
        assert target_sizes.shape[1] == 2  # h, w
        bs, nq = obj_logits.shape[:2]
        t = self.tao if self.temperature else 1.0
        obj_scores, obj_labels = F.softmax(obj_logits / t, -1)[..., :-1].max(-1)
        verb_scores = verb_logits.sigmoid() if self.sigmoid else verb_logits
        if self.sigmoid and self.verb_curing:
            verb_scores = verb_scores * outputs['curing_score']
        verb_scores = verb_scores * obj_scores.unsqueeze(-1)
        img_h, img_w = target_sizes.to(verb_scores.device).unbind(1)
        scale = torch.stack([img_w, img_h, img_w, img_h], dim=1)[:, None, :].to(verb_scores.dtype)
        boxes = torch.cat((box_cxcywh_to_xyxy(sub_boxes) * scale, box_cxcywh_to_xyxy(obj_boxes) * scale), 1)
        keep = None
        if self.zero_shot_hoi_eval:
            assert 'pred_sub_logits' in outputs
            sub_labels = F.softmax(outputs['pred_sub_logits'].float() / t, -1)[..., :-1].argmax(-1)
            keep = (sub_labels == self.subject_category_id)
        # one device->host transfer for the whole batch
        packed = torch.cat([obj_labels.to(verb_scores.dtype).unsqueeze(-1), verb_scores,
                            boxes[:, :nq], boxes[:, nq:],
                            (keep if keep is not None else torch.ones_like(obj_labels, dtype=torch.bool))
                            .to(verb_scores.dtype).unsqueeze(-1)], -1).cpu()
        nv = verb_scores.shape[-1]
        results = []
        for b in range(bs):
            row = packed[b]
            sel = row[:, -1] > 0.5
            row = row[sel]
            ol = row[:, 0].to(torch.int64)
            n = ol.shape[0]
            ids = torch.arange(2 * n)
            results.append({'labels': torch.cat((torch.full_like(ol, self.subject_category_id), ol)),
                            'boxes': torch.cat((row[:, 1 + nv:5 + nv], row[:, 5 + nv:9 + nv])),
                            'verb_scores': row[:, 1:1 + nv], 'sub_ids': ids[:n], 'obj_ids': ids[n:]})
        return results
