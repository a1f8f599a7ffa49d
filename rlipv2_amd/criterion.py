"""Set criterion of the RLIPv2-ParSeDA train step for the configuration of
scripts/RLIP_ParSeDA/train_RLIP_ParSeDA_v2_mixed_vgcoco_resnet.sh: subject classes, cross-entropy
object/subject labels (eos_coef on the last class), quality-focal verb labels with GIoU soft
targets (--giou_verb_label) and pseudo-verb similarity (--pseudo_verb), L1 + GIoU boxes,
cardinality error.

Reference: HungarianMatcherHOI (models/matcher.py:95-270), SetCriterionHOI (models/hoi.py:3627-4766:
loss_obj_labels :3696, loss_obj_cardinality :3909, loss_verb_labels :3925, loss_sub_obj_boxes :4162,
_soft_neg_loss :4481, forward :4654), weight_dict (models/detr.py:571-620), box ops util/box_ops.py.

Host-sync discipline: the reference runs the matcher twice per decoder layer (once for the
assignment, once more inside loss_verb_labels for the GIoU costs) and calls .item() on the
interaction count -- 6 + 1 device->host round trips per step.  Here the cost matrices of all decoder
layers are built on the GPU, copied to the host in ONE transfer, assigned with scipy there, and
the GIoU costs are reused; results are identical.
"""
from __future__ import annotations

import torch
import torch.distributed as dist
import torch.nn.functional as F
from scipy.optimize import linear_sum_assignment
from torch import nn


def box_cxcywh_to_xyxy(x):
    cx, cy, w, h = x.unbind(-1)
    return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)


def _area(b):
    return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])


def generalized_box_iou(a, b):
    """pairwise GIoU of xyxy boxes [N,4] x [M,4] -> [N,M]"""
    area_a, area_b = _area(a), _area(b)
    lt = torch.max(a[:, None, :2], b[:, :2])
    rb = torch.min(a[:, None, 2:], b[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    union = area_a[:, None] + area_b - inter
    iou = inter / union
    lt = torch.min(a[:, None, :2], b[:, :2])
    rb = torch.max(a[:, None, 2:], b[:, 2:])
    wh = (rb - lt).clamp(min=0)
    hull = wh[..., 0] * wh[..., 1]
    return iou - (hull - union) / hull


class HungarianMatcherHOI(nn.Module):
    def __init__(self, cost_obj_class=1.0, cost_verb_class=1.0, cost_bbox=1.0, cost_giou=1.0, subject_class=False):
        super().__init__()
        assert cost_obj_class != 0 or cost_verb_class != 0 or cost_bbox != 0 or cost_giou != 0, 'all costs cant be 0'
        self.cost_obj_class, self.cost_verb_class = cost_obj_class, cost_verb_class
        self.cost_bbox, self.cost_giou = cost_bbox, cost_giou
        self.subject_class = subject_class

    @torch.no_grad()
    def costs(self, outputs, targets):
        """(C [bs*nq, T], cost_giou [bs*nq, T]) on the device of the predictions."""
        obj_prob = outputs['pred_obj_logits'].flatten(0, 1).softmax(-1)
        verb_prob = outputs['pred_verb_logits'].flatten(0, 1).sigmoid()
        sub_box = outputs['pred_sub_boxes'].flatten(0, 1)
        obj_box = outputs['pred_obj_boxes'].flatten(0, 1)
        t_obj = torch.cat([v['obj_labels'] for v in targets])
        t_verb = torch.cat([v['verb_labels'] for v in targets]).to(verb_prob.dtype)        # [T, n_verb]
        t_sub_box = torch.cat([v['sub_boxes'] for v in targets])
        t_obj_box = torch.cat([v['obj_boxes'] for v in targets])
        if verb_prob.shape[1] - 1 == t_verb.shape[1]:                                    # "no verb" token column
            verb_prob = verb_prob[:, :-1]
        tv = t_verb.t()                                                                   # [n_verb, T]
        c_verb = -(verb_prob.matmul(tv) / (tv.sum(dim=0, keepdim=True) + 1e-4)
                   + (1 - verb_prob).matmul(1 - tv) / ((1 - tv).sum(dim=0, keepdim=True) + 1e-4)) / 2
        c_obj = -obj_prob[:, t_obj]
        c_sub_box = torch.cdist(sub_box, t_sub_box, p=1)
        c_obj_box = torch.cdist(obj_box, t_obj_box, p=1) * (t_obj_box != 0).any(dim=1).unsqueeze(0)
        c_box = c_sub_box if c_sub_box.shape[1] == 0 else torch.max(c_sub_box, c_obj_box)
        c_sub_giou = -generalized_box_iou(box_cxcywh_to_xyxy(sub_box), box_cxcywh_to_xyxy(t_sub_box))
        c_obj_giou = -generalized_box_iou(box_cxcywh_to_xyxy(obj_box), box_cxcywh_to_xyxy(t_obj_box)) \
            + c_sub_giou * (t_obj_box == 0).all(dim=1).unsqueeze(0)
        c_giou = c_sub_giou if c_sub_giou.shape[1] == 0 else torch.max(c_sub_giou, c_obj_giou)
        C = self.cost_obj_class * c_obj + self.cost_verb_class * c_verb + self.cost_bbox * c_box \
            + self.cost_giou * c_giou
        if self.subject_class:
            sub_prob = outputs['pred_sub_logits'].flatten(0, 1).softmax(-1)
            t_sub = torch.cat([v['sub_labels'] for v in targets])
            C = C + self.cost_obj_class * (-sub_prob[:, t_sub])
        return C, c_giou

    @staticmethod
    def assign(C_host, bs, nq, sizes):
        C_host = C_host.view(bs, nq, -1)
        out = []
        for i, c in enumerate(C_host.split(sizes, -1)):
            r, col = linear_sum_assignment(c[i])
            out.append((torch.as_tensor(r, dtype=torch.int64), torch.as_tensor(col, dtype=torch.int64)))
        return out

    @torch.no_grad()
    def forward(self, outputs, targets, return_cost=False):
        bs, nq = outputs['pred_obj_logits'].shape[:2]
        C, c_giou = self.costs(outputs, targets)
        idx = self.assign(C.float().cpu(), bs, nq, [len(v['obj_labels']) for v in targets])
        return (idx, c_giou) if return_cost else idx


def soft_neg_loss(pred, gt, eps=1e-6, beta=2):
    """Quality focal loss on probabilities (reference hoi.py:4481-4495)."""
    pred = torch.clamp(pred, eps, 1. - eps)
    loss = torch.pow(torch.abs(gt - pred), beta) * ((1 - gt) * torch.log(1 - pred) + gt * torch.log(pred))
    num_pos = gt.gt(0).float().sum()
    return torch.where(num_pos == 0, -loss.sum(), -loss.sum() / num_pos.clamp(min=1))


def neg_loss(pred, gt, eps=1e-6):
    """CornerNet focal loss on probabilities (reference hoi.py:4455-4479)."""
    pos, neg = gt.eq(1).float(), gt.lt(1).float()
    pred = torch.clamp(pred, eps, 1. - eps)
    pos_loss = (torch.log(pred) * torch.pow(1 - pred, 2) * pos).sum()
    neg_loss_ = (torch.log(1 - pred) * torch.pow(pred, 2) * torch.pow(1 - gt, 4) * neg).sum()
    num_pos = pos.sum()
    return torch.where(num_pos == 0, -neg_loss_, -(pos_loss + neg_loss_) / num_pos.clamp(min=1))


def build_weight_dict(dec_layers, obj_loss_coef=1.0, verb_loss_coef=1.0, bbox_loss_coef=2.5, giou_loss_coef=1.0,
                      aux_loss=True):
    w = {'loss_obj_ce': obj_loss_coef, 'loss_verb_ce': verb_loss_coef, 'loss_sub_bbox': bbox_loss_coef,
         'loss_obj_bbox': bbox_loss_coef, 'loss_sub_giou': giou_loss_coef, 'loss_obj_giou': giou_loss_coef}
    if aux_loss:
        for i in range(dec_layers - 1):
            w.update({f'{k}_{i}': v for k, v in list(w.items()) if not k[-1].isdigit()})
    return w


class SetCriterionHOI(nn.Module):
    def __init__(self, matcher, weight_dict, eos_coef=0.1, subject_class=True, giou_verb_label=True,
                 pseudo_verb=True, use_no_verb_token=False):
        super().__init__()
        self.matcher, self.weight_dict, self.eos_coef = matcher, weight_dict, eos_coef
        self.subject_class, self.giou_verb_label, self.pseudo_verb = subject_class, giou_verb_label, pseudo_verb
        self.use_no_verb_token = use_no_verb_token

    @staticmethod
    def _src_idx(indices):
        batch = torch.cat([torch.full_like(src, i) for i, (src, _) in enumerate(indices)])
        return batch, torch.cat([src for (src, _) in indices])

    def _label_ce(self, logits, targets, indices, key, idx):
        n_cls = logits.shape[-1]
        weight = torch.ones(n_cls, device=logits.device, dtype=logits.dtype)
        weight[-1] = self.eos_coef
        matched = torch.cat([t[key][J.to(t[key].device)] for t, (_, J) in zip(targets, indices)])
        tgt = torch.full(logits.shape[:2], n_cls - 1, dtype=torch.int64, device=logits.device)
        tgt[idx] = matched
        return F.cross_entropy(logits.transpose(1, 2), tgt, weight), matched

    def loss_obj_labels(self, outputs, targets, indices, num_interactions, log=True):
        idx = self._src_idx(indices)
        ce_o, m_o = self._label_ce(outputs['pred_obj_logits'], targets, indices, 'obj_labels', idx)
        losses = {'loss_obj_ce': ce_o}
        if self.subject_class:
            ce_s, m_s = self._label_ce(outputs['pred_sub_logits'], targets, indices, 'sub_labels', idx)
            losses['loss_obj_ce'] = ce_o + ce_s
        if log:
            def err(logits, tgt):
                if tgt.numel() == 0:
                    return torch.zeros([], device=logits.device)
                return 100 - (logits.argmax(-1) == tgt).float().mean() * 100
            losses['obj_class_error'] = err(outputs['pred_obj_logits'][idx], m_o)
            if self.subject_class:
                losses['sub_class_error'] = err(outputs['pred_sub_logits'][idx], m_s)
        return losses

    def loss_obj_cardinality(self, outputs, targets, indices, num_interactions):
        logits = outputs['pred_obj_logits']
        lengths = torch.as_tensor([len(v['obj_labels']) for v in targets], device=logits.device)
        card = (logits.argmax(-1) != logits.shape[-1] - 1).sum(1)
        return {'obj_cardinality_error': F.l1_loss(card.float(), lengths.float())}

    def loss_verb_labels(self, outputs, targets, indices, num_interactions, cost_giou=None):
        logits = outputs['pred_verb_logits']
        idx = self._src_idx(indices)
        nq = logits.shape[1]
        if self.giou_verb_label:
            giou = -cost_giou                                   # the matcher's GIoU cost is negated GIoU
            soft, q0, t0 = [], 0, 0
            for t, (I, J) in zip(targets, indices):
                dev = giou.device
                s = (giou[q0 + I.to(dev), t0 + J.to(dev)] + 1) / 2                     # GIoU -> [0, 1]
                lab = t['verb_labels'][J.to(t['verb_labels'].device)]
                if self.pseudo_verb:
                    lab = lab + outputs['target_verb_sim'][t0 + J.to(dev)]
                soft.append(lab * s.unsqueeze(-1))
                q0 += nq
                t0 += J.shape[0]
            matched = torch.cat(soft)
        else:
            matched = torch.cat([t['verb_labels'][J.to(t['verb_labels'].device)] for t, (_, J) in zip(targets, indices)])
        if self.use_no_verb_token:
            logits = logits[:, :, :-1]
        tgt = torch.zeros_like(logits)
        tgt[idx] = matched.to(logits.dtype)
        prob = logits.sigmoid()
        return {'loss_verb_ce': soft_neg_loss(prob, tgt) if self.giou_verb_label else neg_loss(prob, tgt)}

    def loss_sub_obj_boxes(self, outputs, targets, indices, num_interactions):
        idx = self._src_idx(indices)
        src_s, src_o = outputs['pred_sub_boxes'][idx], outputs['pred_obj_boxes'][idx]
        tgt_s = torch.cat([t['sub_boxes'][i.to(t['sub_boxes'].device)] for t, (_, i) in zip(targets, indices)], dim=0)
        tgt_o = torch.cat([t['obj_boxes'][i.to(t['obj_boxes'].device)] for t, (_, i) in zip(targets, indices)], dim=0)
        if src_s.shape[0] == 0:
            return {'loss_sub_bbox': src_s.sum(), 'loss_obj_bbox': src_o.sum(), 'loss_sub_giou': src_s.sum(),
                    'loss_obj_giou': src_o.sum()}
        exist = (tgt_o != 0).any(dim=1)
        l1_s = F.l1_loss(src_s, tgt_s, reduction='none')
        l1_o = F.l1_loss(src_o, tgt_o, reduction='none')
        g_s = 1 - torch.diag(generalized_box_iou(box_cxcywh_to_xyxy(src_s), box_cxcywh_to_xyxy(tgt_s)))
        g_o = 1 - torch.diag(generalized_box_iou(box_cxcywh_to_xyxy(src_o), box_cxcywh_to_xyxy(tgt_o)))
        return {'loss_sub_bbox': l1_s.sum() / num_interactions,
                'loss_obj_bbox': (l1_o * exist.unsqueeze(1)).sum() / (exist.sum() + 1e-4),
                'loss_sub_giou': g_s.sum() / num_interactions,
                'loss_obj_giou': (g_o * exist).sum() / (exist.sum() + 1e-4)}

    def forward(self, outputs, targets):
        main = {k: v for k, v in outputs.items() if k != 'aux_outputs'}
        layers = [main] + list(outputs.get('aux_outputs', []))
        bs, nq = main['pred_obj_logits'].shape[:2]
        sizes = [len(t['obj_labels']) for t in targets]
        # all layers' cost matrices -> ONE device->host copy -> scipy on the host
        costs = [self.matcher.costs(o, targets) for o in layers]
        stacked = torch.stack([c for c, _ in costs]).float().cpu()
        assignments = [self.matcher.assign(stacked[i], bs, nq, sizes) for i in range(len(layers))]

        num = torch.as_tensor([sum(sizes)], dtype=torch.float, device=main['pred_obj_logits'].device)
        world = 1
        if dist.is_available() and dist.is_initialized():
            dist.all_reduce(num)                               # keeps the loss scale of the reference
            world = dist.get_world_size()
        num_interactions = torch.clamp(num / world, min=1)[0]  # stays on the device: no .item() sync

        losses = {}
        for li, (o, idx, (_, c_giou)) in enumerate(zip(layers, assignments, costs)):
            d = {}
            d.update(self.loss_obj_labels(o, targets, idx, num_interactions, log=(li == 0)))
            d.update(self.loss_verb_labels(o, targets, idx, num_interactions, cost_giou=c_giou))
            d.update(self.loss_sub_obj_boxes(o, targets, idx, num_interactions))
            d.update(self.loss_obj_cardinality(o, targets, idx, num_interactions))
            losses.update(d if li == 0 else {f'{k}_{li - 1}': v for k, v in d.items()})
        return losses

    def weighted_sum(self, loss_dict):
        return sum(loss_dict[k] * self.weight_dict[k] for k in loss_dict if k in self.weight_dict)
