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


class _CxcywhToXyxy(torch.autograd.Function):
    """(cx, cy, w, h) -> (cx - w/2, cy - h/2, cx + w/2, cy + h/2) on the two coordinate pairs at once: 4 launches forward and
    4 backward.  As unbind + four scaled differences + stack it is 9 launches forward and ~11 backward on tensors of a few
    hundred boxes, eight times per train step.  Same values bit for bit (the same subtraction / addition per coordinate; the
    backward's 0.5 * (g1 - g0) equals -0.5 g0 + 0.5 g1: scaling by a power of two is exact)."""

    @staticmethod
    def forward(ctx, x):
        half = 0.5 * x[..., 2:]
        return torch.cat((x[..., :2] - half, x[..., :2] + half), -1)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        lo, hi = g[..., :2], g[..., 2:]
        return torch.cat((lo + hi, 0.5 * (hi - lo)), -1)


def box_cxcywh_to_xyxy(x):
    return _CxcywhToXyxy.apply(x)


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


def paired_giou(a, b):
    """GIoU of xyxy boxes a[i] with b[i] ([N,4] x [N,4] -> [N]): the diagonal of generalized_box_iou"""
    # (explicit products: prod()'s backward inspects the data for zeros on the host -- a sync, and not
    #  graph-capturable.  Coordinates through ONE unbind per box set: every `x[:, k]` / `x[:, :2]` on a tensor that
    #  requires grad is a node whose backward is a zero-fill + a copy + an accumulation -- ~30 launch-bound kernels per
    #  call for the predicted boxes; unbind's backward is one stack.  Same operations per element, same order.)
    ax0, ay0, ax1, ay1 = a.unbind(-1)
    bx0, by0, bx1, by1 = b.unbind(-1)
    iw = (torch.min(ax1, bx1) - torch.max(ax0, bx0)).clamp(min=0)
    ih = (torch.min(ay1, by1) - torch.max(ay0, by0)).clamp(min=0)
    inter = iw * ih
    union = (ax1 - ax0) * (ay1 - ay0) + (bx1 - bx0) * (by1 - by0) - inter
    hw = (torch.max(ax1, bx1) - torch.min(ax0, bx0)).clamp(min=0)
    hh = (torch.max(ay1, by1) - torch.min(ay0, by0)).clamp(min=0)
    hull = hw * hh
    return inter / union - (hull - union) / hull


class LossDict(dict):
    """loss dict of SetCriterionHOI.forward; `total` carries the weighted sum (see weighted_sum)."""
    total = None


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
        # (subject and object boxes through ONE pairwise GIoU: the two diagonal blocks of the [2Q, 2T] matrix are the two cost
        #  matrices; the off-diagonal blocks are wasted arithmetic on a launch-bound call)
        Q, T = sub_box.shape[0], t_sub_box.shape[0]
        giou_all = generalized_box_iou(box_cxcywh_to_xyxy(torch.cat((sub_box, obj_box))),
                                       box_cxcywh_to_xyxy(torch.cat((t_sub_box, t_obj_box))))
        c_sub_giou = -giou_all[:Q, :T]
        c_obj_giou = -giou_all[Q:, T:] + c_sub_giou * (t_obj_box == 0).all(dim=1).unsqueeze(0)
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


native_assignment = True          # (attribute: tests compare with scipy by flipping it)


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

    def forward_per_layer(self, outputs, targets):
        """The reference's control flow: one pass of every loss per decoder layer (kept as the readable
        statement of the criterion and as a cross-check of `forward`, tests/test_modules_cpu.py)."""
        main = {k: v for k, v in outputs.items() if k != 'aux_outputs'}
        layers = [main] + list(outputs.get('aux_outputs', []))
        bs, nq = main['pred_obj_logits'].shape[:2]
        sizes = [len(t['obj_labels']) for t in targets]
        costs = [self.matcher.costs(o, targets) for o in layers]
        stacked = torch.stack([c for c, _ in costs]).float().cpu()
        assignments = [self.matcher.assign(stacked[i], bs, nq, sizes) for i in range(len(layers))]
        num_interactions = self._num_interactions(sizes, main['pred_obj_logits'].device)
        losses = {}
        for li, (o, idx, (_, c_giou)) in enumerate(zip(layers, assignments, costs)):
            d = {}
            d.update(self.loss_obj_labels(o, targets, idx, num_interactions, log=(li == 0)))
            d.update(self.loss_verb_labels(o, targets, idx, num_interactions, cost_giou=c_giou))
            d.update(self.loss_sub_obj_boxes(o, targets, idx, num_interactions))
            d.update(self.loss_obj_cardinality(o, targets, idx, num_interactions))
            losses.update(d if li == 0 else {f'{k}_{li - 1}': v for k, v in d.items()})
        return losses

    @staticmethod
    def _num_interactions(sizes, device, local=False):
        """`local`: this rank's count only, no collective (warm-up steps of a graph capture, which one rank may run
        while the others are in a training step: a capture must not issue collectives of its own)"""
        num = torch.as_tensor([sum(sizes)], dtype=torch.float, device=device)
        world = 1
        if not local and dist.is_available() and dist.is_initialized():
            dist.all_reduce(num)                               # keeps the loss scale of the reference
            world = dist.get_world_size()
        return torch.clamp(num / world, min=1)[0]              # stays on the device: no .item() sync

    def forward(self, outputs, targets):
        """All decoder layers in one pass.  The K layers' predictions are stacked into [K*bs*nq, .] rows,
        so the cost matrix, the label / verb / box losses and the cardinality error are each computed once
        on a K-times larger tensor instead of K times (the per-layer loop is ~330 launch-bound kernels
        forward and as many backward; this is ~110), with ONE device->host copy (costs) and ONE
        host->device copy (matched indices) per step.  Every entry equals `forward_per_layer`.

        The three stages are separate methods so that a HIP-graphed train step can put `prepare` at the end
        of its forward graph and `losses` at the head of its backward graph, with only the host-side
        assignment in between (train.GraphedStep)."""
        state = self.prepare(outputs, targets)
        index = self.assign(state).to(state['dev'], non_blocking=True)
        return self.losses(state, index, self._num_interactions(state['sizes'], state['dev']))

    def prepare(self, outputs, targets):
        """Device-side part before the assignment: stacked float32 predictions, concatenated targets, the
        cost matrix of every (layer, image, query) row against every target, the GIoU costs."""
        main = {k: v for k, v in outputs.items() if k != 'aux_outputs'}
        layers = [main] + list(outputs.get('aux_outputs', []))
        K = len(layers)
        bs, nq = main['pred_obj_logits'].shape[:2]

        def rows(key):                                         # [K*bs, nq, .], float32 (bf16 predictions upcast here)
            return (torch.cat([o[key] for o in layers], 0) if K > 1 else main[key]).float()
        pred = {k: rows(k) for k in ('pred_obj_logits', 'pred_verb_logits', 'pred_sub_boxes', 'pred_obj_boxes')}
        if self.subject_class:
            pred['pred_sub_logits'] = rows('pred_sub_logits')
        C, c_giou = self.matcher.costs(pred, targets)           # [K*bs*nq, T]
        st = {'K': K, 'bs': bs, 'nq': nq, 'dev': main['pred_obj_logits'].device,
              'sizes': [len(t['obj_labels']) for t in targets], 'pred': pred, 'C': C, 'c_giou': c_giou,
              't_obj': torch.cat([t['obj_labels'] for t in targets]),
              't_verb': torch.cat([t['verb_labels'] for t in targets]),
              't_sub_box': torch.cat([t['sub_boxes'] for t in targets]),
              't_obj_box': torch.cat([t['obj_boxes'] for t in targets])}
        if self.subject_class:
            st['t_sub'] = torch.cat([t['sub_labels'] for t in targets])
        if self.giou_verb_label and self.pseudo_verb:
            st['sims'] = [o['target_verb_sim'] for o in layers]
        return st

    @staticmethod
    def matched_pairs(sizes, nq, K):
        """number of (query, target) pairs the assignment yields: min(nq, targets) per image and layer"""
        return K * sum(min(nq, n) for n in sizes)

    def assign(self, state, C_host=None):
        """Host side: the cost matrix comes over in ONE copy (`C_host`: the caller's host copy, e.g. a pinned staging
        buffer; else copied here), the K*bs assignment problems are solved in one native call (`hoi_assign_batch`,
        include/rlipv2_matcher.h: scipy's algorithm and tie-breaking without Python in the loop; `criterion.native_assignment = False` or a
        non-finite cost goes through scipy itself); returns int64 [2, K*n]: rows of the stacked predictions / columns of
        the concatenated targets."""
        K, bs, nq, sizes = state['K'], state['bs'], state['nq'], state['sizes']
        if C_host is None:
            C_host = state['C'].float().cpu()
        C_host = C_host.view(K, bs, nq, sum(sizes))
        if native_assignment and C_host.dtype == torch.float32 and C_host.is_contiguous():
            from . import _lib
            import ctypes
            n = self.matched_pairs(sizes, nq, K)
            out = torch.empty(2, n, dtype=torch.int64)
            if n == 0:
                return out
            got = _lib.lib().hoi_assign_batch(C_host.data_ptr(), K, bs, nq, (ctypes.c_int * bs)(*sizes),
                                              out[0].data_ptr(), out[1].data_ptr(), n)
            if got == n:
                return out
            if got != -1:
                raise RuntimeError(f"hoi_assign_batch returned {got} for {n} expected pairs")
            # (NaN / -inf / infeasible: let scipy raise the reference's error)
        starts = [0]
        for n in sizes:
            starts.append(starts[-1] + n)
        flat_q, tgt_i = [], []
        for k in range(K):
            for i in range(bs):
                r, c = linear_sum_assignment(C_host[k, i, :, starts[i]:starts[i + 1]])
                flat_q.append(torch.as_tensor(r, dtype=torch.int64) + (k * bs + i) * nq)
                tgt_i.append(torch.as_tensor(c, dtype=torch.int64) + starts[i])
        return torch.stack([torch.cat(flat_q), torch.cat(tgt_i)])

    def _constants(self, K, sizes, dev, dtype, kinds):
        """small host-built tensors, cached (creating them would be a host->device copy per step, which is
        also illegal during HIP-graph capture)"""
        key = (K, tuple(sizes), str(dev), dtype, tuple(kinds))
        cache = self.__dict__.setdefault('_const_cache', {})
        if key not in cache:
            w = torch.tensor([[self.weight_dict.get(k + ('' if li == 0 else f'_{li - 1}'), 0.0) for li in range(K)]
                              for k in kinds], device=dev, dtype=dtype)
            cache[key] = (torch.as_tensor(sizes, device=dev, dtype=torch.float), w)
        return cache[key]

    def _ce_weight(self, n_cls, dev, dtype):
        """class weights of the label cross-entropy (eos_coef on the "no object" class), cached: writing a
        Python scalar into a device tensor is a host->device copy"""
        key = ('ce', n_cls, str(dev), dtype, self.eos_coef)
        cache = self.__dict__.setdefault('_const_cache', {})
        if key not in cache:
            w = torch.ones(n_cls, dtype=dtype)
            w[-1] = self.eos_coef
            cache[key] = w.to(dev)
        return cache[key]

    def losses(self, state, index, num_interactions):
        """Device-side part after the assignment (no host interaction: graph-capturable)."""
        K, bs, nq, sizes, dev = state['K'], state['bs'], state['nq'], state['sizes'], state['dev']
        pred, c_giou = state['pred'], state['c_giou']
        flat_q, tgt_i = index[0], index[1]
        n = flat_q.shape[0] // K                                # matched pairs per layer (the same for every layer)
        out = {}

        # --- object / subject labels: weighted cross-entropy per layer (hoi.py:3696) ---
        def label_ce(logits, t_lab):
            n_cls = logits.shape[-1]
            logits = logits.reshape(K * bs * nq, n_cls)
            weight = self._ce_weight(n_cls, dev, logits.dtype)
            matched = t_lab[tgt_i]
            tgt = torch.full((K * bs * nq,), n_cls - 1, dtype=torch.int64, device=dev)
            tgt[flat_q] = matched
            nll = F.cross_entropy(logits, tgt, weight, reduction='none').view(K, -1).sum(1)
            return nll / weight[tgt].view(K, -1).sum(1), logits, matched
        ce, o_logits, m_o = label_ce(pred['pred_obj_logits'], state['t_obj'])
        if self.subject_class:
            ce_s, s_logits, m_s = label_ce(pred['pred_sub_logits'], state['t_sub'])
            ce = ce + ce_s
        out['loss_obj_ce'] = ce

        def class_error(logits, matched):
            if n == 0:
                return torch.zeros([], device=dev)
            return 100 - (logits[flat_q[:n]].argmax(-1) == matched[:n]).float().mean() * 100
        log = {'obj_class_error': class_error(o_logits, m_o)}
        if self.subject_class:
            log['sub_class_error'] = class_error(s_logits, m_s)

        # --- verbs: (quality) focal loss on sigmoid probabilities (hoi.py:3925) ---
        v_logits = pred['pred_verb_logits']
        if self.use_no_verb_token:
            v_logits = v_logits[:, :, :-1]
        nv = v_logits.shape[-1]
        v_logits = v_logits.reshape(K * bs * nq, nv)
        lab = state['t_verb'][tgt_i].to(v_logits.dtype)
        if self.giou_verb_label:
            quality = (1 - c_giou[flat_q, tgt_i]) / 2           # matcher cost is -GIoU; GIoU -> [0, 1]
            if self.pseudo_verb:
                sims = state['sims']
                if all(s_ is sims[0] for s_ in sims):
                    lab = lab + sims[0].index_select(0, tgt_i).float()
                else:
                    lab = lab + torch.stack(sims)[torch.arange(K, device=dev).repeat_interleave(n), tgt_i].float()
            lab = lab * quality.unsqueeze(-1)
        gt = torch.zeros_like(v_logits)
        gt[flat_q] = lab.to(v_logits.dtype)
        p = torch.clamp(v_logits.sigmoid(), 1e-6, 1. - 1e-6)
        if self.giou_verb_label:                                # soft_neg_loss, per layer
            el = torch.pow(torch.abs(gt - p), 2) * ((1 - gt) * torch.log(1 - p) + gt * torch.log(p))
            tot, num_pos = el.view(K, -1).sum(1), gt.gt(0).view(K, -1).sum(1).float()
            out['loss_verb_ce'] = torch.where(num_pos == 0, -tot, -tot / num_pos.clamp(min=1))
        else:                                                   # neg_loss, per layer
            pos, neg = gt.eq(1).float(), gt.lt(1).float()
            pl = (torch.log(p) * torch.pow(1 - p, 2) * pos).view(K, -1).sum(1)
            nl = (torch.log(1 - p) * torch.pow(p, 2) * torch.pow(1 - gt, 4) * neg).view(K, -1).sum(1)
            num_pos = pos.view(K, -1).sum(1)
            out['loss_verb_ce'] = torch.where(num_pos == 0, -nl, -(pl + nl) / num_pos.clamp(min=1))

        # --- boxes: L1 + GIoU of matched pairs (hoi.py:4162) ---
        # (index_select: its backward is index_add_, which is graph-capturable; advanced indexing's backward
        #  is a sort-based index_put_ that is not)
        src_s = pred['pred_sub_boxes'].reshape(-1, 4).index_select(0, flat_q)
        src_o = pred['pred_obj_boxes'].reshape(-1, 4).index_select(0, flat_q)
        if n == 0:
            z_s, z_o = src_s.sum().expand(K), src_o.sum().expand(K)
            out.update(loss_sub_bbox=z_s, loss_obj_bbox=z_o, loss_sub_giou=z_s, loss_obj_giou=z_o)
        else:
            tgt_s, tgt_o = state['t_sub_box'][tgt_i], state['t_obj_box'][tgt_i]
            exist = (tgt_o != 0).any(dim=1)
            n_exist = exist.view(K, -1).sum(1) + 1e-4
            out['loss_sub_bbox'] = (src_s - tgt_s).abs().view(K, -1).sum(1) / num_interactions
            out['loss_obj_bbox'] = ((src_o - tgt_o).abs() * exist.unsqueeze(1)).view(K, -1).sum(1) / n_exist
            # (subject and object pairs through ONE GIoU evaluation: the ~30 launch-bound kernels of a call -- and twice that
            #  in its backward -- once instead of twice; per-pair arithmetic unchanged)
            g_both = 1 - paired_giou(box_cxcywh_to_xyxy(torch.cat((src_s, src_o))),
                                     box_cxcywh_to_xyxy(torch.cat((tgt_s, tgt_o))))
            g_s, g_o = g_both.split(src_s.shape[0])
            out['loss_sub_giou'] = g_s.view(K, -1).sum(1) / num_interactions
            out['loss_obj_giou'] = (g_o * exist).view(K, -1).sum(1) / n_exist

        kinds = [k for k in out if k in self.weight_dict]
        lengths, w = self._constants(K, sizes, dev, out[kinds[0]].dtype, kinds)

        # --- cardinality error (hoi.py:3909), logging only ---
        card = (o_logits.argmax(-1) != o_logits.shape[-1] - 1).view(K, bs, nq).sum(2).float()
        out['obj_cardinality_error'] = (card - lengths).abs().mean(1)

        losses = LossDict()
        for li in range(K):
            sfx = '' if li == 0 else f'_{li - 1}'
            for k, v in out.items():
                losses[k + sfx] = v[li]
            if li == 0:
                losses.update(log)
        # the weighted total as one dot product instead of a chain of ~3*K tiny multiply-adds
        losses.total = (torch.stack([out[k] for k in kinds]) * w).sum()
        return losses

    def weighted_sum(self, loss_dict):
        total = getattr(loss_dict, 'total', None)
        if total is not None:
            return total
        return sum(loss_dict[k] * self.weight_dict[k] for k in loss_dict if k in self.weight_dict)
