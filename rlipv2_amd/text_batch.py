"""Per-batch label-text merging: every image brings its own object / verb
name lists and labels indexing into them; a train step encodes ONE merged text list per batch and re-indexes
all labels into it, then pads the list with sampled negatives to a fixed length so that the text tensors keep
a static shape.

Reference: engine.merge_batch_data (engine.py:700-757), merge_obj_text (:759-782), merge_verb_text (:784-821),
sample_text (:823-937); call site engine.py:93-98.  Same results for the same inputs and the same `random`
state (the sampling draws `random.choice` / `random.uniform` in the reference's order); re-built around
dictionaries and tensor indexing instead of repeated `list.index` scans and per-label Python loops:

  * first-occurrence order de-duplication with a dict: O(total names) instead of O(names^2);
  * object labels: one gather through a per-image remap tensor;
  * verb labels: one scatter of the image's multi-hot columns into the merged width;
  * frequency sampling: `bisect` over the cumulative distribution instead of a linear scan per draw;
  * hard-negative mining: one matrix product for the cosine similarities, one sort.
"""
from __future__ import annotations

import bisect
import random
from itertools import accumulate

import torch
import torch.nn.functional as F


def _merge_names(name_lists):
    merged, index = [], {}
    for names in name_lists:
        for t in names:
            if t not in index:
                index[t] = len(merged)
                merged.append(t)
    return merged, index


def merge_obj_text(text_list, label_list):
    """-> (merged names, per-image int64 labels into the merged list)"""
    merged, index = _merge_names(text_list)
    new_labels = []
    for names, labels in zip(text_list, label_list):
        remap = torch.tensor([index[t] for t in names], dtype=torch.int64, device=labels.device)
        new_labels.append(remap[labels.to(torch.int64)] if len(names) else labels.to(torch.int64))
    return merged, new_labels


def merge_verb_text(text_list, label_list):
    """-> (merged names, per-image float32 multi-hot [n_triplets, len(merged)])"""
    merged, index = _merge_names(text_list)
    new_labels = []
    for names, labels in zip(text_list, label_list):
        out = torch.zeros((labels.shape[0], len(merged)), dtype=torch.float32, device=labels.device)
        if labels.shape[0] and len(names):
            cols = torch.tensor([index[t] for t in names], dtype=torch.int64, device=labels.device)
            # (labels == 1) as the reference tests it; duplicate names within an image fold onto one column
            out.index_put_((torch.arange(labels.shape[0], device=labels.device)[:, None].expand_as(labels), cols[None].expand_as(labels)),
                           (labels == 1).to(torch.float32), accumulate=True)
            out.clamp_(max=1.0)
        new_labels.append(out)
    return merged, new_labels


class TextVocabulary:
    """what sample_text reads from `data_loader.dataset`: the full name lists with their frequencies, and
    (for hard mining) one feature vector per name"""

    def __init__(self, object_names, object_freq, relationship_names, relationship_freq, obj_feature=None,
                 rel_feature=None):
        self.object_names, self.object_freq = list(object_names), dict(object_freq)
        self.relationship_names, self.relationship_freq = list(relationship_names), dict(relationship_freq)
        self.obj_feature, self.rel_feature = obj_feature, rel_feature     # (names, [n, d] tensor) or None


def sample_text(merged_list, text_type, negative_text_sampling, vocab, verb_target=None, obj_target=None,
                sampling_stategy="random"):
    """pad `merged_list` with negatives up to `negative_text_sampling` names; verb targets are zero-padded to
    the new width (reference engine.py:823-937)"""
    assert text_type in ("obj", "rel")
    if sampling_stategy == "hard_mining":
        target = obj_target[0] if text_type == "obj" else verb_target
        if torch.cat(list(target), dim=0).shape[0] == 0:
            sampling_stategy = "freq"                     # no positive triplet in the batch
    if len(merged_list) >= negative_text_sampling:
        return merged_list, verb_target
    present = set(merged_list)
    if sampling_stategy in ("random", "freq"):
        full = vocab.object_names if text_type == "obj" else vocab.relationship_names
        freq = vocab.object_freq if text_type == "obj" else vocab.relationship_freq
        values = list(freq.values())
        total = sum(values)
        cumulative = [c / total for c in accumulate(values)]
        while len(merged_list) < negative_text_sampling:
            if sampling_stategy == "random":
                t = random.choice(full)
            else:
                t = full[bisect.bisect_left(cumulative, random.uniform(0, 1))]
            if t not in present:
                present.add(t)
                merged_list.append(t)
    elif sampling_stategy == "hard_mining":
        names, feats = vocab.obj_feature if text_type == "obj" else vocab.rel_feature
        where = {t: i for i, t in enumerate(names)}
        query = F.normalize(feats[torch.tensor([where[t] for t in merged_list])], p=2, dim=-1)
        sim = query @ F.normalize(feats, p=2, dim=-1).t()                        # [merged, vocabulary]
        if text_type == "obj":
            q = sim[torch.cat(list(obj_target[0]) + list(obj_target[1]), dim=0).to(sim.device)]
        else:
            q = torch.cat(list(verb_target), dim=0).to(sim) @ sim               # sum of the positives' rows
        q = (q / q.max(-1)[0].unsqueeze(-1)).sum(dim=0)
        order = torch.sort(q, dim=0, descending=True)[1].tolist()
        for i in order:
            if len(merged_list) >= negative_text_sampling:
                break
            if names[i] not in present:
                present.add(names[i])
                merged_list.append(names[i])
    else:
        raise ValueError(f"unknown sampling strategy {sampling_stategy!r}")
    assert len(merged_list) == negative_text_sampling
    if text_type == "rel":
        assert verb_target is not None
        pad = negative_text_sampling - verb_target[0].shape[-1]
        return merged_list, [F.pad(v, (0, pad)) for v in verb_target]
    return merged_list, verb_target


def merge_batch_data(kwargs, use_no_obj_token, use_all_text_labels, negative_text_sampling=0, sampling_stategy=None,
                     vocab=None):
    """kwargs = {'targets': [...], 'text': [(obj names, verb names) per image]} -> the same dict with ONE
    merged (obj names, verb names) pair and re-indexed `obj_labels` / `sub_labels` / `verb_labels`."""
    targets, text = kwargs["targets"], kwargs["text"]
    obj_text = [o for (o, _) in text]
    verb_text = [v for (_, v) in text]
    if use_all_text_labels:
        assert len({len(o) for o in obj_text}) <= 1 and len({len(v) for v in verb_text}) <= 1
    merged_obj, new_obj = merge_obj_text(obj_text, [t["obj_labels"] for t in targets])
    _, new_sub = merge_obj_text(obj_text, [t["sub_labels"] for t in targets])
    merged_verb, new_verb = merge_verb_text(verb_text, [t["verb_labels"] for t in targets])
    strategies = (sampling_stategy, sampling_stategy) if "+" not in sampling_stategy else sampling_stategy.split("+")
    n_obj = int(negative_text_sampling * 2 / 3.0)
    merged_obj, _ = sample_text(merged_obj, "obj", n_obj, vocab, obj_target=(new_sub, new_obj),
                                sampling_stategy=strategies[0])
    merged_verb, new_verb = sample_text(merged_verb, "rel", negative_text_sampling - n_obj, vocab, verb_target=new_verb,
                                        sampling_stategy=strategies[1])
    if use_no_obj_token:
        merged_obj.append("no objects")
    kwargs["text"] = [(merged_obj, merged_verb)]
    for t, o, s, v in zip(targets, new_obj, new_sub, new_verb):
        t["obj_labels"], t["sub_labels"], t["verb_labels"] = o, s, v
    return kwargs


# ---- text encoder input: de-duplication, token-length buckets, label-embedding cache ------------------------------------
TOKEN_BUCKETS = (4, 8, 12, 16, 24, 32, 48, 64)


def bucket_token_length(input_ids, attention_mask, pad_id=1, buckets=TOKEN_BUCKETS):
    """The tokenizer pads a batch of label texts to its longest member (`padding="longest"`,
    dab_deformable/deformable_transformer.py:496); a graph-captured step is specific to that width, so every new longest
    label would cost a capture.  Here the width is cut to the longest real row and padded up to the next bucket
    (4, 8, 12, 16, ... tokens): a handful of widths for the whole run, and no encoder work on all-padding columns beyond
    the bucket.  Host-side (the token tensors are built on the host): [n, T] -> [n, T_bucket]."""
    if input_ids.dim() != 2 or input_ids.shape != attention_mask.shape:
        raise ValueError("input_ids / attention_mask must be [n_text, T] of the same shape")
    longest = int(attention_mask.sum(1).max()) if input_ids.numel() else 0
    width = next((b for b in buckets if b >= longest), longest)
    T = input_ids.shape[1]
    if width <= T:
        return input_ids[:, :width].contiguous(), attention_mask[:, :width].contiguous()
    pad = width - T
    return (F.pad(input_ids, (0, pad), value=pad_id), F.pad(attention_mask, (0, pad), value=0))


def dedupe_token_rows(input_ids, attention_mask):
    """Identical label texts (the same token row) are encoded once: -> (unique ids, unique mask, inverse) with
    `unique[inverse] == rows`, first-occurrence order.  The reference encodes every string of `flat_text`
    (deformable_transformer.py:490-498), duplicates included; gathering the pooled rows through `inverse` gives the
    same tensor and, in the backward pass, sums the duplicates' gradients as autograd's index does."""
    seen, keep, inverse = {}, [], []
    rows = [tuple(r) for r in (input_ids * attention_mask + (1 - attention_mask) * -1).tolist()]
    for i, r in enumerate(rows):
        j = seen.get(r)
        if j is None:
            j = seen[r] = len(keep)
            keep.append(i)
        inverse.append(j)
    keep = torch.tensor(keep, dtype=torch.long)
    return input_ids[keep], attention_mask[keep], torch.tensor(inverse, dtype=torch.long)


class LabelEmbeddingCache:
    """Pooled text-encoder outputs per label text, for runs whose text encoder is FROZEN (`--freeze_text_encoder`,
    evaluation, zero-shot inference: hoi.py PostProcessHOI paths): the label vocabulary of a dataset is a few hundred
    strings that recur in every batch, and the 12-layer encoder is ~107 launches per call.  Keyed by the token row; rows
    not seen before are encoded together in one call (bucketed width), everything else is a gather.  With a trainable
    encoder the embeddings change every step -- `encode(..., use_cache=False)` then only de-duplicates within the call."""

    def __init__(self, max_entries=65536):
        self.rows = {}                      # token tuple -> row in self.table
        self.table = None                   # [entries, hidden] on the encoder's device
        self.max_entries = max_entries
        self.hits = self.misses = 0
        self.signature = None

    def clear(self):
        self.rows, self.table = {}, None

    @staticmethod
    def _signature(text_encoder):
        ps = list(text_encoder.parameters())
        return (id(text_encoder), str(ps[0].dtype), str(ps[0].device), ps[0].data_ptr(), sum(p._version for p in ps))

    def encode(self, text_encoder, input_ids, attention_mask, use_cache=None):
        """-> pooled embeddings [n_text, hidden] in the rows' order.  `use_cache` default: the encoder has no trainable
        parameter and is in eval mode (dropout off)."""
        if use_cache is None:
            use_cache = (not any(p.requires_grad for p in text_encoder.parameters())) and not text_encoder.training
        ids_h, am_h = input_ids.cpu(), attention_mask.cpu()
        uid, uam, inverse = dedupe_token_rows(ids_h, am_h)
        device = next(text_encoder.parameters()).device
        if not use_cache:
            uid, uam = bucket_token_length(uid, uam)
            pooled = text_encoder(input_ids=uid.to(device), attention_mask=uam.to(device)).pooler_output
            return pooled[inverse.to(device)]
        # the table belongs to one state of the encoder: other weights (load_state_dict, an optimiser step: in-place writes
        # bump the tensors' version counters), another dtype or device (to_bf16 swaps the storage) start a new table
        sig = self._signature(text_encoder)
        if sig != self.signature:
            self.clear()
            self.signature = sig
        # key = the label's own tokens, without the padding of this batch's longest member (the same label in a batch of
        # another width must hit)
        keys = [tuple(t for t, m in zip(r, a) if m) for r, a in zip(uid.tolist(), uam.tolist())]
        new = [i for i, k in enumerate(keys) if k not in self.rows]
        self.hits += len(keys) - len(new)
        self.misses += len(new)
        if new:
            if len(self.rows) + len(new) > self.max_entries:
                self.clear()
                new = list(range(len(keys)))
            nid, nam = bucket_token_length(uid[new], uam[new])
            with torch.no_grad():
                fresh = text_encoder(input_ids=nid.to(device), attention_mask=nam.to(device)).pooler_output
            base = 0 if self.table is None else self.table.shape[0]
            self.table = fresh if self.table is None else torch.cat((self.table, fresh), 0)
            for j, i in enumerate(new):
                self.rows[keys[i]] = base + j
        index = torch.tensor([self.rows[k] for k in keys], dtype=torch.long)[inverse]
        return self.table[index.to(device)]
