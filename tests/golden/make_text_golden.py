"""Golden vectors for the per-batch text merging (engine.merge_batch_data and helpers), generated in the
build container.  engine.py cannot be imported (its module-level imports need pycocotools), so the four
functions are extracted from the file with `ast` and executed here -- nothing of the reference is stored,
only the inputs' generator and the outputs.

usage: python tests/golden/make_text_golden.py
"""
import ast
import json
import os
import random

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def reference_functions():
    src = open(os.path.join(REF, "engine.py")).read()
    tree = ast.parse(src)
    wanted = {"merge_batch_data", "merge_obj_text", "merge_verb_text", "sample_text"}
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in wanted]
    ns = {"torch": torch, "np": np, "F": torch.nn.functional, "choice": random.choice, "uniform": random.uniform}
    exec(compile(ast.Module(body=body, type_ignores=[]), "engine_extract", "exec"), ns)
    return ns


OBJ_VOCAB = ["person", "dog", "cat", "horse", "bicycle", "car", "tree", "cup", "table", "chair", "ball", "kite",
             "boat", "bird", "phone", "book", "bench", "umbrella"]
REL_VOCAB = ["ride", "hold", "sit on", "look at", "carry", "throw", "kick", "feed", "walk", "read", "talk on",
             "stand under", "fly"]


def batch_case():
    """3 images with overlapping, differently ordered name lists; image 2 has no triplets"""
    texts = [(["person", "horse", "tree"], ["ride", "look at", "feed"]),
             (["dog", "person", "ball", "tree"], ["throw", "look at", "kick", "hold"]),
             (["cup", "table"], ["sit on"])]
    targets = [
        {"obj_labels": torch.tensor([1, 2]), "sub_labels": torch.tensor([0, 0]),
         "verb_labels": torch.tensor([[1., 1., 0.], [0., 1., 0.]])},
        {"obj_labels": torch.tensor([2, 0, 3]), "sub_labels": torch.tensor([1, 1, 0]),
         "verb_labels": torch.tensor([[1., 0., 1., 0.], [0., 1., 0., 1.], [0., 1., 0., 0.]])},
        {"obj_labels": torch.zeros(0, dtype=torch.int64), "sub_labels": torch.zeros(0, dtype=torch.int64),
         "verb_labels": torch.zeros(0, 1)},
    ]
    return texts, targets


def vocab_case():
    g = torch.Generator().manual_seed(17)
    obj_freq = {n: int(v) for n, v in zip(OBJ_VOCAB, torch.randint(1, 200, (len(OBJ_VOCAB),), generator=g))}
    rel_freq = {n: int(v) for n, v in zip(REL_VOCAB, torch.randint(1, 200, (len(REL_VOCAB),), generator=g))}
    obj_feat = torch.randn(len(OBJ_VOCAB), 16, generator=g)
    rel_feat = torch.randn(len(REL_VOCAB), 16, generator=g)
    return obj_freq, rel_freq, obj_feat, rel_feat


class _Dataset:
    pass


class _Loader:
    def __init__(self):
        obj_freq, rel_freq, obj_feat, rel_feat = vocab_case()
        d = _Dataset()
        d.object_names, d.relationship_names = list(OBJ_VOCAB), list(REL_VOCAB)
        d.object_freq, d.relationship_freq = obj_freq, rel_freq
        d.obj_feature, d.rel_feature = (list(OBJ_VOCAB), obj_feat), (list(REL_VOCAB), rel_feat)
        self.dataset = d


def main():
    ns = reference_functions()
    rec = {}
    for strategy in ("random", "freq", "hard_mining", "freq+random"):
        texts, targets = batch_case()
        random.seed(1234)
        out = ns["merge_batch_data"]({"targets": targets, "text": texts}, use_no_obj_token=True,
                                     use_all_text_labels=False, negative_text_sampling=24,
                                     sampling_stategy=strategy, data_loader=_Loader())
        (obj_names, verb_names), = out["text"]
        rec[strategy] = {"obj_names": obj_names, "verb_names": verb_names,
                         "targets": [{k: v.tolist() for k, v in t.items()} for t in out["targets"]]}
    with open(os.path.join(HERE, "text_merge.json"), "w") as f:
        json.dump(rec, f, indent=1)
    print("text_merge.json written:", {k: (len(v["obj_names"]), len(v["verb_names"])) for k, v in rec.items()})


if __name__ == "__main__":
    main()
