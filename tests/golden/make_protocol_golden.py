"""Golden vectors for BASELINE config 5's training protocol (mixed-dataset pre-training), generated in the build
container by EXECUTING the reference -- nothing of the reference is stored, only inputs' recipes and outputs:

  * `BatchIterativeDistributedSampler` (datasets/mixed_dataset.py:48-214) is extracted from its file with `ast`
    (the module's imports need cv2 / pycocotools) and iterated for several (sizes, batch, paradigm, world, rank) cases;
  * the gradient-update statement of `train_one_epoch` (engine.py:136-165: the `if args.gradient_strategy == ...`
    block) is extracted with `ast` and executed, iteration by iteration, on a closed-form toy model.

usage: python tests/golden/make_protocol_golden.py   -> tests/golden/protocol.json
"""
import ast
import json
import math
import os
from types import SimpleNamespace
from typing import Iterator, Optional, TypeVar

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def reference_sampler_class():
    tree = ast.parse(open(os.path.join(REF, "datasets", "mixed_dataset.py")).read())
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "BatchIterativeDistributedSampler"]
    from torch.utils.data import Dataset, Sampler
    ns = {"torch": torch, "math": math, "Sampler": Sampler, "Dataset": Dataset, "Optional": Optional, "Iterator": Iterator,
          "T_co": TypeVar("T_co", covariant=True), "dist": torch.distributed}
    exec(compile(ast.Module(body=cls, type_ignores=[]), "mixed_dataset_extract", "exec"), ns)
    return ns["BatchIterativeDistributedSampler"]


SAMPLER_CASES = [
    dict(sizes=[23, 40, 75], batch=2, paradigm="0,1,2,2", world=2, shuffle=True, seed=3, epoch=0, drop_last=False),
    dict(sizes=[23, 40, 75], batch=2, paradigm="0,1,2,2", world=2, shuffle=True, seed=3, epoch=1, drop_last=False),
    dict(sizes=[16, 33], batch=4, paradigm="0,1", world=1, shuffle=False, seed=0, epoch=0, drop_last=False),
    dict(sizes=[30, 100, 64], batch=3, paradigm="0,1,2", world=4, shuffle=True, seed=11, epoch=2, drop_last=True),
    dict(sizes=[9, 50], batch=2, paradigm="0,1,1,1", world=3, shuffle=True, seed=5, epoch=0, drop_last=False),
]


def sampler_golden():
    cls = reference_sampler_class()
    out = []
    for case in SAMPLER_CASES:
        ds = SimpleNamespace(datasets=[list(range(n)) for n in case["sizes"]])
        per_rank = []
        for rank in range(case["world"]):
            s = cls(ds, case["batch"], case["paradigm"], num_replicas=case["world"], rank=rank, shuffle=case["shuffle"],
                    seed=case["seed"], drop_last=case["drop_last"])
            s.set_epoch(case["epoch"])
            per_rank.append({"batches": [list(map(int, b)) for b in iter(s)], "len": int(len(s))})
        out.append({"case": case, "ranks": per_rank})
    return out


def toy_problem():
    """closed-form toy: Linear(4, 3) + 8 batches; the same recipe is rebuilt by the test"""
    model = torch.nn.Linear(4, 3)
    with torch.no_grad():
        model.weight.copy_(torch.arange(12, dtype=torch.float32).reshape(3, 4) * 0.1 - 0.5)
        model.bias.copy_(torch.tensor([0.1, -0.2, 0.3]))
    batches = []
    for k in range(8):
        x = torch.sin(torch.arange(20, dtype=torch.float32).reshape(5, 4) * (0.3 + 0.1 * k))
        y = torch.cos(torch.arange(15, dtype=torch.float32).reshape(5, 3) * (0.2 + 0.05 * k))
        batches.append((x, y))
    return model, batches


def update_golden():
    tree = ast.parse(open(os.path.join(REF, "engine.py")).read())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "train_one_epoch"][0]
    stmt = None
    for node in ast.walk(fn):
        if isinstance(node, ast.If) and isinstance(node.test, ast.Compare) and "gradient_strategy" in ast.dump(node.test) \
                and "gradient_accumulation" in ast.dump(node.test) and "accumulation_losses" in ast.dump(node):
            stmt = node
            break
    assert stmt is not None
    code = compile(ast.Module(body=[stmt], type_ignores=[]), "engine_update_extract", "exec")
    out = {}
    for strategy, paradigm in (("gradient_accumulation", [0, 1, 2, 2]), ("gradient_accumulation", [0, 1]), ("vanilla", [0, 1, 2])):
        model, batches = toy_problem()
        optimizer = torch.optim.AdamW(model.parameters(), lr=0.05, weight_decay=1e-4)
        ns = {"args": SimpleNamespace(gradient_strategy=strategy), "iterative_paradigm": paradigm, "optimizer": optimizer,
              "model": model, "max_norm": 0.1, "torch": torch}
        trace = []
        for i, (x, y) in enumerate(batches):
            ns["i"] = i
            ns["losses"] = ((model(x) - y) ** 2).sum() * (1.0 + 0.25 * paradigm[i % len(paradigm)])
            exec(code, ns)
            trace.append([float(v) for v in model.weight.detach().flatten()] + [float(v) for v in model.bias.detach()])
        out[f"{strategy}:{','.join(map(str, paradigm))}"] = trace
    return out


def main():
    rec = {"sampler": sampler_golden(), "update": update_golden()}
    with open(os.path.join(HERE, "protocol.json"), "w") as f:
        json.dump(rec, f)
    print("wrote protocol.json:", len(rec["sampler"]), "sampler cases,", len(rec["update"]), "update traces")


if __name__ == "__main__":
    main()
