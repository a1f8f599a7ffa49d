"""BASELINE config 5's protocol (mixed-dataset pre-training): the iterative batch sampler and the gradient-accumulation
update against goldens produced by EXECUTING the reference (tests/golden/make_protocol_golden.py: the sampler class of
datasets/mixed_dataset.py:48-214 and the update statement of engine.py:136-165, both extracted with `ast`)."""
import json
import os

import pytest
import torch

from rlipv2_amd import train

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "protocol.json")))


@pytest.mark.parametrize("k", range(len(GOLD["sampler"])))
def test_batch_iterative_sampler_yields_the_reference_batches(k):
    rec = GOLD["sampler"][k]
    c = rec["case"]
    for rank, want in enumerate(rec["ranks"]):
        s = train.BatchIterativeSampler(c["sizes"], c["batch"], c["paradigm"], num_replicas=c["world"], rank=rank,
                                        shuffle=c["shuffle"], seed=c["seed"], drop_last=c["drop_last"])
        s.set_epoch(c["epoch"])
        assert [list(b) for b in s] == want["batches"]
        assert len(s) == want["len"]
    # the ranks of one batch partition it: batch j of every rank comes from dataset paradigm[j % len]
    paradigm = [int(d) for d in c["paradigm"].split(",")]
    bounds = [0]
    for n in c["sizes"]:
        bounds.append(bounds[-1] + n)
    for j, b in enumerate(rec["ranks"][0]["batches"]):
        d = paradigm[j % len(paradigm)]
        assert all(bounds[d] <= i < bounds[d + 1] for i in b)


def _toy():
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


@pytest.mark.parametrize("key", sorted(GOLD["update"]))
def test_accumulated_update_follows_the_reference_loop(key):
    strategy, paradigm = key.split(":")
    paradigm = [int(d) for d in paradigm.split(",")]
    model, batches = _toy()
    opt = torch.optim.AdamW(model.parameters(), lr=0.05, weight_decay=1e-4)
    upd = train.AccumulatedUpdate(model.parameters(), opt, paradigm, strategy=strategy, max_norm=0.1)
    for i, (x, y) in enumerate(batches):
        loss = ((model(x) - y) ** 2).sum() * (1.0 + 0.25 * paradigm[i % len(paradigm)])
        stepped = upd.add(loss)
        assert stepped == (strategy == "vanilla" or (i + 1) % len(paradigm) == 0)
        got = [float(v) for v in model.weight.detach().flatten()] + [float(v) for v in model.bias.detach()]
        assert got == GOLD["update"][key][i], (i, got, GOLD["update"][key][i])       # float32 on the CPU: bit for bit


def test_gradient_form_of_the_accumulated_update_is_the_same_update():
    """add_gradients (per-batch backward, float32 accumulators: what the graphed step uses) == add (one backward over the
    summed loss, the reference's form)"""
    paradigm = [0, 1, 2, 2]
    ref_model, batches = _toy()
    ref_opt = torch.optim.AdamW(ref_model.parameters(), lr=0.05, weight_decay=1e-4)
    ref = train.AccumulatedUpdate(ref_model.parameters(), ref_opt, paradigm)
    model, _ = _toy()
    opt = torch.optim.AdamW(model.parameters(), lr=0.05, weight_decay=1e-4)
    upd = train.AccumulatedUpdate(model.parameters(), opt, paradigm)
    for i, (x, y) in enumerate(batches):
        w = 1.0 + 0.25 * paradigm[i % 4]
        ref.add(((ref_model(x) - y) ** 2).sum() * w)
        opt.zero_grad()
        (((model(x) - y) ** 2).sum() * w).backward()
        upd.add_gradients()
        torch.testing.assert_close(model.weight, ref_model.weight, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(model.bias, ref_model.bias, rtol=1e-5, atol=1e-6)


def test_accumulation_needs_a_real_paradigm():
    model, _ = _toy()
    with pytest.raises(AssertionError):
        train.AccumulatedUpdate(model.parameters(), torch.optim.SGD(model.parameters(), lr=0.1), [0])
    with pytest.raises(ValueError):
        train.AccumulatedUpdate(model.parameters(), torch.optim.SGD(model.parameters(), lr=0.1), [0, 1], strategy="other")
