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


def test_parameter_without_gradient_in_the_round_is_skipped_like_the_reference():
    """a trainable parameter no batch of the round touches keeps `.grad = None`: torch.optim.AdamW then skips it (no weight
    decay, no step count) exactly as after the reference's single backward over the summed losses (engine.py:136-153);
    one that gets a gradient in SOME batch of the round is updated with the sum"""
    paradigm = [0, 1]
    model, batches = _toy()
    unused = torch.nn.Parameter(torch.ones(3))
    sometimes = torch.nn.Parameter(torch.full((3,), 0.5))
    params = list(model.parameters()) + [unused, sometimes]
    ref_model, _ = _toy()
    ref_some = torch.nn.Parameter(torch.full((3,), 0.5))
    ref_params = list(ref_model.parameters()) + [torch.nn.Parameter(torch.ones(3)), ref_some]
    opt = torch.optim.AdamW(params, lr=0.05, weight_decay=0.1)
    ref_opt = torch.optim.AdamW(ref_params, lr=0.05, weight_decay=0.1)
    upd = train.AccumulatedUpdate(params, opt, paradigm)
    ref = train.AccumulatedUpdate(ref_params, ref_opt, paradigm)
    for i, (x, y) in enumerate(batches[:4]):
        extra = (lambda p: (p * y[0]).sum()) if i % 2 == 1 else (lambda p: 0.0)
        ref.add(((ref_model(x) - y) ** 2).sum() + extra(ref_some))
        opt.zero_grad(set_to_none=True)
        (((model(x) - y) ** 2).sum() + extra(sometimes)).backward()
        upd.add_gradients()
    assert torch.equal(unused.detach(), torch.ones(3)) and unused not in opt.state
    torch.testing.assert_close(sometimes, ref_some, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(model.weight, ref_model.weight, rtol=1e-5, atol=1e-6)


class _Samples:
    def __init__(self, x):
        self.tensors = x


class _ToyCriterion:
    def __call__(self, outputs, targets):
        return {"l": ((outputs - targets) ** 2).sum()}

    @staticmethod
    def weighted_sum(d):
        return d["l"]


class _ToyStep(torch.nn.Module):
    def __init__(self, model):
        super().__init__()
        self.model = model

    def forward(self, samples, text, targets):
        return self.model(samples.tensors)


def test_train_round_runs_the_backward_of_step_objects():
    """train_round with a step object that owns its backward (EagerSyncStep here; GraphedStep / GraphedStepCache go through
    the same `run` protocol) must produce the update of the plain module -- not an optimiser step on all-zero gradients."""
    paradigm = [0, 1]
    m1, batches = _toy()
    m2, _ = _toy()
    crit = _ToyCriterion()
    o1 = torch.optim.AdamW(m1.parameters(), lr=0.05, weight_decay=1e-4)
    o2 = torch.optim.AdamW(m2.parameters(), lr=0.05, weight_decay=1e-4)
    plain = _ToyStep(m1)
    obj = train.EagerSyncStep(_ToyStep(m2), crit)
    before = m2.weight.detach().clone()
    for r in range(2):
        rb = [(_Samples(x), None, y) for x, y in batches[2 * r:2 * r + 2]]
        l1 = train.train_round(plain, crit, o1, rb, paradigm)
        l2 = train.train_round(obj, crit, o2, rb, paradigm)
        assert [float(a) for a in l1] == [float(b) for b in l2]
    assert not torch.equal(m2.weight.detach(), before)
    torch.testing.assert_close(m2.weight, m1.weight, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(m2.bias, m1.bias, rtol=1e-6, atol=1e-7)
