"""Host-side logic of rlipv2_amd.optim.FusedMasterAdamW that needs no GPU (the update itself is a HIP kernel pair,
tests/test_optim_gpu.py): which gradients are collected, and that the per-step layout check is skipped for gradient objects
that already passed it."""
import torch

from rlipv2_amd import optim


def _bare(params):
    o = object.__new__(optim.FusedMasterAdamW)               # (the constructor refuses CPU parameters, as the product must)
    o.params, o._relayout, o._checked = params, {}, [None] * len(params)
    return o


def test_gradient_collection_and_layout_check_caching():
    ps = [torch.nn.Parameter(torch.zeros(4, 6, dtype=torch.bfloat16)) for _ in range(5)]
    ps[3] = torch.nn.Parameter(torch.zeros(2, 3, 4, 5, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last))
    o = _bare(ps)
    grads = [torch.ones_like(p) for p in ps]
    grads[3] = torch.ones(2, 3, 4, 5, dtype=torch.bfloat16)  # arrives contiguous for a channels-last parameter: re-laid out
    for p, g in zip(ps, grads):
        p.grad = g
    ps[1].grad = None                                        # no gradient this step: skipped, as torch.optim does
    calls = []
    real = optim._same_layout
    optim._same_layout = lambda a, b: (calls.append(1), real(a, b))[1]
    try:
        idx, got = o._grads()
        assert idx == [0, 2, 3, 4] and len(calls) == 4
        assert got[0] is grads[0] and got[2] is not grads[3] and optim._same_layout.__name__ == "<lambda>"
        assert real(got[2], ps[3]) and torch.equal(got[2], grads[3])
        calls.clear()
        idx2, got2 = o._grads()                              # same objects again: only the re-laid-out one is looked at
        assert idx2 == idx and len(calls) == 1 and all(a is b for a, b in zip(got, got2))
        ps[1].grad = torch.ones_like(ps[1])                  # a parameter joins, another one's gradient object changes
        ps[0].grad = torch.ones_like(ps[0])
        calls.clear()
        idx3, got3 = o._grads()
        assert idx3 == [0, 1, 2, 3, 4] and len(calls) == 3 and got3[0] is ps[0].grad
    finally:
        optim._same_layout = real
