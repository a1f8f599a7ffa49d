"""The native batched assignment (csrc/hoi_assign.hip, include/rlipv2_matcher.h) against scipy.optimize.linear_sum_assignment,
the solver the reference calls per image (models/matcher.py:91): identical (row, col) pairs, not merely equal cost -- random
float costs, integer costs with many ties, rectangular either way, empty images, non-finite entries."""
import ctypes

import numpy as np
import pytest
import torch
from scipy.optimize import linear_sum_assignment


def _native(C, sizes):
    from rlipv2_amd import _lib
    K, bs, nq, T = C.shape
    n = K * sum(min(nq, s) for s in sizes)
    out = torch.empty(2, max(n, 1), dtype=torch.int64)
    got = _lib.lib().hoi_assign_batch(C.data_ptr(), K, bs, nq, (ctypes.c_int * bs)(*sizes), out[0].data_ptr(),
                                      out[1].data_ptr(), n)
    return got, out[:, :n]


def _scipy(C, sizes):
    K, bs, nq, T = C.shape
    rows, cols = [], []
    for k in range(K):
        s = 0
        for i, n in enumerate(sizes):
            r, c = linear_sum_assignment(C[k, i, :, s:s + n].numpy())
            rows.append(torch.as_tensor(r, dtype=torch.int64) + (k * bs + i) * nq)
            cols.append(torch.as_tensor(c, dtype=torch.int64) + s)
            s += n
    return torch.stack([torch.cat(rows), torch.cat(cols)])


@pytest.mark.parametrize("nq,sizes,K", [(300, [8, 8, 8, 8], 3), (150, [6, 0, 11, 3], 4), (5, [9, 2], 2), (7, [7], 1),
                                        (1, [1, 1], 1), (40, [40, 41, 39], 2)])
@pytest.mark.parametrize("kind", ["float", "ties", "few_values"])
def test_native_assignment_equals_scipy(nq, sizes, K, kind):
    g = torch.Generator().manual_seed(nq * 31 + sum(sizes) + K)
    shape = (K, len(sizes), nq, sum(sizes))
    if kind == "float":
        C = torch.randn(shape, generator=g)
    elif kind == "ties":
        C = torch.randint(0, 4, shape, generator=g).float()
    else:
        C = torch.randint(0, 2, shape, generator=g).float() * 0.5 - torch.randint(0, 2, shape, generator=g).float()
    got, pairs = _native(C.contiguous(), sizes)
    want = _scipy(C, sizes)
    assert got == want.shape[1]
    assert torch.equal(pairs, want)


def test_native_assignment_rejects_what_scipy_rejects():
    C = torch.zeros(1, 1, 4, 3)
    C[0, 0, 1, 1] = float("nan")
    assert _native(C, [3])[0] == -1
    C[0, 0, 1, 1] = float("-inf")
    assert _native(C, [3])[0] == -1
    C[0, 0, 1, 1] = float("inf")                      # +inf is a legal (forbidden-edge) cost
    got, pairs = _native(C, [3])
    assert got == 3 and torch.equal(pairs, _scipy(C, [3]))
    C[0, 0, :, 1] = float("inf")                      # a target nobody may take: infeasible
    assert _native(C, [3])[0] == -1
    with pytest.raises(ValueError):
        linear_sum_assignment(C[0, 0].numpy())


def test_criterion_assign_uses_the_native_solver_and_matches_scipy(monkeypatch):
    from rlipv2_amd import criterion as crit
    g = torch.Generator().manual_seed(3)
    K, bs, nq, sizes = 3, 2, 20, [4, 7]
    state = {"K": K, "bs": bs, "nq": nq, "sizes": sizes, "C": torch.randn(K * bs * nq, sum(sizes), generator=g)}
    obj = crit.SetCriterionHOI.__new__(crit.SetCriterionHOI)
    a = crit.SetCriterionHOI.assign(obj, state)
    monkeypatch.setattr(crit, "native_assignment", False)
    b = crit.SetCriterionHOI.assign(obj, state)
    assert torch.equal(a, b) and a.shape == (2, K * sum(sizes))
