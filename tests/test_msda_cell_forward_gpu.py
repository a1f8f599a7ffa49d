"""The experimental forward variant MSDA_VARIANT_CELL (csrc/msda_patch.hip: cell_forward_kernel -- per-cell sampling
windows in LDS, bilinear sums on the matrix cores) against the CPU oracle and against the product forward kernel.

The kernel was written while no GPU was available to the build and has never run on hardware: every test here is a
`first_contact` test (tests/conftest.py) -- it runs in a child process with a timeout and is reported as XPASS / XFAIL, so that
an unvalidated kernel can neither hang nor colour the GPU suite; the file is sorted last.  Tolerance: bfloat16
output, weights split hi + lo (2^-16 relative) -> the same bar as every other bfloat16 forward kernel of the suite."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import msda_oracle as O  # noqa: E402
from conftest import load_golden  # noqa: E402

from rlipv2_amd import msda  # noqa: E402
from test_msda_gpu import PYRAMID, bf16_round, random_problem, run_hip  # noqa: E402

pytestmark = [pytest.mark.gpu, pytest.mark.first_contact(timeout=300)]


def _check(value, shapes, starts, loc, aw):
    vb = bf16_round(value)
    ref = O.forward(vb.astype(np.float64), shapes, starts, loc.astype(np.float64), aw.astype(np.float64))
    out = run_hip(("cell", "quad"), torch.bfloat16, vb, shapes, starts, loc, aw)[0]
    assert np.isfinite(out).all()
    np.testing.assert_allclose(out, ref, rtol=2.0 ** -7, atol=1e-3 * float(np.abs(ref).max()))
    prod = run_hip(("quad", "quad"), torch.bfloat16, vb, shapes, starts, loc, aw)[0]
    # two bfloat16 roundings of nearly the same float32 sum: at most one unit in the last place apart, rarely
    np.testing.assert_allclose(out, prod, rtol=2.0 ** -7, atol=1e-3 * float(np.abs(ref).max()))


@pytest.mark.parametrize("case", ["model_enc", "pyr_enc"])
def test_cell_forward_goldens(case):
    g = load_golden(case)                      # includes samples on / beyond every level border
    _check(g["value"], g["shapes"], g["starts"], g["loc"], g["aw"])


@pytest.mark.parametrize("spread", [1.0, 3.0, 40.0])
def test_cell_forward_random_encoder_problem(spread):
    """spread 40 px: windows that do not fit the LDS budget -- the in-kernel direct route"""
    rng = np.random.default_rng(17)
    value, shapes, starts, loc, aw, _ = random_problem(rng, 3, PYRAMID, 8, 32, 77, 4, spread=spread, enc=True)
    _check(value, shapes, starts, loc, aw)


def test_cell_forward_full_size_agrees_with_the_product_kernel():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    from tools.msda_inputs import make_inputs
    for mode in ("model", "init", "uniform"):
        inp = make_inputs(2, mode=mode, dtype=torch.bfloat16, seed=5)
        a = (inp["value"], inp["shapes"], inp["starts"], inp["loc"], inp["aw"])
        msda.set_variant("cell", "quad")
        try:
            out = msda.ms_deform_attn_forward(*a, 64).float()
        finally:
            msda.set_variant("auto")
        ref = msda.ms_deform_attn_forward(*a, 64).float()
        assert torch.isfinite(out).all()
        torch.testing.assert_close(out, ref, rtol=2.0 ** -7, atol=1e-3 * float(ref.abs().max()))


def test_cell_forward_is_refused_where_it_does_not_apply():
    rng = np.random.default_rng(3)
    value, shapes, starts, loc, aw, _ = random_problem(rng, 1, PYRAMID, 8, 32, 50, 4, enc=False)     # Lq != S
    with pytest.raises(RuntimeError):
        run_hip(("cell", "quad"), torch.bfloat16, bf16_round(value), shapes, starts, loc, aw)
    value, shapes, starts, loc, aw, _ = random_problem(rng, 1, PYRAMID, 8, 32, 50, 4, enc=True)
    with pytest.raises(RuntimeError):
        run_hip(("cell", "quad"), torch.float32, value, shapes, starts, loc, aw)                       # bfloat16 only
