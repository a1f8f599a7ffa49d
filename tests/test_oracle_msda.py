"""CPU tests: pin the oracle (C and numpy restatements) to the reference's vectors.

The golden .npz files were produced by importing the reference's own
`ms_deform_attn_core_pytorch` (models/ops/functions/ms_deform_attn_func.py:45-65) and
differentiating it with autograd -- see tests/golden/make_msda_golden.py.  Tolerances:
float64 -> torch.allclose defaults (rtol 1e-5, atol 1e-8), the reference's own bar
(models/ops/test.py:44); float32 -> rtol 1e-2 / atol 1e-3 is the reference's bar
(test.py:60), we hold the oracle to a much tighter 1e-4 / 1e-5.
"""
import numpy as np
import pytest

from conftest import boundary_samples, kink_samples
from oracle import msda_oracle as O


def _args(g, dt):
    return (g["value"].astype(dt), g["shapes"], g["starts"], g["loc"].astype(dt), g["aw"].astype(dt))


@pytest.mark.parametrize("impl", ["c", "numpy"])
def test_forward_f64_matches_reference(msda_golden, impl):
    g = msda_golden
    fwd = O.forward if impl == "c" else O.forward_numpy
    out = fwd(*_args(g, np.float64))
    np.testing.assert_allclose(out, g["out_f64"], rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("impl", ["c", "numpy"])
def test_backward_f64_matches_reference(msda_golden, impl):
    g = msda_golden
    bwd = O.backward if impl == "c" else O.backward_numpy
    gv, gl, ga = bwd(*_args(g, np.float64), g["grad_out"].astype(np.float64))
    np.testing.assert_allclose(gv, g["g_value_f64"], rtol=1e-5, atol=1e-8)
    keep = ~boundary_samples(g)          # see conftest.boundary_samples
    np.testing.assert_allclose(gl[keep], g["g_loc_f64"][keep], rtol=1e-5, atol=1e-8)
    assert np.all(gl[~keep][np.abs(g["g_loc_f64"][~keep]) == 0] == 0)
    np.testing.assert_allclose(ga, g["g_aw_f64"], rtol=1e-5, atol=1e-8)


def test_f32_matches_reference_f32(msda_golden):
    g = msda_golden
    out = O.forward(*_args(g, np.float32))
    gv, gl, ga = O.backward(*_args(g, np.float32), g["grad_out"])
    scale = lambda ref: 1e-5 * max(1.0, float(np.abs(ref).max()))
    np.testing.assert_allclose(out, g["out_f32"], rtol=1e-4, atol=scale(g["out_f32"]))
    np.testing.assert_allclose(gv, g["g_value_f32"], rtol=1e-4, atol=scale(g["g_value_f32"]))
    keep = ~kink_samples(g)
    np.testing.assert_allclose(gl[keep], g["g_loc_f32"][keep], rtol=1e-4, atol=scale(g["g_loc_f32"]))
    np.testing.assert_allclose(ga, g["g_aw_f32"], rtol=1e-4, atol=scale(g["g_aw_f32"]))


def test_openmp_build_equals_serial():
    from conftest import load_golden
    g = load_golden("model_dec")
    a = _args(g, np.float32)
    np.testing.assert_array_equal(O.forward(*a), O.forward(*a, omp=True))
    s = O.backward(*a, g["grad_out"])
    p = O.backward(*a, g["grad_out"], omp=True)
    for x, y in zip(s, p):
        np.testing.assert_array_equal(x, y)


def test_excluded_boundary_samples_are_zero():
    """ms_deform_im2col_cuda.cuh:285: h_im == H and w_im == -1 contribute nothing."""
    shapes = np.array([[4, 5]], dtype=np.int64)
    starts = np.array([0], dtype=np.int64)
    value = np.ones((1, 20, 1, 3))
    aw = np.ones((1, 1, 1, 1, 2))
    loc = np.array([[-0.5 / 5, 0.5], [0.5, 4.5 / 4]]).reshape(1, 1, 1, 1, 2, 2)
    assert np.all(O.forward(value, shapes, starts, loc, aw) == 0)
    assert np.all(O.forward_numpy(value, shapes, starts, loc, aw) == 0)
    loc2 = np.array([[0.0, 0.0], [1.0, 1.0]]).reshape(1, 1, 1, 1, 2, 2)   # corners: weight 1/4 each
    np.testing.assert_allclose(O.forward(value, shapes, starts, loc2, aw), 0.5)


def test_nan_location_is_skipped():
    shapes = np.array([[2, 2]], dtype=np.int64)
    starts = np.array([0], dtype=np.int64)
    value = np.ones((1, 4, 1, 2), dtype=np.float32)
    aw = np.ones((1, 1, 1, 1, 1), dtype=np.float32)
    loc = np.full((1, 1, 1, 1, 1, 2), np.nan, dtype=np.float32)
    assert np.all(O.forward(value, shapes, starts, loc, aw) == 0)
    assert np.all(O.forward_numpy(value, shapes, starts, loc, aw) == 0)


def test_torch_grid_sample_restatement_matches_golden():
    """oracle.forward_torch (per-level grid_sample, the reference's CPU formulation,
    ms_deform_attn_func.py:45-65) against the reference-generated goldens: output and, away from the
    exclusion boundary, all three gradients through autograd."""
    import torch
    from conftest import MSDA_GOLDEN_CASES, boundary_samples, load_golden
    for case in MSDA_GOLDEN_CASES:
        g = load_golden(case)
        value = torch.from_numpy(g["value"]).double().requires_grad_(True)
        loc = torch.from_numpy(g["loc"]).double().requires_grad_(True)
        aw = torch.from_numpy(g["aw"]).double().requires_grad_(True)
        out = O.forward_torch(value, [tuple(int(v) for v in hw) for hw in g["shapes"]], loc, aw)
        np.testing.assert_allclose(out.detach().numpy(), g["out_f64"], rtol=1e-9, atol=1e-12)
        out.backward(torch.from_numpy(g["grad_out"]).double().reshape(out.shape))
        np.testing.assert_allclose(value.grad.numpy(), g["g_value_f64"], rtol=1e-8, atol=1e-11)
        np.testing.assert_allclose(aw.grad.numpy(), g["g_aw_f64"], rtol=1e-8, atol=1e-11)
        keep = ~boundary_samples(g)
        np.testing.assert_allclose(loc.grad.numpy()[keep], g["g_loc_f64"][keep], rtol=1e-8, atol=1e-10)
