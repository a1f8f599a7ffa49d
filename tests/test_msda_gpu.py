"""GPU parity tests (run on the MI355X box with -m gpu): the HIP kernels, called through the
C ABI, against the CPU oracle and the committed golden vectors.

Tolerances
----------
float64 : rtol 1e-5 / atol 1e-8 -- torch.allclose defaults, the reference's own bar
          (models/ops/test.py:44).
float32 : rtol 1e-4, atol 1e-5 x max|ref| -- two orders tighter than the reference's
          rtol 1e-2 / atol 1e-3 (test.py:60); grad_value sums up to hundreds of float atomics
          in arbitrary order, hence the relative-to-max floor.
bfloat16: value / grad_out are rounded to bf16 BEFORE the oracle sees them (the kernels
          accumulate in float32), so outputs differ only by the final rounding of `out` to
          bf16: rtol 2^-7.  Gradients are float32 and use the float32 tolerance.
"""
import numpy as np
import pytest
import torch

from conftest import MSDA_GOLDEN_CASES, boundary_samples, kink_samples, load_golden
from oracle import msda_oracle as O
from rlipv2_amd import _lib, msda

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _starts(shapes):
    hw = shapes[:, 0] * shapes[:, 1]
    return np.concatenate([[0], np.cumsum(hw)[:-1]]).astype(np.int64)


def _to_dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return t if dtype is None else t.to(dtype)


def run_hip(variant, tdtype, value, shapes, starts, loc, aw, grad_out=None):
    """numpy in -> numpy out through rlipv2_amd.msda (C ABI underneath)."""
    aux = torch.float64 if tdtype == torch.float64 else torch.float32
    msda.set_variant(*variant) if isinstance(variant, tuple) else msda.set_variant(variant)
    try:
        v = _to_dev(value, tdtype)
        sh, st = _to_dev(shapes), _to_dev(starts)
        l, a = _to_dev(loc, aux), _to_dev(aw, aux)
        out = msda.ms_deform_attn_forward(v, sh, st, l, a, 64)
        res = [out.float().cpu().numpy() if tdtype == torch.bfloat16 else out.cpu().numpy()]
        if grad_out is not None:
            go = _to_dev(grad_out, tdtype)
            gv, gl, ga = msda.ms_deform_attn_backward(v, sh, st, l, a, go, 64)
            res += [gv.float().cpu().numpy() if tdtype == torch.bfloat16 else gv.cpu().numpy(),
                    gl.cpu().numpy(), ga.cpu().numpy()]
        torch.cuda.synchronize()
    finally:
        msda.set_variant("auto")
    return res


def bf16_round(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).bfloat16().float().numpy()


def close32(got, ref, rtol=1e-4, atol_rel=1e-5):
    np.testing.assert_allclose(got, ref, rtol=rtol, atol=atol_rel * max(1.0, float(np.abs(ref).max())))


WINDOW_FWD = True    # window-staged tile forward (msda_quad.hip), valid when Lq == S
WINDOW_BWD = True


def variants_for(D, L, P, tdtype, S=0, Lq=-1):
    """(forward, backward) kernel pairs to exercise for a problem."""
    v = [("generic", "generic")]
    if D == 32 and L == 4 and P == 4 and tdtype != torch.float64:
        v.append(("quad", "quad"))
        # "window" backward = reduce kernel + sorted scatter kernel; valid for any Lq
        v.append(("window" if (WINDOW_FWD and S == Lq) else "quad", "window" if WINDOW_BWD else "quad"))
        v.append(("quad", "dest"))       # destination-stationary grad_value (msda_dest.hip)
        if tdtype == torch.bfloat16 and Lq >= 4096:
            v.append(("coarse", "dest"))  # forward with the coarse levels resident in LDS (msda_quad.hip)
    v.append(("auto", "auto"))
    return v


def random_problem(rng, N, shapes, M, D, Lq, P, spread=2.0, enc=False):
    shapes = np.asarray(shapes, dtype=np.int64)
    L = shapes.shape[0]
    S = int((shapes[:, 0] * shapes[:, 1]).sum())
    value = rng.standard_normal((N, S, M, D)).astype(np.float32)
    if enc:
        ref = []
        for (H, W) in shapes:
            ys, xs = np.meshgrid((np.arange(H) + 0.5) / H, (np.arange(W) + 0.5) / W, indexing="ij")
            ref.append(np.stack([xs.ravel(), ys.ravel()], -1))
        ref = np.concatenate(ref, 0)[None].repeat(N, 0)
        Lq = S
    else:
        ref = rng.uniform(-0.05, 1.05, size=(N, Lq, 2))
    off = rng.standard_normal((N, Lq, M, L, P, 2)) * spread
    norm = np.stack([shapes[:, 1], shapes[:, 0]], -1).astype(np.float64)
    loc = (ref[:, :, None, None, None, :] + off / norm[None, None, None, :, None, :]).astype(np.float32)
    logits = rng.standard_normal((N, Lq, M, L * P))
    aw = np.exp(logits - logits.max(-1, keepdims=True))
    aw = (aw / aw.sum(-1, keepdims=True)).reshape(N, Lq, M, L, P).astype(np.float32)
    grad_out = rng.standard_normal((N, Lq, M * D)).astype(np.float32)
    return value, shapes, _starts(shapes), loc, aw, grad_out


# ---------------------------------------------------------------------------------------------
# golden vectors (reference-generated)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", MSDA_GOLDEN_CASES)
def test_golden_f64(case):
    g = load_golden(case)
    out, gv, gl, ga = run_hip("generic", torch.float64, g["value"], g["shapes"], g["starts"], g["loc"], g["aw"],
                              g["grad_out"])
    np.testing.assert_allclose(out, g["out_f64"], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(gv, g["g_value_f64"], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(ga, g["g_aw_f64"], rtol=1e-5, atol=1e-8)
    keep = ~boundary_samples(g)
    np.testing.assert_allclose(gl[keep], g["g_loc_f64"][keep], rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("case", MSDA_GOLDEN_CASES)
def test_golden_f32_all_variants(case):
    g = load_golden(case)
    N, S, M, D = g["value"].shape
    L, P = g["loc"].shape[3], g["loc"].shape[4]
    keep = ~kink_samples(g)
    for variant in variants_for(D, L, P, torch.float32, S, g["loc"].shape[1]):
        out, gv, gl, ga = run_hip(variant, torch.float32, g["value"], g["shapes"], g["starts"], g["loc"], g["aw"],
                                  g["grad_out"])
        close32(out, g["out_f32"])
        close32(gv, g["g_value_f32"])
        close32(ga, g["g_aw_f32"])
        close32(gl[keep], g["g_loc_f32"][keep])


@pytest.mark.parametrize("case", ["model_enc", "model_dec", "pyr_enc", "pyr_dec", "testpy_d32", "testpy_d71"])
def test_golden_bf16_all_variants(case):
    g = load_golden(case)
    N, S, M, D = g["value"].shape
    L, P = g["loc"].shape[3], g["loc"].shape[4]
    vb, gob = bf16_round(g["value"]), bf16_round(g["grad_out"])
    args = (vb.astype(np.float64), g["shapes"], g["starts"], g["loc"].astype(np.float64), g["aw"].astype(np.float64))
    ref_out = O.forward(*args)
    ref_gv, ref_gl, ref_ga = O.backward(*args, gob.astype(np.float64))
    keep = ~kink_samples(g)
    for variant in variants_for(D, L, P, torch.bfloat16, S, g["loc"].shape[1]):
        out, gv, gl, ga = run_hip(variant, torch.bfloat16, vb, g["shapes"], g["starts"], g["loc"], g["aw"], gob)
        np.testing.assert_allclose(out, ref_out, rtol=2.0 ** -7, atol=1e-3 * float(np.abs(ref_out).max()))
        # grad_value comes back rounded to bf16 through the torch-facing wrapper
        np.testing.assert_allclose(gv, ref_gv, rtol=2.0 ** -7, atol=1e-3 * float(np.abs(ref_gv).max()))
        close32(ga, ref_ga)
        close32(gl[keep], ref_gl[keep])


# ---------------------------------------------------------------------------------------------
# seeded random problems vs the oracle (sizes the oracle finishes in seconds)
# ---------------------------------------------------------------------------------------------
PYRAMID = [(25, 34), (13, 17), (7, 9), (4, 5)]


@pytest.mark.parametrize("enc", [True, False])
@pytest.mark.parametrize("tdtype", [torch.float32, torch.bfloat16])
def test_random_model_shape_vs_oracle(enc, tdtype):
    rng = np.random.default_rng(5 + int(enc))
    value, shapes, starts, loc, aw, go = random_problem(rng, 3, PYRAMID, 8, 32, 77, 4, spread=3.0, enc=enc)
    if tdtype == torch.bfloat16:
        value, go = bf16_round(value), bf16_round(go)
    ref_out = O.forward(value.astype(np.float64), shapes, starts, loc.astype(np.float64), aw.astype(np.float64))
    ref = O.backward(value.astype(np.float64), shapes, starts, loc.astype(np.float64), aw.astype(np.float64),
                     go.astype(np.float64))
    g = dict(loc=loc, shapes=shapes)
    keep = ~kink_samples(g)
    for variant in variants_for(32, 4, 4, tdtype, value.shape[1], loc.shape[1]):
        out, gv, gl, ga = run_hip(variant, tdtype, value, shapes, starts, loc, aw, go)
        if tdtype == torch.bfloat16:
            np.testing.assert_allclose(out, ref_out, rtol=2.0 ** -7, atol=1e-3)
            np.testing.assert_allclose(gv, ref[0], rtol=2.0 ** -7, atol=2e-3 * float(np.abs(ref[0]).max()))
        else:
            close32(out, ref_out)
            close32(gv, ref[0])
        close32(gl[keep], ref[1][keep])
        close32(ga, ref[2])


@pytest.mark.parametrize("M,D,L,P", [(2, 2, 2, 2), (3, 30, 1, 5), (8, 64, 4, 4), (1, 71, 3, 2), (4, 32, 4, 2),
                                     (2, 130, 2, 3)])
def test_generic_shapes_vs_oracle(M, D, L, P):
    rng = np.random.default_rng(M * 1000 + D)
    shapes = [(9, 11), (5, 6), (3, 3), (2, 2)][:L]
    value, shapes, starts, loc, aw, go = random_problem(rng, 2, shapes, M, D, 13, P)
    for tdtype, npdt in ((torch.float64, np.float64), (torch.float32, np.float32)):
        ref_out = O.forward(value.astype(npdt), shapes, starts, loc.astype(npdt), aw.astype(npdt))
        ref = O.backward(value.astype(npdt), shapes, starts, loc.astype(npdt), aw.astype(npdt), go.astype(npdt))
        out, gv, gl, ga = run_hip("auto", tdtype, value, shapes, starts, loc, aw, go)
        if tdtype == torch.float64:
            for a, b in ((out, ref_out), (gv, ref[0]), (gl, ref[1]), (ga, ref[2])):
                np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-11)
        else:
            keep = ~kink_samples(dict(loc=loc, shapes=shapes))
            close32(out, ref_out); close32(gv, ref[0]); close32(ga, ref[2]); close32(gl[keep], ref[1][keep])


# ---------------------------------------------------------------------------------------------
# the reference's own test recipe (models/ops/test.py): autograd Function + gradcheck in float64
# ---------------------------------------------------------------------------------------------
def _testpy_inputs(D, dtype):
    N, M, Lq, L, P = 1, 2, 2, 2, 2
    shapes = torch.as_tensor([(6, 4), (3, 2)], dtype=torch.long, device=DEV)
    starts = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    S = int(shapes.prod(1).sum())
    torch.manual_seed(3)
    value = (torch.rand(N, S, M, D, device=DEV) * 0.01).to(dtype)
    loc = torch.rand(N, Lq, M, L, P, 2, device=DEV).to(dtype)
    aw = torch.rand(N, Lq, M, L, P, device=DEV) + 1e-5
    aw = (aw / aw.sum(-1, keepdim=True).sum(-2, keepdim=True)).to(dtype)
    return value, shapes, starts, loc, aw


@pytest.mark.parametrize("D", [30, 32, 64, 71, 1025])
def test_reference_gradcheck_recipe(D):
    # models/ops/test.py:67-82, channel counts of test.py:89-90 (2048 / 3096: the same generic kernel, tests/test_zz_round6_gpu.py)
    value, shapes, starts, loc, aw = _testpy_inputs(D, torch.float64)
    value.requires_grad_(True); loc.requires_grad_(True); aw.requires_grad_(True)
    assert torch.autograd.gradcheck(msda.MSDeformAttnFunction.apply, (value, shapes, starts, loc, aw, 2))


def test_autograd_function_returns_grads_in_input_dtypes():
    rng = np.random.default_rng(0)
    value, shapes, starts, loc, aw, go = random_problem(rng, 2, PYRAMID, 8, 32, 20, 4)
    for vdt, adt in ((torch.float32, torch.float32), (torch.bfloat16, torch.float32),
                     (torch.bfloat16, torch.bfloat16)):
        v = _to_dev(value, vdt).requires_grad_(True)
        l = _to_dev(loc, adt).requires_grad_(True)
        a = _to_dev(aw, adt).requires_grad_(True)
        out = msda.MSDeformAttnFunction.apply(v, _to_dev(shapes), _to_dev(starts), l, a, 64)
        assert out.dtype == vdt and out.shape == (2, 20, 256)
        out.backward(_to_dev(go, vdt))
        assert v.grad.dtype == vdt and l.grad.dtype == adt and a.grad.dtype == adt
        assert torch.isfinite(v.grad.float()).all()


# ---------------------------------------------------------------------------------------------
# edge cases: empty / ragged / out-of-range inputs, argument errors
# ---------------------------------------------------------------------------------------------
def test_empty_queries_and_empty_batch():
    shapes = torch.tensor([[4, 5], [2, 3], [1, 2], [1, 1]], dtype=torch.long, device=DEV)
    starts = torch.tensor([0, 20, 26, 28], dtype=torch.long, device=DEV)
    value = torch.randn(2, 29, 8, 32, device=DEV)
    loc = torch.rand(2, 0, 8, 4, 4, 2, device=DEV)
    aw = torch.rand(2, 0, 8, 4, 4, device=DEV)
    out = msda.ms_deform_attn_forward(value, shapes, starts, loc, aw, 64)
    assert out.shape == (2, 0, 256)
    gv, gl, ga = msda.ms_deform_attn_backward(value, shapes, starts, loc, aw, torch.zeros(2, 0, 256, device=DEV), 64)
    assert gv.shape == value.shape and float(gv.abs().sum()) == 0.0 and gl.numel() == 0 and ga.numel() == 0


def test_all_samples_outside_give_zero():
    rng = np.random.default_rng(1)
    value, shapes, starts, loc, aw, go = random_problem(rng, 1, PYRAMID, 8, 32, 9, 4)
    loc = loc * 0 + np.float32(7.5)
    for variant in variants_for(32, 4, 4, torch.float32, value.shape[1], loc.shape[1]):
        out, gv, gl, ga = run_hip(variant, torch.float32, value, shapes, starts, loc, aw, go)
        assert not out.any() and not gv.any() and not gl.any() and not ga.any()


def test_nan_location_is_skipped_like_reference():
    rng = np.random.default_rng(2)
    value, shapes, starts, loc, aw, go = random_problem(rng, 1, PYRAMID, 8, 32, 5, 4)
    loc[0, 1, 2, 1, 3, 0] = np.nan
    ref = O.forward(value, shapes, starts, loc, aw)
    for variant in variants_for(32, 4, 4, torch.float32, value.shape[1], loc.shape[1]):
        out = run_hip(variant, torch.float32, value, shapes, starts, loc, aw)[0]
        assert np.isfinite(out).all()
        close32(out, ref)


def test_ragged_query_counts_cover_partial_waves():
    rng = np.random.default_rng(3)
    for Lq in (1, 2, 3, 7, 15, 17, 63, 65):
        value, shapes, starts, loc, aw, go = random_problem(rng, 1, PYRAMID, 8, 32, Lq, 4)
        ref_out = O.forward(value, shapes, starts, loc, aw)
        ref = O.backward(value, shapes, starts, loc, aw, go)
        keep = ~kink_samples(dict(loc=loc, shapes=shapes))
        for variant in variants_for(32, 4, 4, torch.float32, value.shape[1], loc.shape[1]):
            out, gv, gl, ga = run_hip(variant, torch.float32, value, shapes, starts, loc, aw, go)
            close32(out, ref_out); close32(gv, ref[0]); close32(ga, ref[2]); close32(gl[keep], ref[1][keep])


def test_im2col_step_error_matches_reference():
    rng = np.random.default_rng(4)
    value, shapes, starts, loc, aw, go = random_problem(rng, 3, PYRAMID, 8, 32, 4, 4)
    args = [_to_dev(x) for x in (value, shapes, starts, loc, aw)]
    with pytest.raises(RuntimeError, match="must divide im2col_step"):
        msda.ms_deform_attn_forward(*args, 2)            # 3 % 2 != 0  (ms_deform_attn_cuda.cu:52)
    msda.ms_deform_attn_forward(*args, 64)               # min(batch, 64) = 3 -> fine


def test_non_contiguous_input_raises():
    rng = np.random.default_rng(4)
    value, shapes, starts, loc, aw, go = random_problem(rng, 2, PYRAMID, 8, 32, 4, 4)
    v = _to_dev(value).transpose(2, 3)
    with pytest.raises(RuntimeError, match="has to be contiguous"):
        msda.ms_deform_attn_forward(v, _to_dev(shapes), _to_dev(starts), _to_dev(loc), _to_dev(aw), 64)


def test_mismatched_operand_shapes_raise_instead_of_reading_out_of_bounds():
    """An un-broadcast sampling_loc ([1, Lq, ...] next to a batch of 2) used to be read out of bounds by the kernels (they
    take N from value): the wrapper now rejects it, as it rejects a grad_output of the wrong shape."""
    rng = np.random.default_rng(5)
    value, shapes, starts, loc, aw, go = random_problem(rng, 2, PYRAMID, 8, 32, 4, 4)
    dv, ds, dst, dl, da, dg = (_to_dev(t) for t in (value, shapes, starts, loc, aw, go))
    with pytest.raises(RuntimeError, match="do not match"):
        msda.ms_deform_attn_forward(dv, ds, dst, dl[:1].contiguous(), da, 64)
    with pytest.raises(RuntimeError, match="do not match"):
        msda.ms_deform_attn_forward(dv, ds, dst, dl, da[:, :-1].contiguous(), 64)
    with pytest.raises(RuntimeError, match="grad_output"):
        msda.ms_deform_attn_backward(dv, ds, dst, dl, da, dg[:, :-1].contiguous(), 64)


# ---------------------------------------------------------------------------------------------
# full-size (BASELINE config 2: N=4, 800x1333 pyramid) through size-independent properties
# ---------------------------------------------------------------------------------------------
FULL = [(100, 167), (50, 84), (25, 42), (13, 21)]


def _full_inputs(dtype, Lq=None, seed=0):
    g = torch.Generator(device=DEV).manual_seed(seed)
    shapes = torch.tensor(FULL, dtype=torch.long, device=DEV)
    starts = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    S = int(shapes.prod(1).sum())
    N, M, D, L, P = 4, 8, 32, 4, 4
    Lq = S if Lq is None else Lq
    value = torch.randn(N, S, M, D, device=DEV, generator=g).to(dtype)
    loc = torch.rand(N, Lq, M, L, P, 2, device=DEV, generator=g) * 1.1 - 0.05
    aw = torch.softmax(torch.randn(N, Lq, M, L * P, device=DEV, generator=g), -1).view(N, Lq, M, L, P)
    return value, shapes, starts, loc, aw


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_full_size_variants_agree_and_constant_field_is_reproduced(dtype):
    value, shapes, starts, loc, aw = _full_inputs(dtype)
    outs = {}
    for variant in variants_for(32, 4, 4, dtype, value.shape[1], loc.shape[1]):
        msda.set_variant(*variant)
        outs[variant] = msda.ms_deform_attn_forward(value, shapes, starts, loc, aw, 64).float()
    msda.set_variant("auto")
    base = outs[("generic", "generic")]
    tol = 2e-2 if dtype == torch.bfloat16 else 2e-5
    for k, o in outs.items():
        assert (o - base).abs().max().item() <= tol * max(1.0, base.abs().max().item()), k
    # partition of unity: samples whose 4 corners are all inside return the constant of a constant
    # field (coarsest level is 13x21: pixel coordinate >= 0 needs loc >= 0.5/13)
    loc_in = loc.clamp(0.04, 0.96)
    ones = torch.ones_like(value)
    o = msda.ms_deform_attn_forward(ones, shapes, starts, loc_in, aw, 64).float()
    assert (o - 1).abs().max().item() < (1e-2 if dtype == torch.bfloat16 else 1e-5)


def test_full_size_linearity_and_adjoint_identity():
    value, shapes, starts, loc, aw = _full_inputs(torch.float32, seed=1)
    v2 = torch.randn_like(value)
    f = lambda v: msda.ms_deform_attn_forward(v, shapes, starts, loc, aw, 64)
    lhs = f(value * 0.5 + v2 * 2.0)
    rhs = f(value) * 0.5 + f(v2) * 2.0
    assert (lhs - rhs).abs().max().item() < 1e-4 * rhs.abs().max().item()
    # <grad_out, J v> == <J^T grad_out, v>: forward and grad_value are adjoint linear maps
    go = torch.randn_like(lhs)
    gv, gl, ga = msda.ms_deform_attn_backward(value, shapes, starts, loc, aw, go, 64)
    a = (go.double() * f(v2).double()).sum().item()
    b = (gv.double() * v2.double()).sum().item()
    assert abs(a - b) <= 1e-4 * max(abs(a), abs(b), 1.0)
    # grad_attn_weight is the per-sample bilinear value contracted with grad_out: summing
    # aw * g_aw over samples gives <grad_out, out>
    c = (ga.double() * aw.double()).sum().item()
    d = (go.double() * f(value).double()).sum().item()
    assert abs(c - d) <= 1e-4 * max(abs(c), abs(d), 1.0)


def test_full_size_decoder_shape_variants_agree():
    value, shapes, starts, loc, aw = _full_inputs(torch.float32, Lq=300, seed=2)
    go = torch.randn(4, 300, 256, device=DEV)
    res = {}
    for variant in variants_for(32, 4, 4, torch.float32, value.shape[1], loc.shape[1]):
        msda.set_variant(*variant)
        res[variant] = msda.ms_deform_attn_backward(value, shapes, starts, loc, aw, go, 64)
    msda.set_variant("auto")
    for k, r in res.items():
        for x, y in zip(r, res[("generic", "generic")]):
            assert (x - y).abs().max().item() <= 1e-4 * max(1.0, y.abs().max().item()), k


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("mode", ["model", "uniform", "far"])
def test_full_size_encoder_backward_variants_agree(dtype, mode):
    """Encoder shape at BASELINE batch 4: the LDS-window backward against the generic kernel, on
    model-like locations (windows fit), uniform locations (windows overflow -> global-atomic
    fallback for most corners) and far offsets (every window is clipped)."""
    from tools.msda_inputs import make_inputs
    inp = make_inputs(4, mode="uniform" if mode == "uniform" else "model", dtype=dtype, device=DEV, seed=7)
    loc = inp["loc"]
    if mode == "far":
        loc = (loc + 0.17).contiguous()
    a = (inp["value"], inp["shapes"], inp["starts"], loc, inp["aw"], inp["grad_out"])
    res = {}
    for variant in variants_for(32, 4, 4, dtype, inp["dims"][1], inp["dims"][5]):
        if variant[1] == "quad":
            continue                      # 34 ms of scattered global atomics; covered at small sizes
        msda.set_variant(*variant)
        res[variant] = [t.float() for t in msda.ms_deform_attn_backward(*a, 64)]
    msda.set_variant("auto")
    base = res[("generic", "generic")]
    tol = 4e-3 if dtype == torch.bfloat16 else 1e-4      # grad_value returns rounded to bf16
    for k, r in res.items():
        for name, x, y in zip(("g_value", "g_loc", "g_aw"), r, base):
            err = (x - y).abs().max().item() / max(1e-6, y.abs().max().item())
            assert err <= (tol if name == "g_value" else 1e-4), (k, name, err)


# ---------------------------------------------------------------------------------------------
# destination-stationary backward (msda_dest.hip): determinism, full-size oracle comparison
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_dest_backward_is_bitwise_repeatable(dtype):
    """No float atomics and a fixed summation order: two runs give identical bits (the reference's
    atomicAdd scatter, ms_deform_im2col_cuda.cuh:122-158, and the window variant do not)."""
    from tools.msda_inputs import make_inputs
    inp = make_inputs(2, mode="model", dtype=dtype, device=DEV, seed=11)
    a = (inp["value"], inp["shapes"], inp["starts"], inp["loc"], inp["aw"], inp["grad_out"])
    msda.set_variant("quad", "dest")
    try:
        first = msda.ms_deform_attn_backward(*a, 64)
        for _ in range(3):
            again = msda.ms_deform_attn_backward(*a, 64)
            for x, y in zip(first, again):
                assert torch.equal(x, y)
    finally:
        msda.set_variant("auto")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("Lq", [150, 300, 37])
def test_sparse_backward_for_few_queries_is_repeatable_and_matches_the_generic_kernel(dtype, Lq):
    """Decoder shapes (msda_sparse.hip: one workgroup per (image, head, level), records counting-sorted by pixel in
    LDS, runs summed in record order): bit-for-bit repeatable, grad_value equal to the generic kernel's to float32 /
    bfloat16 rounding, also when hundreds of samples fall on ONE pixel."""
    from tools.msda_inputs import make_inputs
    inp = make_inputs(4, Lq=Lq, mode="decoder", dtype=dtype, device=DEV, seed=5)
    loc = inp["loc"].clone()
    loc[0, : Lq // 2, 0, 3] = 0.5                       # half of image 0's queries: head 0, level 3, all points on one spot
    a = (inp["value"], inp["shapes"], inp["starts"], loc, inp["aw"], inp["grad_out"])
    msda.set_variant("quad", "dest")
    try:
        first = msda.ms_deform_attn_backward(*a, 64)
        again = msda.ms_deform_attn_backward(*a, 64)
        msda.set_variant("generic", "generic")
        ref = msda.ms_deform_attn_backward(*a, 64)
    finally:
        msda.set_variant("auto")
    torch.cuda.synchronize()
    for x, y in zip(first, again):
        assert torch.equal(x, y)
    gv, rv = first[0].float(), ref[0].float()
    tol = 2e-2 if dtype == torch.bfloat16 else 1e-4
    assert float((gv - rv).abs().max()) <= tol * float(rv.abs().max())
    assert int((gv != 0).any(-1).sum()) == int((rv != 0).any(-1).sum()) or dtype == torch.bfloat16


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_full_size_single_image_vs_oracle(dtype):
    """One 800x1333 image (S = Lq = 22223) through the product kernels (auto: quad forward, K1 + destination-
    stationary backward) against the CPU oracle -- ties the full-size fast path to the oracle directly."""
    from tools.msda_inputs import make_inputs
    inp = make_inputs(1, mode="model", dtype=dtype, device=DEV, seed=5)
    value = inp["value"].float().cpu().numpy()
    go = inp["grad_out"].float().cpu().numpy()
    shapes = inp["shapes"].cpu().numpy()
    starts = inp["starts"].cpu().numpy()
    loc = inp["loc"].cpu().numpy()
    aw = inp["aw"].cpu().numpy()
    ref_out = O.forward(value, shapes, starts, loc, aw, omp=True)
    ref_gv, ref_gl, ref_ga = O.backward(value, shapes, starts, loc, aw, go, omp=True)
    out = msda.ms_deform_attn_forward(inp["value"], inp["shapes"], inp["starts"], inp["loc"], inp["aw"], 64)
    gv, gl, ga = msda.ms_deform_attn_backward(inp["value"], inp["shapes"], inp["starts"], inp["loc"], inp["aw"],
                                              inp["grad_out"], 64)
    out, gv = out.float().cpu().numpy(), gv.float().cpu().numpy()
    if dtype == torch.bfloat16:
        np.testing.assert_allclose(out, ref_out, rtol=2 ** -7, atol=2 ** -7 * float(np.abs(ref_out).max()))
        np.testing.assert_allclose(gv, ref_gv, rtol=2 ** -7, atol=2 ** -7 * float(np.abs(ref_gv).max()))
    else:
        close32(out, ref_out)
        close32(gv, ref_gv)
    close32(ga.cpu().numpy(), ref_ga)
    keep = ~kink_samples(dict(loc=loc, shapes=shapes))
    close32(gl.cpu().numpy()[keep], ref_gl[keep])


@pytest.mark.parametrize("N", [2, 8])
def test_full_size_other_batch_shapes(N):
    """BASELINE configs 3 (batch 8 per GPU) and 4 (batch 2 per GPU) at the 800x1333 pyramid: product kernels against
    the generic kernel, plus the adjoint identity <grad_out, J v> == <J^T grad_out, v> (size-independent)."""
    from tools.msda_inputs import make_inputs
    inp = make_inputs(N, mode="model", dtype=torch.bfloat16, device=DEV, seed=20 + N)
    a = (inp["value"], inp["shapes"], inp["starts"], inp["loc"], inp["aw"])
    out = msda.ms_deform_attn_forward(*a, 64).float()
    gv, gl, ga = [t.float() for t in msda.ms_deform_attn_backward(*a, inp["grad_out"], 64)]
    msda.set_variant("generic")
    try:
        out_g = msda.ms_deform_attn_forward(*a, 64).float()
        gv_g, gl_g, ga_g = [t.float() for t in msda.ms_deform_attn_backward(*a, inp["grad_out"], 64)]
    finally:
        msda.set_variant("auto")
    for name, x, y, tol in (("out", out, out_g, 2e-2), ("g_value", gv, gv_g, 4e-3), ("g_loc", gl, gl_g, 1e-4),
                            ("g_aw", ga, ga_g, 1e-4)):
        err = (x - y).abs().max().item() / max(1e-6, y.abs().max().item())
        assert err <= tol, (name, err)
    v2 = torch.randn_like(inp["value"])
    lhs = (inp["grad_out"].double() * msda.ms_deform_attn_forward(v2, *a[1:], 64).double()).sum().item()
    rhs = (gv.double() * v2.double()).sum().item()
    assert abs(lhs - rhs) <= 2e-2 * max(abs(lhs), abs(rhs), 1.0)          # bf16 outputs on both sides


def test_reference_test_recipe_through_the_compat_shim():
    """models/ops/test.py:25-64 (N1 M2 D2 Lq2 L2 P2, shapes (6,4),(3,2), value = rand * 0.01), run through the
    import paths unmodified reference code uses: `MultiScaleDeformableAttention` and
    `models.ops.functions.MSDeformAttnFunction`, against the oracle in double and float."""
    import importlib
    import sys
    import rlipv2_amd.compat as compat
    saved = {k: v for k, v in sys.modules.items() if k == "MultiScaleDeformableAttention" or k.startswith("models")}
    try:
        for k in saved:
            del sys.modules[k]
        compat.install()
        F = importlib.import_module("models.ops.functions").MSDeformAttnFunction
        MSDA = importlib.import_module("MultiScaleDeformableAttention")
        torch.manual_seed(3)
        N, M, D, Lq, L, P = 1, 2, 2, 2, 2, 2
        shapes = torch.as_tensor([(6, 4), (3, 2)], dtype=torch.long, device=DEV)
        starts = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
        S = int(shapes.prod(1).sum())
        value = torch.rand(N, S, M, D, device=DEV) * 0.01
        loc = torch.rand(N, Lq, M, L, P, 2, device=DEV)
        aw = torch.rand(N, Lq, M, L, P, device=DEV) + 1e-5
        aw /= aw.sum(-1, keepdim=True).sum(-2, keepdim=True)
        for dt, tol in ((torch.float64, dict(rtol=1e-5, atol=1e-8)), (torch.float32, dict(rtol=1e-2, atol=1e-3))):
            ref = O.forward_numpy(value.double().cpu().numpy(), shapes.cpu().numpy(), starts.cpu().numpy(),
                                  loc.double().cpu().numpy(), aw.double().cpu().numpy())
            got = F.apply(value.to(dt), shapes, starts, loc.to(dt), aw.to(dt), 2)
            np.testing.assert_allclose(got.double().cpu().numpy(), ref, **tol)           # test.py:44 / :60
            raw = MSDA.ms_deform_attn_forward(value.to(dt), shapes, starts, loc.to(dt), aw.to(dt), 2)
            assert torch.equal(raw, got)
    finally:
        for k in [k for k in sys.modules if k == "MultiScaleDeformableAttention" or k.startswith("models")]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_shapes_that_do_not_sum_to_len_in_are_rejected():
    """reference ms_deform_attn.py:96 asserts sum(H*W) == Len_in; the op checks the host copy of the shapes."""
    rng = np.random.default_rng(9)
    value, shapes, starts, loc, aw, go = random_problem(rng, 1, [(8, 8), (4, 4), (2, 2), (1, 1)], 8, 32, 10, 4)
    bad = shapes.copy()
    bad[0, 0] = 9
    v, l, a, g = _to_dev(value), _to_dev(loc), _to_dev(aw), _to_dev(go)
    with pytest.raises(RuntimeError, match=r"sum\(H\*W\)"):
        msda.ms_deform_attn_backward(v, _to_dev(bad), _to_dev(starts), l, a, g, 64)


# ---- fused sampling geometry + sampling (msda_fused_forward / msda_fused_backward_ws) -----------------------------------
def _fused_inputs(dtype, refdim, N=2, Lq=None, pyramid=((25, 34), (13, 17), (7, 9), (4, 5)), seed=3):
    """Raw projection rows + reference points of a small encoder-like (Lq = S, 2-d) or decoder-like (4-d) call."""
    g = torch.Generator(device=DEV).manual_seed(seed)
    shapes = torch.tensor(pyramid, dtype=torch.long, device=DEV)
    starts = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    S = int(shapes.prod(1).sum())
    Lq = S if Lq is None else Lq
    M, D, L, P = 8, 32, 4, 4
    value = (torch.randn(N, S, M, D, device=DEV, generator=g) * 0.5).to(dtype)
    qproj = torch.randn(N, Lq, M * L * P * 3, device=DEV, generator=g)
    qproj[..., :M * L * P * 2] *= 3.0                                   # offsets of a few pixels / a fraction of the box
    qproj = qproj.to(dtype)
    if refdim == 2:
        ref = torch.rand(N, Lq, L, 2, device=DEV, generator=g)
    else:
        ref = torch.cat([torch.rand(N, Lq, L, 2, device=DEV, generator=g) * 0.8 + 0.1,
                         torch.rand(N, Lq, L, 2, device=DEV, generator=g) * 0.4 + 0.05], -1)
    grad_out = torch.randn(N, Lq, M * D, device=DEV, generator=g).to(dtype)
    msda.attach_host_shapes(shapes, pyramid)
    return value, shapes, starts, qproj, ref, grad_out, (M, L, P)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("refdim,Lq", [(2, None), (4, 150), (2, 37)])
def test_fused_geometry_route_equals_the_two_step_route(dtype, refdim, Lq):
    """FusedMSDeformAttnFunction == MSDeformAttnFunction(SamplingGeometryFunction(.)): the same geometry instructions
    (msda_geometry.h) and the same gather / scatter kernels underneath, so the comparison is exact for bfloat16 (same
    forward kernel structure) and to float32 rounding where the two-step route picks another forward kernel."""
    value, shapes, starts, qproj, ref, grad_out, (M, L, P) = _fused_inputs(dtype, refdim, Lq=Lq)
    res = []
    for fused in (True, False):
        v = value.clone().requires_grad_(True)
        q = qproj.clone().requires_grad_(True)
        if fused:
            assert msda.fused_supported(v, shapes, ref, q.shape[1], L, P, True)
            out = msda.FusedMSDeformAttnFunction.apply(v, shapes, starts, q, ref, 64)
        else:
            loc, aw = msda.SamplingGeometryFunction.apply(q, ref, shapes, M, L, P)
            out = msda.MSDeformAttnFunction.apply(v, shapes, starts, loc, aw, 64)
        out.backward(grad_out)
        res.append((out.detach().float(), v.grad.float(), q.grad.float()))
    torch.cuda.synchronize()
    (o1, gv1, gq1), (o2, gv2, gq2) = res
    if dtype == torch.bfloat16:
        assert torch.equal(o1, o2)
    else:
        assert torch.allclose(o1, o2, rtol=1e-5, atol=1e-6 * float(o2.abs().max()))
    assert torch.equal(gv1, gv2)            # same destination-stationary pass on the same locations / weights
    assert torch.equal(gq1, gq2)            # same K1 arithmetic, same geometry backward


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fused_geometry_route_vs_oracle(dtype):
    """The fused route against the CPU oracle fed with locations / weights computed in float64 from the same rows."""
    value, shapes, starts, qproj, ref, grad_out, (M, L, P) = _fused_inputs(dtype, 2, N=1)
    N, Lq = qproj.shape[:2]
    v = value.clone().requires_grad_(True)
    q = qproj.clone().requires_grad_(True)
    out = msda.FusedMSDeformAttnFunction.apply(v, shapes, starts, q, ref, 64)
    out.backward(grad_out)
    torch.cuda.synchronize()
    # float64 restatement of ms_deform_attn.py:101-109 on the CPU
    qd = qproj.double().cpu()
    off = qd[..., :M * L * P * 2].reshape(N, Lq, M, L, P, 2)
    aw = torch.softmax(qd[..., M * L * P * 2:].reshape(N, Lq, M, L * P), -1).reshape(N, Lq, M, L, P)
    norm = torch.stack([shapes[:, 1], shapes[:, 0]], -1).double().cpu()
    loc = ref.double().cpu()[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
    a = [value.double().cpu().numpy(), shapes.cpu().numpy(), starts.cpu().numpy(), loc.numpy(), aw.numpy()]
    ref_out = O.forward(*a)
    ref_gv, ref_gl, ref_ga = O.backward(*a, grad_out.double().cpu().numpy())
    tol = 2 ** -7 if dtype == torch.bfloat16 else 1e-4
    got = out.detach().float().cpu().numpy()
    assert np.abs(got - ref_out).max() <= tol * np.abs(ref_out).max()
    gv = v.grad.float().cpu().numpy()
    assert np.abs(gv - ref_gv).max() <= tol * np.abs(ref_gv).max()
    # chain rule of the geometry in float64: offsets scale by 1 / (W, H); softmax backward
    g_off = torch.from_numpy(ref_gl) / norm[None, None, None, :, None, :]
    ga = torch.from_numpy(ref_ga)
    g_logit = aw * (ga - (aw * ga).sum((-1, -2), keepdim=True))
    ref_gq = torch.cat([g_off.reshape(N, Lq, -1), g_logit.reshape(N, Lq, -1)], -1).numpy()
    gq = q.grad.float().cpu().numpy()
    # the location gradient jumps where a sample crosses a pixel centre: compare away from those samples
    keep = np.broadcast_to(~kink_samples({"loc": a[3], "shapes": a[1]}, 1e-3)[..., None], ref_gl.shape).reshape(N, Lq, -1)
    keep = np.concatenate([keep, np.ones((N, Lq, M * L * P), dtype=bool)], -1)
    err = np.abs(gq - ref_gq)[keep].max()
    assert err <= (2 ** -6 if dtype == torch.bfloat16 else 1e-3) * np.abs(ref_gq).max()


def test_coarse_lds_forward_is_bit_identical_to_the_direct_gather_forward():
    """quad_forward_coarse_kernel (rows of the trailing levels in LDS; an explicit variant, faster than the direct
    gathers on spread-out locations, slower on model-like ones) does the same arithmetic in the same order as quad_forward_kernel: bit-identical outputs,
    also for locations outside [0, 1], NaN locations, and a pyramid whose two coarse levels do NOT fit the LDS budget
    together (only the last one is staged then)."""
    for pyramid in (FULL, ((60, 70), (40, 50), (34, 40), (20, 24))):     # second: 1 360 + 480 rows: only level 3 staged
        g = torch.Generator(device=DEV).manual_seed(5)
        shapes = torch.tensor(pyramid, dtype=torch.long, device=DEV)
        starts = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
        S = int(shapes.prod(1).sum())
        N, M, D, L, P, Lq = 2, 8, 32, 4, 4, 5003
        value = torch.randn(N, S, M, D, device=DEV, generator=g).bfloat16()
        loc = torch.rand(N, Lq, M, L, P, 2, device=DEV, generator=g) * 1.3 - 0.15
        loc[0, 17, 3, 2, 1, 0] = float("nan")
        aw = torch.softmax(torch.randn(N, Lq, M, L * P, device=DEV, generator=g), -1).view(N, Lq, M, L, P)
        outs = []
        for variant in ("quad", "coarse"):
            msda.set_variant(variant, "auto")
            outs.append(msda.ms_deform_attn_forward(value, shapes, starts, loc, aw, 64))
        msda.set_variant("auto")
        torch.cuda.synchronize()
        assert torch.equal(outs[0], outs[1])
