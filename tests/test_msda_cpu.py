"""The product's CPU arm of the op (include/rlipv2_msda_cpu.h, rlipv2_amd/csrc/msda_cpu.cpp, dispatched by rlipv2_amd/msda.py for
CPU tensors) against the reference-generated goldens (tests/golden/make_msda_golden.py: the reference's own
`ms_deform_attn_core_pytorch` + autograd, models/ops/functions/ms_deform_attn_func.py:45-65).  The oracle is not involved:
this is product code checked against the reference's vectors directly.  Tolerances as for the oracle: float64 -> allclose
defaults (the reference's bar, models/ops/test.py:44), float32 -> rtol 1e-4 / 1e-5 x max (reference bar 1e-2 / 1e-3, test.py:60)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import boundary_samples, kink_samples, load_golden
from rlipv2_amd import _lib, msda
from rlipv2_amd.msda import MSDeformAttnFunction

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tensors(g, dtype):
    return (torch.from_numpy(g["value"]).to(dtype), torch.from_numpy(g["shapes"]), torch.from_numpy(g["starts"]),
            torch.from_numpy(g["loc"]).to(dtype), torch.from_numpy(g["aw"]).to(dtype))


def test_exports_equal_the_header():
    text = open(os.path.join(ROOT, "include", "rlipv2_msda_cpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = set(re.findall(r"\b(msda_[a-z_0-9]+)\s*\(", text))
    assert declared == set(_lib.CPU_EXPORTS)
    L = _lib.cpu_lib()
    for name in _lib.CPU_EXPORTS:
        getattr(L, name)
    assert L.msda_cpu_abi_version() == 1


def test_f64_matches_reference(msda_golden):
    g = msda_golden
    value, shapes, starts, loc, aw = _tensors(g, torch.float64)
    value.requires_grad_(True), loc.requires_grad_(True), aw.requires_grad_(True)
    out = MSDeformAttnFunction.apply(value, shapes, starts, loc, aw, 64)
    np.testing.assert_allclose(out.detach().numpy(), g["out_f64"], rtol=1e-5, atol=1e-8)
    out.backward(torch.from_numpy(g["grad_out"]).double())
    np.testing.assert_allclose(value.grad.numpy(), g["g_value_f64"], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(aw.grad.numpy(), g["g_aw_f64"], rtol=1e-5, atol=1e-8)
    keep = ~boundary_samples(g)                  # conftest.boundary_samples: where the reference's two implementations differ
    np.testing.assert_allclose(loc.grad.numpy()[keep], g["g_loc_f64"][keep], rtol=1e-5, atol=1e-8)


def test_f32_matches_reference_f32(msda_golden):
    g = msda_golden
    value, shapes, starts, loc, aw = _tensors(g, torch.float32)
    out = msda.ms_deform_attn_forward(value, shapes, starts, loc, aw, 64)
    gv, gl, ga = msda.ms_deform_attn_backward(value, shapes, starts, loc, aw, torch.from_numpy(g["grad_out"]), 64)
    scale = lambda ref: 1e-5 * max(1.0, float(np.abs(ref).max()))
    np.testing.assert_allclose(out.numpy(), g["out_f32"], rtol=1e-4, atol=scale(g["out_f32"]))
    np.testing.assert_allclose(gv.numpy(), g["g_value_f32"], rtol=1e-4, atol=scale(g["g_value_f32"]))
    keep = ~kink_samples(g)
    np.testing.assert_allclose(gl.numpy()[keep], g["g_loc_f32"][keep], rtol=1e-4, atol=scale(g["g_loc_f32"]))
    np.testing.assert_allclose(ga.numpy(), g["g_aw_f32"], rtol=1e-4, atol=scale(g["g_aw_f32"]))


def test_reference_gradcheck_recipe_on_the_cpu():
    """models/ops/test.py:64-86 (check_gradient_numerical), float64, the reference's shapes -- on the product's CPU op."""
    torch.manual_seed(3)
    N, M, D, Lq, L, P = 1, 2, 2, 2, 2, 2
    shapes = torch.as_tensor([(6, 4), (3, 2)], dtype=torch.long)
    starts = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    S = int(shapes.prod(1).sum())
    for D in (2, 30, 32, 71, 1025, 2048, 3096):                      # models/ops/test.py:89: 30, 32, 64, 71, 1025, 2048, 3096
        value = (torch.rand(N, S, M, D, dtype=torch.float64) * 0.01).requires_grad_(True)
        loc = torch.rand(N, Lq, M, L, P, 2, dtype=torch.float64).requires_grad_(True)
        aw = torch.rand(N, Lq, M, L, P, dtype=torch.float64) + 1e-5
        aw = (aw / aw.sum(-1, keepdim=True).sum(-2, keepdim=True)).requires_grad_(True)
        # full Jacobians up to 71 channels; the large counts in fast mode (directional derivatives: the full Jacobian of `value`
        # at 3096 channels is 18 GB)
        assert torch.autograd.gradcheck(MSDeformAttnFunction.apply, (value, shapes, starts, loc, aw, 2), fast_mode=D > 71)


def test_bf16_cpu_tensors_are_widened():
    g = load_golden("model_dec")
    value, shapes, starts, loc, aw = _tensors(g, torch.float32)
    vb = value.to(torch.bfloat16)
    out = msda.ms_deform_attn_forward(vb, shapes, starts, loc, aw, 64)
    ref = msda.ms_deform_attn_forward(vb.float(), shapes, starts, loc, aw, 64)
    assert out.dtype == torch.bfloat16 and torch.equal(out, ref.to(torch.bfloat16))
    gv, gl, ga = msda.ms_deform_attn_backward(vb, shapes, starts, loc, aw, torch.from_numpy(g["grad_out"]).to(torch.bfloat16), 64)
    assert gv.dtype == torch.bfloat16 and gl.dtype == torch.float32 and ga.dtype == torch.float32


def test_results_do_not_depend_on_the_thread_count():
    """every output element has one writer and a fixed summation order (backward: one task per (image, head, level))"""
    g = load_golden("pyr_enc")
    value, shapes, starts, loc, aw = _tensors(g, torch.float32)
    go = torch.from_numpy(g["grad_out"])
    omp = ctypes.CDLL("libgomp.so.1")
    res = []
    for n in (1, 5):
        omp.omp_set_num_threads(n)
        res.append([msda.ms_deform_attn_forward(value, shapes, starts, loc, aw, 64),
                    *msda.ms_deform_attn_backward(value, shapes, starts, loc, aw, go, 64)])
    omp.omp_set_num_threads(os.cpu_count() or 1)
    for a, b in zip(*res):
        assert torch.equal(a, b)


def test_edge_cases():
    """empty query set, empty batch, dropped samples (.cuh:285-288), NaN locations, value rows outside every level"""
    shapes = torch.tensor([[4, 5]])
    starts = torch.tensor([0])
    value = torch.ones(1, 20, 1, 3)
    aw = torch.ones(1, 1, 1, 1, 2)
    loc = torch.tensor([[-0.5 / 5, 0.5], [0.5, 4.5 / 4]]).reshape(1, 1, 1, 1, 2, 2)        # w_im == -1 ; h_im == H
    assert torch.all(msda.ms_deform_attn_forward(value, shapes, starts, loc, aw, 64) == 0)
    gv, gl, ga = msda.ms_deform_attn_backward(value, shapes, starts, loc, aw, torch.ones(1, 1, 3), 64)
    assert torch.all(gv == 0) and torch.all(gl == 0) and torch.all(ga == 0)
    loc2 = torch.tensor([[0.0, 0.0], [1.0, 1.0]]).reshape(1, 1, 1, 1, 2, 2)                # image corners: weight 1/4 each
    torch.testing.assert_close(msda.ms_deform_attn_forward(value, shapes, starts, loc2, aw, 64), torch.full((1, 1, 3), 0.5))
    nan = torch.full((1, 1, 1, 1, 2, 2), float("nan"))
    assert torch.all(msda.ms_deform_attn_forward(value, shapes, starts, nan, aw, 64) == 0)
    # no queries / no images
    out = msda.ms_deform_attn_forward(value, shapes, starts, torch.empty(1, 0, 1, 1, 2, 2), torch.empty(1, 0, 1, 1, 2), 64)
    assert out.shape == (1, 0, 3)
    out = msda.ms_deform_attn_forward(torch.empty(0, 20, 1, 3), shapes, starts, torch.empty(0, 4, 1, 1, 2, 2), torch.empty(0, 4, 1, 1, 2), 64)
    assert out.shape == (0, 4, 3)
    # S larger than the pyramid: the extra rows get a zero gradient; a level reaching outside value is refused
    big = torch.ones(1, 25, 1, 3)
    gv, _, _ = msda.ms_deform_attn_backward(big, shapes, starts, loc2, aw, torch.ones(1, 1, 3), 64)
    assert torch.all(gv[:, 20:] == 0) and float(gv.sum()) == pytest.approx(2 * 3 * 0.25)
    with pytest.raises(RuntimeError, match="does not lie inside"):
        msda.ms_deform_attn_forward(torch.ones(1, 19, 1, 3), shapes, starts, loc2, aw, 64)
    with pytest.raises(RuntimeError, match="im2col_step"):
        msda.ms_deform_attn_forward(torch.ones(3, 20, 1, 3), shapes, starts, loc2.expand(3, -1, -1, -1, -1, -1).contiguous(),
                                    aw.expand(3, -1, -1, -1, -1).contiguous(), 2)
    with pytest.raises(RuntimeError, match="contiguous"):
        msda.ms_deform_attn_forward(value.transpose(1, 3), shapes, starts, loc2, aw, 64)


def test_overlapping_levels_take_the_unowned_route():
    """level_start_index ranges that overlap (the reference's atomicAdd scatter tolerates any layout, .cuh:142-156): no task owns a
    grad_value row, so the backward walks all levels of an (image, head) in one task over a zeroed grad_value -- same numbers as
    the oracle (which scatters like the reference), for any thread count"""
    from oracle import msda_oracle as O
    rng = np.random.default_rng(5)
    shapes = np.array([[6, 4], [3, 2], [5, 3]], dtype=np.int64)
    starts = np.array([0, 20, 10], dtype=np.int64)                     # level 1 overlaps level 0's tail, level 2 overlaps both
    N, S, M, D, Lq, L, P = 2, 45, 3, 5, 7, 3, 2                        # (S = sum of H W, as the oracle insists; rows 26.. unused)
    value = rng.standard_normal((N, S, M, D))
    loc = rng.uniform(-0.1, 1.1, (N, Lq, M, L, P, 2))
    aw = rng.uniform(0, 1, (N, Lq, M, L, P))
    go = rng.standard_normal((N, Lq, M * D))
    t = lambda a: torch.from_numpy(a)
    ref_out = O.forward(value, shapes, starts, loc, aw)
    ref = O.backward(value, shapes, starts, loc, aw, go)
    omp = ctypes.CDLL("libgomp.so.1")
    try:
        for n in (1, 6):
            omp.omp_set_num_threads(n)
            out = msda.ms_deform_attn_forward(t(value), t(shapes), t(starts), t(loc), t(aw), 64)
            got = msda.ms_deform_attn_backward(t(value), t(shapes), t(starts), t(loc), t(aw), t(go), 64)
            np.testing.assert_allclose(out.numpy(), ref_out, rtol=1e-12, atol=1e-13)
            for a, b in zip(got, ref):
                np.testing.assert_allclose(a.numpy(), b, rtol=1e-11, atol=1e-12)
    finally:
        omp.omp_set_num_threads(os.cpu_count() or 1)


@pytest.mark.parametrize("nd,masked", [(4, True), (4, False), (2, True)])
def test_sample_then_project_is_the_same_function(nd, masked):
    """deform_attn.sample_then_project: the few-query cross-attention with sampling and value projection exchanged
    (MSDeformAttn._sampled_projection; reference arithmetic models/ops/modules/ms_deform_attn.py:98-118) against the standard
    order, float64 on the product's CPU op: output and the gradients of the query, the memory, the reference points and every
    parameter agree to rounding -- with a padding mask, with samples outside the image (zero padding: the bias term must carry
    the same bilinear coverage), 2-d and 4-d reference points."""
    from rlipv2_amd import deform_attn
    torch.manual_seed(7 + nd)
    pyr = [(12, 16), (6, 8), (3, 4), (2, 2)]
    shapes = torch.tensor(pyr)
    starts = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    S, N, Lq, C = int(shapes.prod(1).sum()), 2, 5, 64
    m = deform_attn.MSDeformAttn(C, 4, 8, 4).double()
    with torch.no_grad():
        m.sampling_offsets.weight.copy_(0.3 * torch.randn_like(m.sampling_offsets.weight))
        m.attention_weights.weight.copy_(0.5 * torch.randn_like(m.attention_weights.weight))
        m.value_proj.bias.copy_(torch.randn_like(m.value_proj.bias))
    q0, src0 = torch.randn(N, Lq, C, dtype=torch.float64), torch.randn(N, S, C, dtype=torch.float64)
    ref0 = torch.rand(N, Lq, 4, nd, dtype=torch.float64) * 1.2 - 0.1             # some reference points outside [0, 1]
    if nd == 4:
        ref0[..., 2:] = ref0[..., 2:].abs() * 0.5 + 0.05
    mask = (torch.rand(N, S) < 0.2) if masked else None
    go = torch.randn(N, Lq, C, dtype=torch.float64)
    res = {}
    for stp in (True, False):
        deform_attn.sample_then_project = stp
        try:
            for p in m.parameters():
                p.grad = None
            q, src, ref = (t.clone().requires_grad_(True) for t in (q0, src0, ref0))
            out = m(q, ref, src, shapes, starts, mask)
            out.backward(go)
            res[stp] = [out.detach(), q.grad, src.grad, ref.grad] + [p.grad for p in m.parameters()]
        finally:
            deform_attn.sample_then_project = False
    assert Lq * 8 <= deform_attn.SAMPLE_THEN_PROJECT_MAX_FRACTION * S         # (the route was eligible)
    for a, b in zip(res[True], res[False]):
        torch.testing.assert_close(a, b, rtol=1e-9, atol=1e-11)
    # many queries (the encoder's self-attention): the standard order is kept whatever the switch says
    deform_attn.sample_then_project = True
    try:
        big = torch.randn(N, S, C, dtype=torch.float64)
        refs = torch.rand(N, S, 4, 2, dtype=torch.float64)
        calls = []
        orig = deform_attn.MSDeformAttn._sampled_projection
        deform_attn.MSDeformAttn._sampled_projection = lambda self, *a: calls.append(1) or orig(self, *a)
        m(big, refs, src0, shapes, starts, None)
        assert not calls
    finally:
        deform_attn.sample_then_project = False
        deform_attn.MSDeformAttn._sampled_projection = orig


def test_full_size_properties_of_the_cpu_arm():
    """BASELINE config 2's encoder call on ONE image (800 x 1333 pyramid, S = Lq = 22 223, M 8, D 32, L = P = 4) through the CPU
    twins, checked by properties that need no reference at this size: linearity in `value`, the adjoint identity
    <grad_out, J v> = <J^T grad_out, v> between forward and the grad_value of backward, partition of unity (value = 1 and
    in-range samples -> out = sum of the attention weights = 1), and grad_attn_weight = the per-sample bilinear values dotted
    with grad_out (forward with a one-hot weight)."""
    torch.manual_seed(0)
    pyr = [(100, 167), (50, 84), (25, 42), (13, 21)]
    shapes = torch.tensor(pyr)
    starts = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    S, M, D = int(shapes.prod(1).sum()), 8, 32
    ref = torch.cat([torch.stack(torch.meshgrid((torch.arange(h) + 0.5) / h, (torch.arange(w) + 0.5) / w, indexing="ij"), -1)
                     .flip(-1).reshape(-1, 2) for h, w in pyr])                                  # (x, y) pixel centres
    off = torch.randn(1, S, M, 4, 4, 2) * 2.0 / torch.tensor([[w, h] for h, w in pyr], dtype=torch.float32)[None, None, None, :, None, :]
    loc = (ref[None, :, None, None, None, :] + off).contiguous()
    aw = torch.softmax(torch.randn(1, S, M, 16), -1).view(1, S, M, 4, 4).contiguous()
    v1, v2 = torch.randn(1, S, M, D), torch.randn(1, S, M, D)
    go = torch.randn(1, S, M * D)
    f = lambda v: msda.ms_deform_attn_forward(v, shapes, starts, loc, aw, 64)                      # noqa: E731
    o1, o2 = f(v1), f(v2)
    torch.testing.assert_close(f(2.0 * v1 - 0.5 * v2), 2.0 * o1 - 0.5 * o2, rtol=1e-4, atol=1e-4)
    gv, gl, ga = msda.ms_deform_attn_backward(v1, shapes, starts, loc, aw, go, 64)
    lhs, rhs = float((go.double() * o2.double()).sum()), float((gv.double() * v2.double()).sum())
    assert abs(lhs - rhs) <= 1e-4 * max(abs(lhs), abs(rhs), 1.0), (lhs, rhs)
    wh = torch.tensor([[w, h] for h, w in pyr], dtype=torch.float32)[None, None, None, :, None, :]
    px = loc * wh - 0.5                                              # all four corners inside the level: 0 <= px <= size - 1
    inside = ((px >= 0) & (px <= wh - 1)).all(-1).all(-1).all(-1)                               # [1, S, M]: every sample of the (query, head)
    ones = f(torch.ones(1, S, M, D)).view(1, S, M, D)
    assert int(inside.sum()) > S and float((ones[inside] - 1.0).abs().max()) < 1e-5
    # grad_attn_weight of sample (l, p) = <grad_out, bilinear value at that sample>: forward with the one-hot weight
    hot = torch.zeros_like(aw)
    hot[..., 2, 1] = 1.0
    sampled = msda.ms_deform_attn_forward(v1, shapes, starts, loc, hot, 64).view(1, S, M, D)
    want = (sampled * go.view(1, S, M, D)).sum(-1)
    torch.testing.assert_close(ga[..., 2, 1], want, rtol=1e-4, atol=1e-4)


def test_random_problems_against_the_oracle():
    """hypothesis-style sweep without the dependency on its shrinker: 40 seeded random problems (1-3 images, 1-5 levels of 1-9 pixels
    a side, 1-4 heads, 1-9 channels, 0-7 queries, 1-5 points; locations spread far beyond [0, 1], some exactly on pixel centres and on
    the -1 / size drop boundary) through the CPU twins against the oracle's float64 C restatement."""
    from oracle import msda_oracle as O
    rng = np.random.default_rng(2024)
    for case in range(40):
        N, L, M, D, Lq, P = (int(rng.integers(a, b + 1)) for a, b in ((1, 3), (1, 5), (1, 4), (1, 9), (0, 7), (1, 5)))
        pyr = rng.integers(1, 10, size=(L, 2)).astype(np.int64)
        starts = np.concatenate(([0], np.cumsum(pyr[:, 0] * pyr[:, 1])[:-1])).astype(np.int64)
        S = int((pyr[:, 0] * pyr[:, 1]).sum())
        loc = rng.uniform(-0.4, 1.4, size=(N, Lq, M, L, P, 2))
        if Lq:
            wh = np.stack([pyr[:, 1], pyr[:, 0]], -1)[None, None, None, :, None, :]
            snap = rng.random(loc.shape) < 0.15                        # exact pixel centres / the drop boundary
            k = rng.integers(-1, 11, size=loc.shape)
            loc = np.where(snap, (np.minimum(k, wh) + 0.5) / wh, loc)
        value = rng.standard_normal((N, S, M, D))
        aw = rng.random((N, Lq, M, L, P))
        go = rng.standard_normal((N, Lq, M * D))
        ref_out = O.forward(value, pyr, starts, loc, aw)
        ref_gv, ref_gl, ref_ga = O.backward(value, pyr, starts, loc, aw, go)
        t = [torch.from_numpy(np.ascontiguousarray(a)) for a in (value, pyr, starts, loc, aw)]
        out = msda.ms_deform_attn_forward(*t, 64)
        gv, gl, ga = msda.ms_deform_attn_backward(*t, torch.from_numpy(go), 64)
        np.testing.assert_allclose(out.numpy(), ref_out, rtol=1e-9, atol=1e-11, err_msg=f"case {case}")
        np.testing.assert_allclose(gv.numpy(), ref_gv, rtol=1e-9, atol=1e-11, err_msg=f"case {case}")
        np.testing.assert_allclose(ga.numpy(), ref_ga, rtol=1e-9, atol=1e-11, err_msg=f"case {case}")
        np.testing.assert_allclose(gl.numpy(), ref_gl, rtol=1e-9, atol=1e-10, err_msg=f"case {case}")
