"""GPU tests of the "records" route of the encoder backward (csrc/msda_cell_forward.inc EMIT, csrc/msda_cell_records.inc): device
code written in round 5 without a GPU, validated on the lane-level model only (tests/test_records_emulated.py).  An experiment
behind msda.records_route, OFF in the product -- sorted LAST in the GPU suite (tests/conftest.py: GPU_SUITE_ORDER), after the
bench contract test, so that a failure here cannot hide any evidence about the product path."""
import os
import sys

import pytest
import torch

# first_contact: each test in a child process with a timeout, outcome XPASS / XFAIL (tests/conftest.py) -- the route is OFF in the product
pytestmark = [pytest.mark.gpu, pytest.mark.first_contact(timeout=420)]
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def same_bits(a, b):
    return torch.equal(a.view(torch.int16 if a.element_size() == 2 else torch.int32), b.view(torch.int16 if b.element_size() == 2 else torch.int32))


def within_a_rounding(a, b, what):
    """The gradients that go through the float32 formulas of ms_deform_im2col_cuda.cuh:87-159.  On the lane-level model the two
    kernels are bit-equal (same source expressions, same host compiler); on the device the compiler contracts the expressions into
    FMAs per kernel -- even per sample position inside the product kernel (profiles/r05_records_route_static.txt) --, so the bar here
    is: bit-equal OR within one rounding of the output type (float32: any number of last-bit differences; bfloat16 rows: a float32
    last bit only shows where it crosses a bfloat16 rounding boundary, so few elements may differ)."""
    if same_bits(a, b):
        return
    a32, b32 = a.float(), b.float()
    lim = 2.0 ** -7 if a.dtype == torch.bfloat16 else 2e-5
    worst = float((a32 - b32).abs().max() / b32.abs().max().clamp_min(1e-30))
    share = float((a32 != b32).float().mean())
    assert worst <= lim and (share <= 0.05 or a.dtype != torch.bfloat16), \
        f"{what}: max difference {worst:.2e} of the maximum, {share:.2%} of the elements differ"


def _encoder_call(pyr, N, M, seed, refdim):
    """a bfloat16 encoder call in the module's operands: value, projection rows, reference points, grad_out"""
    from rlipv2_amd import msda
    g = torch.Generator().manual_seed(seed)
    shapes = torch.tensor(pyr, dtype=torch.int64, device=DEV)
    msda.attach_host_shapes(shapes, pyr)
    starts = torch.cat([shapes.new_zeros(1), (shapes[:, 0] * shapes[:, 1]).cumsum(0)[:-1]])
    S = sum(h * w for h, w in pyr)
    refp = torch.cat([torch.stack(torch.meshgrid((torch.arange(H) + 0.5) / H, (torch.arange(W) + 0.5) / W, indexing="ij")[::-1], -1).reshape(-1, 2)
                      for H, W in pyr])
    ref = refp[None, :, None, :].expand(N, S, 4, 2)
    if refdim == 4:
        ref = torch.cat([ref, torch.tensor([0.2, 0.15]).expand(N, S, 4, 2)], -1)
    value = (0.5 * torch.randn(N, S, M, 32, generator=g)).to(torch.bfloat16)
    qproj = torch.randn(N, S, M * 48, generator=g)
    qproj[..., :M * 32] *= 2.0                                       # offsets of a few pixels
    gout = torch.randn(N, S, M * 32, generator=g).to(torch.bfloat16)
    return (value.to(DEV), shapes, starts, qproj.to(torch.bfloat16).to(DEV), ref.contiguous().float().to(DEV), gout.to(DEV))


@pytest.mark.parametrize("refdim", [2, 4])
@pytest.mark.parametrize("pyr,N,M", [([(20, 27), (10, 14), (5, 7), (3, 4)], 1, 2), ([(100, 134), (50, 67), (25, 34), (13, 17)], 2, 8)])
def test_records_route_matches_the_product_kernels_bit_for_bit(pyr, N, M, refdim):
    """msda.records_route (cell_forward_kernel with EMIT + cell_records_backward_kernel, never run on hardware before this test)
    against the product route of the train step through the same autograd function: grad_value bit for bit, the gradient of the
    projection rows within a rounding (see within_a_rounding), both operand orders; the output to bfloat16 rounding (the two forward
    kernels sum in different orders); run-to-run bit-repeatable.  On the host model everything was bit-equal
    (tests/test_records_emulated.py)."""
    from rlipv2_amd import msda
    value0, shapes, starts, qproj0, ref, gout = _encoder_call(pyr, N, M, seed=11, refdim=refdim)

    def run(route, swap):
        msda.records_route, msda.records_swap = route, swap
        try:
            value, qproj = value0.clone().requires_grad_(True), qproj0.clone().requires_grad_(True)
            out = msda.FusedMSDeformAttnFunction.apply(value, shapes, starts, qproj, ref, 64)
            fwd = msda.last_variant["fwd"]
            out.backward(gout)
            torch.cuda.synchronize()
            return out.detach(), value.grad, qproj.grad, fwd, msda.last_variant["bwd"]
        finally:
            msda.records_route, msda.records_swap = False, True
    base = run(False, False)
    assert base[4] == "dest+geometry"
    for swap in (False, True):
        got = run(True, swap)
        assert got[3] == "cell+geometry+records" and got[4] == "records+geometry"
        assert float((got[0].float() - base[0].float()).abs().max()) <= 2.0 ** -6 * float(base[0].float().abs().max())
        assert same_bits(got[1], base[1]), "grad_value differs (same patch pass on the same operands: must be bit-equal)"
        within_a_rounding(got[2], base[2], "grad of the projection rows")
    # and twice the same bits (no atomics, fixed summation order)
    first, again = run(True, False), run(True, False)
    assert all(same_bits(x, y) for x, y in zip(first[:3], again[:3]))


def test_records_route_op_signature_against_the_oracle():
    """the C ABI of the route with the op's own operands (refdim 0: sampling_loc / attn_weight in, their gradients out) against
    the oracle, and bit-equal to msda_backward_ws; includes out-of-range, corner and NaN locations"""
    import ctypes

    import numpy as np

    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import msda_oracle as O
    from test_cell_forward_emulated import make_problem
    from rlipv2_amd import _lib
    L = _lib.lib()
    pyr, starts, S, value, loc, aw = make_problem([(25, 34), (13, 17), (7, 9), (4, 5)], 2, (2.0, 3.0, 3.0, 3.0), seed=7)
    M = 2
    gout = torch.randn(1, S, M * 32, generator=torch.Generator().manual_seed(3)).to(torch.bfloat16)
    t = lambda a, dt=None: torch.as_tensor(np.ascontiguousarray(a)).to(dt or torch.float32).to(DEV).contiguous()   # noqa: E731
    v, lo, a_, go = t(value, torch.bfloat16), t(loc), t(aw), gout.to(DEV)
    sh, st = t(pyr, torch.int64), t(starts, torch.int64)
    hs = (ctypes.c_int64 * 8)(*[int(x) for x in np.asarray(pyr).reshape(-1)])
    dims = (1, S, M, 32, 4, S, 4)
    BF16, FLAG = _lib.MSDA_BF16, _lib.FLAG_GRAD_VALUE_BF16
    rec_bytes = int(L.msda_records_bytes(BF16, hs, *dims))
    ws_bytes = int(L.msda_backward_workspace_bytes(BF16, hs, *dims))
    assert rec_bytes > 0 and ws_bytes > 0
    records = torch.full((rec_bytes,), 0xA5, dtype=torch.uint8, device=DEV)
    out = torch.zeros(1, S, M * 32, dtype=torch.bfloat16, device=DEV)
    stream = torch.cuda.current_stream().cuda_stream
    assert L.msda_records_forward(BF16, v.data_ptr(), sh.data_ptr(), st.data_ptr(), hs, None, None, 0, lo.data_ptr(), a_.data_ptr(),
                                  *dims, out.data_ptr(), records.data_ptr(), rec_bytes, stream) == 0
    res = {}
    for name in ("product", "records", "records_swap"):
        gv = torch.zeros_like(v)
        gl, ga = torch.full_like(lo, float("nan")), torch.full_like(a_, float("nan"))
        ws = torch.zeros(ws_bytes, dtype=torch.uint8, device=DEV)
        if name == "product":
            rc = L.msda_backward_ws(_lib.VARIANT_DEST | FLAG, BF16, v.data_ptr(), sh.data_ptr(), st.data_ptr(), hs, lo.data_ptr(),
                                    a_.data_ptr(), go.data_ptr(), *dims, gv.data_ptr(), gl.data_ptr(), ga.data_ptr(), ws.data_ptr(),
                                    ws_bytes, stream)
        else:
            rc = L.msda_records_backward(FLAG | (_lib.FLAG_RECORDS_SWAP if name == "records_swap" else 0), BF16, v.data_ptr(),
                                         sh.data_ptr(), st.data_ptr(), hs, lo.data_ptr(), a_.data_ptr(), None, 0, go.data_ptr(), *dims,
                                         gv.data_ptr(), gl.data_ptr(), ga.data_ptr(), None, records.data_ptr(), rec_bytes,
                                         ws.data_ptr(), ws_bytes, stream)
        assert rc == 0, (name, rc)
        torch.cuda.synchronize()
        res[name] = (gv.cpu(), gl.cpu(), ga.cpu())
    for name in ("records", "records_swap"):
        assert same_bits(res[name][0], res["product"][0]), f"{name}: grad_value"
        within_a_rounding(res[name][1], res["product"][1], f"{name}: grad_sampling_loc")
        within_a_rounding(res[name][2], res["product"][2], f"{name}: grad_attn_weight")
    a64 = (value.astype(np.float64), pyr, starts, loc.astype(np.float64), aw.astype(np.float64))
    o_out = O.forward(*a64)
    o_gv, o_gl, o_ga = O.backward(*a64, gout.float().numpy().astype(np.float64))
    assert np.abs(out.float().cpu().numpy() - o_out).max() <= 2.0 ** -7 * np.abs(o_out).max()
    assert np.abs(res["records"][0].float().numpy() - o_gv).max() <= 2.0 ** -7 * np.abs(o_gv).max()
    np.testing.assert_allclose(res["records"][2].numpy(), o_ga, rtol=1e-4, atol=1e-5 * float(np.abs(o_ga).max()))


def test_records_route_with_far_samples_takes_the_sorting_pass():
    """projection rows whose offsets reach tens of pixels: the forward's binning raises the "far" flag, the backward rebuilds the
    float32 locations / weights from the group records (the route saves none) and the gated sorting pass writes grad_value --
    the same bits as the product route, which takes the same detour with its saved tensors (on the hardware the sorting pass
    is bit-repeatable: per-wave histograms, lane order)"""
    from rlipv2_amd import msda
    value0, shapes, starts, qproj0, ref, gout = _encoder_call([(64, 96), (32, 48), (16, 24), (8, 12)], 1, 8, seed=5, refdim=2)
    qproj0 = qproj0.clone()
    qproj0[..., :8 * 32] *= 10.0

    def run(route):
        msda.records_route = route
        try:
            value, qproj = value0.clone().requires_grad_(True), qproj0.clone().requires_grad_(True)
            out = msda.FusedMSDeformAttnFunction.apply(value, shapes, starts, qproj, ref, 64)
            out.backward(gout)
            torch.cuda.synchronize()
            return out.detach(), value.grad, qproj.grad
        finally:
            msda.records_route = False
    base, got = run(False), run(True)
    assert torch.isfinite(got[1].float()).all() and torch.isfinite(got[2].float()).all()
    within_a_rounding(got[2], base[2], "grad of the projection rows")
    assert same_bits(got[1], base[1]), "grad_value differs"


def test_train_step_with_the_records_route_is_the_same_step():
    """a small bfloat16 ParSeDA train step (2 x 384 x 480 images, 4 encoder layers) with msda.records_route on against the same
    step on the product kernels, by the tolerances of the host-route self-check (rlipv2_amd/routes.py): same loss, same gradients
    within the noise two runs of the plain step show; then the same with the step captured as a HIP graph"""
    from rlipv2_amd import msda, routes, train
    from test_zz_round5_gpu import _small_step
    model, criterion, step, batch = _small_step()
    try:
        ref_loss, ref_grads = routes._run(step, criterion, batch, None, 1234)
        again_loss, again_grads = routes._run(step, criterion, batch, None, 1234)
        noise = routes.distance(again_grads, ref_grads)
        seen = []
        real = msda.ms_deform_attn_fused_backward
        msda.records_route = True
        msda.ms_deform_attn_fused_backward = lambda *a: (seen.append(a[8] is not None if len(a) > 8 else False), real(*a))[1]
        try:
            loss, grads = routes._run(step, criterion, batch, None, 1234)
        finally:
            msda.ms_deform_attn_fused_backward = real
        assert seen and all(seen[-4:]), "the encoder layers did not take the records route"
        why = routes.compare(loss, grads, ref_loss, ref_grads, noise)
        assert why is None, why
        # captured (the records buffers live in the graph's pool, the control-block memset is a graph node): the gradients the
        # graphed step delivers with the route on, against the eager product step; dropout must be off for that comparison
        for mod in model.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
            elif isinstance(getattr(mod, "dropout", None), float):
                mod.dropout = 0.0
        msda.records_route = False
        ref_loss, ref_grads = routes._run(step, criterion, batch, None, 7)
        noise = routes.distance(routes._run(step, criterion, batch, None, 7)[1], ref_grads)
        msda.records_route = True
        graphed = train.graph_step_module(step, model, batch, None, criterion=criterion)
        _, total = graphed.run(*batch)
        torch.cuda.synchronize()
        got = [p.grad.detach().clone() for p in step.parameters() if p.requires_grad]
        why = routes.compare(float(total.float()), got, ref_loss, ref_grads, noise=noise)
        assert why is None, why
    finally:
        msda.records_route = False
