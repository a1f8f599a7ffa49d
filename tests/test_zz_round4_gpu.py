"""GPU tests written in round 4, when no GPU was available at any time: they have never run.  They live in a file that sorts
last so that, under `pytest -x`, a surprise here cannot cut the established suite short.  What they check was verified on the
CPU as far as the CPU goes (goldens, bit-equality with the replaced form, node logic with torch stand-ins for the kernels)."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import test_modules_cpu as C  # noqa: E402,F401

from rlipv2_amd import decoder, deform_attn, encoder, parseda  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _golden_loss_gradients(model, bb, gg, g, trace):
    """outputs + gradients of the golden's loss (feature maps and sentinel parameters) of one run of the small model"""
    deform_attn.MSDeformAttn.trace = trace
    try:
        mc, out, feats, _ = C.run_small_parseda(model, bb, gg, device=DEV)
        loss = 0
        for k in C.KEYS:
            loss = loss + (out[k].float() * g["g_" + k].to(DEV)).sum() \
                + (out["aux_outputs"][0][k].float() * g["g_" + k].to(DEV)).sum() * 0.5
        loss.backward()
    finally:
        deform_attn.MSDeformAttn.trace = None
    grads = {f"g_feat{i}": t.grad.float().cpu() for i, (t, _) in enumerate(feats)}
    params = dict(model.named_parameters(remove_duplicate=False))
    for key in g:
        if key.startswith("gparam_") and g[key].numel():
            name = key[len("gparam_"):].replace("__", ".")
            grads[name] = params[name].grad.float().cpu()
    return out, grads


def test_full_parseda_bf16_gradients_against_float32_on_rounded_weights():
    """The headline dtype at model level, with stated tolerances.  bf16 policy: weights / activations / value in bfloat16;
    sampling geometry, softmax, bilinear weights and accumulators in float32.

    OUTPUTS against the float32 reference golden (two images of different size, 5 outputs + first auxiliary layer; measured
    with tools/bf16_parity_probe.py on MI355X in round 3, margins ~1.5-3x):
      logits  max |err| <= 2e-2 x max |reference|     (measured: subject / object 4.2e-3, verb 1.24e-2)
      boxes   max |err| <= 5e-3 absolute               (measured: 8.0e-4 / 1.8e-3)

    GRADIENTS of the golden's loss against a float32 run OF THIS MODEL on the bf16-rounded weights and inputs (itself pinned to
    the reference by test_full_parseda_f32), with the sampling pinned: every MSDeformAttn call of the float32 run is handed the
    bf16 run's projection rows and reference points (straight-through: values from the bf16 run, gradients through its own
    graph), so both runs take the same floor() decisions -- the model-level counterpart of `kink_samples` at op level: a sample
    whose location one bf16 rounding moves across a pixel centre changes ITS gradient completely and says nothing about the
    kernels.  What is left is the arithmetic: cosine >= 0.95 for the three feature maps and every sentinel parameter (round 3
    accepted 0.88 against the golden on UNROUNDED weights; a wrong-sign contribution of 45 % of the gradient norm passed that,
    at 0.95 it is 16 %).  The bar comes from the same comparison run with PyTorch's own CPU bfloat16 arithmetic, which rounds
    every intermediate (0.968-1.000, relative L2 0.02-0.27: weight rounding alone explains most of the distance to the golden);
    the values measured here are printed so that the bar can be raised on the first GPU run of the round.
    Parity proper is claimed in float32 (test_full_parseda_f32: logits 1e-3 rel, boxes 1e-4 abs)."""
    g = C.load("parseda")
    model, bb = C.build_small_parseda()
    model = model.to(DEV).to(torch.bfloat16)
    gb = {k: (v.to(torch.bfloat16) if v.dtype == torch.float32 else v) for k, v in g.items()}
    gb["img_mask"] = g["img_mask"]
    recorded = []

    def record(module, qproj, ref):
        recorded.append((qproj.detach().clone(), ref.detach().clone()))
        return qproj, ref

    out, grads = _golden_loss_gradients(model, bb, gb, g, record)
    for k in C.KEYS:
        for got, ref, what in ((out[k], g[k], k), (out["aux_outputs"][0][k], g["aux0_" + k], "aux " + k)):
            err = (got.float().cpu() - ref).abs().max().item()
            tol = 5e-3 if "boxes" in k else 2e-2 * ref.abs().max().item()
            assert err <= tol, (what, err, tol)

    ref_model, ref_bb = C.build_small_parseda()
    with torch.no_grad():
        for p in ref_model.parameters():
            p.copy_(p.to(torch.bfloat16).float())
    ref_model = ref_model.to(DEV)
    g32 = {k: (v.to(torch.bfloat16).float() if v.dtype == torch.float32 else v) for k, v in g.items()}
    g32["img_mask"] = g["img_mask"]
    replay = iter(recorded)

    def force(module, qproj, ref):
        q_rec, r_rec = next(replay)
        return qproj + (q_rec.float() - qproj).detach(), r_rec.float()

    _, ref_grads = _golden_loss_gradients(ref_model, ref_bb, g32, g, force)
    assert next(replay, None) is None and len(recorded) >= 6

    def cosine(a, b):
        a, b = a.double().flatten(), b.double().flatten()
        return float((a @ b) / (a.norm() * b.norm() + 1e-300))

    worst = {}
    for name, got in grads.items():
        assert torch.isfinite(got).all(), name
        ref = ref_grads[name]
        worst[name] = (cosine(got, ref), float((got - ref).double().norm() / (ref.double().norm() + 1e-300)))
    print("bf16 vs float32-on-rounded-weights, sampling pinned (cosine, relative L2):",
          {k: (round(c, 4), round(r, 4)) for k, (c, r) in worst.items()})
    for name, (c, r) in worst.items():
        assert c >= 0.95, (name, c, r)


@pytest.mark.first_contact(timeout=420)
def test_linked_encoder_layer_equals_the_unlinked_nodes_bf16():
    """DeformableTransformerEncoderLayer in bfloat16 with its two residual blocks linked (the value projection's and the FFN's
    input-gradient GEMMs accumulate into the tensors the fused LayerNorms return; encoder.py / linear.py, round 4) against the
    same layer with plain nodes (autograd sums the contributions): same kernels, so the outputs are equal and the gradients of
    the input, of `pos` and of all parameters agree to bf16 accumulation noise.  Tokens >= 4096 so that the fused FFN applies."""
    from rlipv2_amd import linear
    from rlipv2_amd.msda import attach_host_shapes
    torch.manual_seed(11)
    pyr = [(40, 54), (20, 27), (10, 14), (5, 7)]
    S = sum(h * w for h, w in pyr)
    shapes = torch.tensor(pyr, dtype=torch.long, device=DEV)
    attach_host_shapes(shapes, pyr)
    starts = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    layer = encoder.DeformableTransformerEncoderLayer(256, 1024, 0.0, "relu", 4, 8, 4).to(DEV).to(torch.bfloat16).train()
    N = 2
    x0 = torch.randn(N, S, 256, device=DEV).to(torch.bfloat16)
    pos0 = (0.5 * torch.randn(N, S, 256, device=DEV)).to(torch.bfloat16)
    gy = torch.randn(N, S, 256, device=DEV).to(torch.bfloat16)
    ref = encoder.encoder_reference_points(pyr, torch.ones(N, 4, 2, device=DEV), DEV)
    params = list(layer.parameters())

    def run(linked):
        linear.residual_gradient_in_gemm = linked
        try:
            for p in params:
                p.grad = None
            x = x0.clone().requires_grad_()
            pos = pos0.clone().requires_grad_()
            y = layer(x * 1.0, pos, ref, shapes, starts, None)
            kinds = set()
            stack = [y.grad_fn]
            while stack:
                n = stack.pop()
                if n is None or len(kinds) > 400:
                    continue
                kinds.add(type(n).__name__)
                stack.extend(f for f, _ in n.next_functions)
            y.backward(gy)
            return [y.detach().float(), x.grad.float(), pos.grad.float()] + [p.grad.float() for p in params], kinds
        finally:
            linear.residual_gradient_in_gemm = False            # (the package default: routes.validate switches it on)

    linked, kinds = run(True)
    plain, kinds_plain = run(False)
    assert "_AddIntoBackward" in kinds and "_AliasBackward" in kinds and "_AddIntoBackward" not in kinds_plain, kinds
    assert torch.equal(linked[0], plain[0])
    # (one GEMM accumulating in float32 against a rounded GEMM + a rounded sum: about one bf16 rounding apart per element)
    biggest = max(float(b.norm()) for b in plain[1:])
    for a, b in zip(linked[1:], plain[1:]):
        assert float((a - b).norm()) <= 2e-2 * float(b.norm()) + 1e-3 * biggest, (float((a - b).norm()), float(b.norm()))


@pytest.mark.first_contact(timeout=420)
def test_gradient_links_change_nothing_in_the_train_step_bf16():
    """Whole train step of a small bf16 model (encoder layers with both residual blocks linked, the image memory shared by the
    decoders' value projections: linear.residual_gradient_in_gemm) against the same step with plain autograd sums: same loss,
    every parameter gradient within bf16 accumulation noise.  Dropout off; 2 x 384 x 480 images = 7 656 tokens (fused FFN on)."""
    from rlipv2_amd import linear, train
    torch.manual_seed(0)
    margs = parseda.default_args(num_queries=40, enc_layers=4, dec_layers=2)
    model, criterion = train.build_training(margs, device=DEV, with_text_encoder=True)
    train.to_bf16(model)
    for mod in model.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    model.train()
    step = train.ParSeDATrainStep(model)
    batch = train.synthetic_batch(2, 384, 480, n_obj=13, n_verb=7, triplets=3, device=DEV, seed=1)
    batch[0].tensors = batch[0].tensors.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    train.freeze_parameters_without_gradient(step, criterion, batch)
    res = {}
    for linked in (True, False, True):
        linear.residual_gradient_in_gemm = linked
        try:
            model.zero_grad(set_to_none=True)
            loss = criterion.weighted_sum(criterion(step(*batch), batch[2]))
            loss.backward()
            got = (float(loss), {n: p.grad.float().clone() for n, p in model.named_parameters() if p.grad is not None})
        finally:
            linear.residual_gradient_in_gemm = False            # (the package default: routes.validate switches it on)
        if linked and True in res:                               # the linked step again: how far two runs of it are apart
            noise = max(float((got[1][n] - res[True][1][n]).norm() / res[True][1][n].norm().clamp_min(1e-12)) for n in got[1])
        res[linked] = got
    assert abs(res[True][0] - res[False][0]) <= 1e-3 * abs(res[False][0])
    assert res[True][1].keys() == res[False][1].keys()
    # whole gradient within 2 % (or 4x the run-to-run distance of the linked step itself); every parameter within 5 % of its
    # own norm plus a floor relative to the largest gradient (parameters whose gradient is rounding noise)
    num = sum(float((res[True][1][n] - g).norm()) ** 2 for n, g in res[False][1].items()) ** 0.5
    den = sum(float(g.norm()) ** 2 for g in res[False][1].values()) ** 0.5
    assert num <= max(2e-2, 4 * noise) * den, (num / den, noise)
    biggest = max(float(g.norm()) for g in res[False][1].values())
    for n, g in res[False][1].items():
        d = float((res[True][1][n] - g).norm())
        assert d <= 5e-2 * float(g.norm()) + 1e-3 * biggest, (n, d, float(g.norm()))


@pytest.mark.first_contact(timeout=420)
def test_residual_gradient_in_the_ffn_gemm_matches_the_unlinked_nodes():
    """linear.ffn_residual_norm: norm(x + FFN(x)) with the residual's gradient accumulated by the FFN's last GEMM (beta = 1, in
    place into the tensor the LayerNorm's backward returned) against the same two nodes unlinked (autograd sums the two
    gradients).  Same kernels either way: outputs equal, gradients within bf16 accumulation noise; the linked form's FFN node
    returns no gradient of its own."""
    from rlipv2_amd import linear
    torch.manual_seed(4)
    T = 4 * 2222
    lin1 = torch.nn.Linear(256, 1024).cuda().to(torch.bfloat16)
    lin2 = torch.nn.Linear(1024, 256).cuda().to(torch.bfloat16)
    ln = torch.nn.LayerNorm(256).cuda().to(torch.bfloat16)
    x0 = torch.randn(4, T // 4, 256, device="cuda").to(torch.bfloat16)
    dy = torch.randn(4, T // 4, 256, device="cuda").to(torch.bfloat16)
    params = (*lin1.parameters(), *lin2.parameters(), *ln.parameters())

    def run(linked):
        linear.residual_gradient_in_gemm = linked
        try:
            for p in params:
                p.grad = None
            x = x0.clone().requires_grad_()
            src = x * 1.0                                      # a non-leaf input, as in the encoder layer
            y = linear.ffn_residual_norm(src, lin1, lin2, ln)
            names = [type(n).__name__ for n, _ in y.grad_fn.next_functions if n is not None]
            y.backward(dy)
            return [y.detach().float(), x.grad.float()] + [p.grad.float() for p in params], names
        finally:
            linear.residual_gradient_in_gemm = False            # (the package default: routes.validate switches it on)

    linked, names = run(True)
    plain, _ = run(False)
    assert any("_Alias" in n for n in names), names            # the linked route was taken
    assert torch.equal(linked[0], plain[0])
    biggest = max(float(p.norm()) for p in plain[1:])
    for f, p in zip(linked[1:], plain[1:]):
        assert float((f - p).norm()) <= 2e-2 * float(p.norm()) + 1e-3 * biggest, (float((f - p).norm()), float(p.norm()))


def test_step_cache_keeps_one_off_shapes_eager_and_evicts():
    """GraphedStepCache(capture_after=2, max_buckets=1): a bucket runs eagerly on first sight, is captured on the second, the
    least recently used capture goes when another bucket is captured; every kind of step gives the eager loss and leaves
    gradients."""
    from rlipv2_amd import train
    torch.manual_seed(0)
    margs = parseda.default_args(num_queries=40)
    model, criterion = train.build_training(margs, device=DEV, with_text_encoder=True)
    train.to_bf16(model)
    step = train.ParSeDATrainStep(model)
    model.train()
    for mod in model.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0

    def make(triplets, pad, seed):
        b = train.synthetic_batch(2, 256, 320, n_obj=13, n_verb=7, triplets=triplets, device=DEV, seed=seed)
        b[0].tensors = b[0].tensors.to(torch.bfloat16)
        if pad:
            b[0].mask[1, :, 280:] = True
            b[0].tensors[1, :, :, 280:] = 0
            b[0].no_padding = False
        return b

    train.freeze_parameters_without_gradient(step, criterion, make(3, False, 0))
    cache = train.GraphedStepCache(step, model, criterion=criterion, capture_after=2, max_buckets=1)
    kinds = []
    for batch in (make(3, False, 0), make(3, False, 1), make(5, True, 2), make(5, True, 3), make(3, False, 4)):
        with torch.no_grad():
            eager = criterion.weighted_sum(criterion(step(*batch), batch[2])).float()
        g = cache.get(batch)
        kinds.append(type(g).__name__)
        _, total = g.run(*batch)
        torch.testing.assert_close(total.float(), eager, rtol=3e-2, atol=3e-2)
        assert all(p.grad is not None for p in step.parameters() if p.requires_grad)
    assert kinds == ["EagerSyncStep", "GraphedStep", "EagerSyncStep", "GraphedStep", "GraphedStep"], kinds
    assert cache.captures == 3 and cache.evictions == 2 and len(cache.graphs) == 1


@pytest.mark.first_contact(timeout=420)
def test_box_head_equals_the_op_sequence():
    """decoder.box_head (one launch of the refinement kernel forward, sigmoid's backward) against
    sigmoid(delta + inverse_sigmoid(ref)) as PyTorch ops: values within 1e-6, gradients equal after the cast back."""
    from rlipv2_amd.blocks import inverse_sigmoid
    decoder.one_launch_box_head = True
    try:
        _box_head_cases(inverse_sigmoid)
    finally:
        decoder.one_launch_box_head = False                 # (the package default: routes.validate switches it on)


def _box_head_cases(inverse_sigmoid):
    for dtype in (torch.bfloat16, torch.float32):
        g = torch.Generator().manual_seed(5)
        N, n = 3, 37
        obj = torch.rand(N, n, 4, generator=g).cuda()
        obj[0, 0] = torch.tensor([0.0, 1.0, 1e-7, 1 - 1e-7])
        delta = (torch.randn(N, n, 4, generator=g) * 2).to(dtype).cuda()
        d1 = delta.clone().requires_grad_(True)
        d2 = delta.clone().requires_grad_(True)
        y1 = decoder.box_head(d1, obj)
        y2 = (d2.float() + inverse_sigmoid(obj)).sigmoid()
        assert type(y1.grad_fn).__name__ == "BoxHeadFunctionBackward"
        assert (y1 - y2).abs().max() <= 1e-6
        w = torch.randn(N, n, 4, generator=g).cuda()
        (y1 * w).sum().backward()
        (y2 * w).sum().backward()
        assert d1.grad.dtype == dtype and (d1.grad.float() - d2.grad.float()).abs().max() <= (2.0 ** -7 if dtype == torch.bfloat16 else 1e-6)
        # a reference that carries a gradient (the learnable anchors of layer 0) keeps the differentiable op sequence
        r = obj.clone().clamp(0.05, 0.95).requires_grad_(True)
        decoder.box_head(delta, r).sum().backward()
        assert r.grad is not None and torch.isfinite(r.grad).all()
