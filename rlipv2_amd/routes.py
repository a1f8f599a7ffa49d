"""Host routes of the train step that only exist on the GPU, and the start-up self-check that switches them on.

A *route* here is a restructure of the backward graph that computes the same function with fewer passes over HBM but
depends on behaviour no CPU test can show (an in-place GEMM epilogue into a tensor autograd also holds, a one-launch kernel
in place of an op sequence).  Such a route is OFF by default in the product package.  `validate()` runs the caller's own
train step (same model, same batch) with all of them off -- twice, to measure the run-to-run distance of the step itself --
and then once per route with only that route on; a route whose loss or gradients differ by more than the stated tolerances
stays off.  The verdicts travel into bench.py's JSON line (`config.host_routes`).  Data-parallel runs agree on the verdict
with one MIN all-reduce, so every rank runs the same graph.

Routes:
  residual_gradient_in_gemm  linear.py: the encoder layer's two residual blocks and the decoders' shared value-projection input
                             run linked backward nodes (the input-gradient GEMMs accumulate with beta = 1 into the tensor the
                             fused LayerNorm's backward returned; reference blocks: dab_deformable/deformable_transformer.py:
                             1283-1300, 1346-1401)
  one_launch_box_head        decoder.py: sigmoid(delta + inverse_sigmoid(ref)) of the prediction heads (hoi.py:2122-2138) as one
                             launch of csrc/decoder_glue.hip with sigmoid's own backward
  fused_wide_layer_norm      norm.py / swin.py: the Swin blocks' residual add + LayerNorm (models/swin/swin_transformer.py:386-401) as
                             one pass per direction at the Swin widths (csrc/layernorm_wide.hip); no effect on the R50 configurations
  fused_window_attention     swin.py: softmax(scale q k^T + bias + mask) v of WindowAttention (models/swin/swin_transformer.py:262-301)
                             as one kernel per direction on the packed qkv tensor (csrc/window_attention.hip); Swin configurations only
"""
from __future__ import annotations

import importlib

import torch

GPU_ONLY_ROUTES = {
    "residual_gradient_in_gemm": ("rlipv2_amd.linear", "residual_gradient_in_gemm"),
    "one_launch_box_head": ("rlipv2_amd.decoder", "one_launch_box_head"),
    "fused_wide_layer_norm": ("rlipv2_amd.norm", "fused_wide_layer_norm"),
    "fused_window_attention": ("rlipv2_amd.swin", "fused_window_attention"),
}

# tolerances (tests/test_zz_round4_gpu.py::test_gradient_links_change_nothing_in_the_train_step_bf16): loss within 1e-3
# relative; whole gradient within max(2 %, 4 x the step's own run-to-run distance); every parameter within 5 % of its own
# norm plus a floor of 1e-3 of the largest parameter gradient
LOSS_RTOL = 1e-3
WHOLE_RTOL = 2e-2
NOISE_FACTOR = 4.0
PARAM_RTOL = 5e-2
PARAM_FLOOR = 1e-3


def get(name: str) -> bool:
    mod, attr = GPU_ONLY_ROUTES[name]
    return bool(getattr(importlib.import_module(mod), attr))


def set_route(name: str, on: bool) -> None:
    mod, attr = GPU_ONLY_ROUTES[name]
    setattr(importlib.import_module(mod), attr, bool(on))


def set_all(on: bool) -> None:
    for name in GPU_ONLY_ROUTES:
        set_route(name, on)


def state() -> dict:
    return {name: get(name) for name in GPU_ONLY_ROUTES}


def compare(loss, grads, ref_loss, ref_grads, noise=0.0):
    """None when (loss, grads) is the same step as (ref_loss, ref_grads) within the tolerances above, else the reason."""
    if not (loss == loss and abs(loss) != float("inf")):
        return "non-finite loss"
    if abs(loss - ref_loss) > LOSS_RTOL * abs(ref_loss) + 1e-6:
        return f"loss {loss:.6g} vs {ref_loss:.6g}"
    if len(grads) != len(ref_grads):
        return f"{len(grads)} gradients vs {len(ref_grads)}"
    diff = torch._foreach_norm(torch._foreach_sub([g.float() for g in grads], [g.float() for g in ref_grads]))
    ref = torch._foreach_norm([g.float() for g in ref_grads])
    diff, ref = torch.stack(diff).double().cpu(), torch.stack(ref).double().cpu()
    if not torch.isfinite(diff).all():
        return "non-finite gradient"
    whole, whole_ref = float(diff.square().sum().sqrt()), float(ref.square().sum().sqrt())
    if whole > max(WHOLE_RTOL, NOISE_FACTOR * noise) * whole_ref:
        return f"whole gradient differs by {whole / max(whole_ref, 1e-300):.3g} (step noise {noise:.3g})"
    bound = PARAM_RTOL * ref + PARAM_FLOOR * float(ref.max()) + NOISE_FACTOR * noise * whole_ref
    bad = int((diff > bound).sum())
    if bad:
        k = int((diff - bound).argmax())
        return f"{bad} parameter gradient(s) out of tolerance (worst: #{k}, {float(diff[k]):.3g} against norm {float(ref[k]):.3g})"
    return None


def distance(grads, ref_grads) -> float:
    diff = torch.stack(torch._foreach_norm(torch._foreach_sub([g.float() for g in grads], [g.float() for g in ref_grads])))
    ref = torch.stack(torch._foreach_norm([g.float() for g in ref_grads]))
    return float(diff.double().square().sum().sqrt() / ref.double().square().sum().sqrt().clamp_min(1e-300))


def _run(step_module, criterion, batch, autocast_dtype, seed):
    """loss + gradients of one eager forward / backward of the caller's step under a fixed random state (the dropouts of the
    language layers are live in train mode: same seed -> same masks, the routes do not change what the forward draws)"""
    samples, text, targets = batch
    params = [p for p in step_module.parameters() if p.requires_grad]
    for p in params:
        p.grad = None
    torch.manual_seed(seed)
    with torch.autocast(samples.tensors.device.type, dtype=autocast_dtype, enabled=autocast_dtype is not None):
        outputs = step_module(samples, text, targets)
    loss = criterion.weighted_sum(criterion(outputs, targets))
    loss.backward()
    grads = [p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p) for p in params]
    for p in params:
        p.grad = None
    return float(loss.detach().float()), grads


def validate(step_module, criterion, batch, autocast_dtype=None, group=None, seed=20251003, log=None, names=None) -> dict:
    """Self-check described in the module docstring.  Leaves every route that passed ON and the others OFF, and returns
    {route: "on" | "off (self-check failed: <reason>)" | "off (not applicable: <why>)"} for the routes in `names` (default: all).  No optimiser step is taken and the
    parameters' .grad are left empty; the caller's random state is restored."""
    import torch.distributed as dist
    device = batch[0].tensors.device
    if device.type != "cuda":
        set_all(False)
        return {name: "off (not applicable: the route exists on the GPU only)" for name in GPU_ONLY_ROUTES}
    names = list(GPU_ONLY_ROUTES) if names is None else [n for n in GPU_ONLY_ROUTES if n in set(names)]
    cpu_state, cuda_state = torch.get_rng_state(), torch.cuda.get_rng_state(device)
    verdict, measured = {}, {}
    try:
        set_all(False)
        ref_loss, ref_grads = _run(step_module, criterion, batch, autocast_dtype, seed)
        _, again = _run(step_module, criterion, batch, autocast_dtype, seed)
        noise = distance(again, ref_grads)
        del again
        for name in names:
            set_all(False)
            set_route(name, True)
            try:
                loss, grads = _run(step_module, criterion, batch, autocast_dtype, seed)
                why = compare(loss, grads, ref_loss, ref_grads, noise)
                measured[name] = distance(grads, ref_grads)          # (logged: what the tolerances can be tightened to)
                del grads
            except Exception as e:                                   # noqa: BLE001 -- a route that raises stays off
                why = f"{type(e).__name__}: {e}"
            verdict[name] = why
        if log is not None:
            log(f"[routes] step noise {noise:.3g} (whole-gradient distance of two plain runs; bar: max({WHOLE_RTOL:g}, {NOISE_FACTOR:g} x noise)); "
                + ", ".join(f"{k}: {'ok' if v is None else v}" + (f" (distance {measured[k]:.3g})" if k in measured else "")
                            for k, v in verdict.items()))
    finally:
        set_all(False)
        torch.set_rng_state(cpu_state)
        torch.cuda.set_rng_state(cuda_state, device)
    ok = torch.tensor([0 if verdict[name] is not None else 1 for name in names], dtype=torch.int32, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)       # every rank runs the same graph
    result = {}
    for name, passed in zip(names, ok.tolist()):
        set_route(name, bool(passed))
        result[name] = "on" if passed else "off (self-check failed: %s)" % (verdict[name] or "on another rank")
    return result
