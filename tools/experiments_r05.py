"""A/B of the unmeasured kernel arms in ONE call, every arm in its own child process with a timeout (an arm that has never run
on hardware may hang or fault: the parent only ever loses that arm).  Used two ways:

    python tools/experiments_r05.py                    # prints one JSON object (and a readable table on stderr)
    bench.py (default, 1 GPU, train_step)              # runs it AFTER its timed region and puts the object into the JSON line
                                                       # as `experiments` -- evidence only, the product path is not changed

Arms (ablation build, tools/_build/librlipv2_msda_ablation.so = `make -C rlipv2_amd/csrc ablation`; csrc/msda_patch.hip):
  cell 2-4   cell_backward_kernel<., MODE>: geometry once per quad + operand swap | + loads up front, scalar level starts |
             mode 3 without the swap (the last two arms only with --all)                                  (reference work: ms_deform_im2col_cuda.cuh:87-159, 301-403)
  patch multi   patch_dest_multi_kernel (mask-word prefetch not in a branch)
  patch cellg   (round 6) that kernel reading grad_out rows from a cell-major copy -- no query decode per candidate, neighbouring
                candidates share 128-byte lines; the copy kernel's time is part of the arm
  records    the "records" route (csrc/msda_cell_forward.inc EMIT + csrc/msda_cell_records.inc): the forward leaves per-sample
             records / window tables / patch masks, the backward runs no geometry and no binning -- fused call, forward and
             backward times against the product kernels, gradients bit for bit                                  (same lines)
  step       if (and only if) the records route's gradients are bit-equal: one short bench.py run of the whole train step with it
  fwd cell   cell_forward_kernel (explicit variant "cell" of the product library) against the product forward (.cuh:237-299)
  uniform    (round 6, `--uniform-arms`) SURVEY 8d input A, uniform locations: the far-return arm (cell kernel stops once a far sample
             is seen, gated K1 writes the location / weight gradients) and the queue-fed launches of the sorting fallback
  swin       the two Swin routes of round 5 (csrc/window_attention.hip, csrc/layernorm_wide.hip; models/swin/swin_transformer.py:
             262-301, 386-401) at the Swin-L stage-0 shapes against the PyTorch op sequences they replace
Every backward arm is compared with the default's gradients (64-bit digests of the raw bits, computed on the device, and -- through
a file in a private temporary directory the default arm leaves -- per-tensor closeness: grad_value must be bit-equal, the float32-formula gradients within a
rounding) on the encoder shape (N = 4, 800x1333 pyramid, bf16, model-like locations), B0 signature and fused geometry route;
time = HIP events around 20 calls of the whole backward.
"""
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ABLATION_LIB = os.path.join(ROOT, "tools", "_build", "librlipv2_msda_ablation.so")

ARMS = [("default", {}),
        ("cell 3 (geometry once per quad + operand swap + loads up front)", {"RLIPV2_CELL_SHARED": "3"}),
        ("cell 2 (geometry once per quad + operand swap)", {"RLIPV2_CELL_SHARED": "2"}),
        ("patch multi", {"RLIPV2_PATCH_MULTI": "1"}),
        ("cell 4 (3 without swap)", {"RLIPV2_CELL_SHARED": "4"}),
        ("cell 3 + patch multi", {"RLIPV2_CELL_SHARED": "3", "RLIPV2_PATCH_MULTI": "1"}),
        # round 6: the patch pass reads grad_out rows from a cell-major copy (grad_out_cells_kernel + patch_dest_multi_kernel<., ., true>)
        ("patch cellg", {"RLIPV2_PATCH_CELLG": "1"}),
        ("cell 3 + patch cellg", {"RLIPV2_CELL_SHARED": "3", "RLIPV2_PATCH_CELLG": "1"})]
KEYS = ("RLIPV2_CELL_SHARED", "RLIPV2_PATCH_REPS", "RLIPV2_PATCH_MULTI", "RLIPV2_PATCH_CELLG", "RLIPV2_CELL_FAR_RETURN", "RLIPV2_DEST_QUEUE")


def digest(t):
    """two 64-bit sums over the raw bits of a tensor (position-weighted: a permutation of equal values is seen too)"""
    import torch
    raw = t.contiguous().view(torch.int16 if t.element_size() == 2 else torch.int32).reshape(-1).to(torch.int64)
    pos = torch.arange(raw.numel(), device=raw.device, dtype=torch.int64) % 65521 + 1
    return [int(raw.sum()), int((raw * pos).sum())]


REF_ENV = "RLIPV2_EXPERIMENTS_REF_FILE"      # the default arm's gradients, for the other arms' children: a file in a private
                                             # temporary directory the parent makes (mkdtemp, mode 0700) and removes


def ref_file():
    return os.environ.get(REF_ENV)


def closeness(res, ref):
    """per tensor: bit-equal?, largest difference relative to the reference's largest magnitude, share of elements that differ.
    (What decides on the hardware: grad_value comes out of the SAME kernel from the same operands in every arm and must be bit-equal;
    the other gradients go through float32 formulas whose FMA contraction the compiler chooses per kernel -- even per sample position
    inside the product kernel, profiles/r05_records_route_static.txt -- so there the bar is "within a rounding of the output type".)"""
    import torch
    rows = []
    for a, b in zip(res, ref):
        a32, b32 = a.float(), b.float()
        rows.append({"equal_bits": bool(torch.equal(a.view(torch.int16 if a.element_size() == 2 else torch.int32),
                                                    b.view(torch.int16 if b.element_size() == 2 else torch.int32))),
                     "max_diff_rel_to_max": float((a32 - b32).abs().max() / b32.abs().max().clamp_min(1e-30)),
                     "differing_share": float((a32 != b32).float().mean())})
    return rows


def child_backward(arm):
    import torch
    from rlipv2_amd import msda
    from tools.msda_inputs import PYRAMID_800x1333, make_inputs
    from tools.patch_check import timed
    from tools.r03_experiments import fused_problem
    out = {}
    inp = make_inputs(4, mode="model", dtype=torch.bfloat16, seed=3)
    a = (inp["value"], inp["shapes"], inp["starts"], inp["loc"], inp["aw"], inp["grad_out"])
    res = msda.ms_deform_attn_backward(*a, 64)
    saved = {"b0": [t.cpu() for t in res]}
    out["b0"] = {"digest": [digest(t) for t in res], "finite": all(bool(torch.isfinite(t.float()).all()) for t in res),
                 "us": round(timed(lambda: msda.ms_deform_attn_backward(*a, 64), iters=20), 1)}
    msda.attach_host_shapes(inp["shapes"], PYRAMID_800x1333)
    qproj, ref = fused_problem(4, inp)
    _, loc, aw = msda.ms_deform_attn_fused_forward(inp["value"], inp["shapes"], inp["starts"], qproj, ref, True)
    hs = msda.host_shapes(inp["shapes"])
    f = lambda: msda.ms_deform_attn_fused_backward(inp["value"], inp["shapes"], inp["starts"], loc, aw, ref, inp["grad_out"], hs)  # noqa: E731
    res = [t for t in f() if torch.is_tensor(t)]
    saved["fused"] = [t.cpu() for t in res]
    out["fused"] = {"digest": [digest(t) for t in res], "finite": all(bool(torch.isfinite(t.float()).all()) for t in res),
                    "us": round(timed(f, iters=20), 1)}
    if arm == 0 and ref_file():
        torch.save(saved, ref_file())                      # (grad_value / grad_loc / grad_aw and grad_value / grad_qproj: ~300 MB)
    elif ref_file() and os.path.exists(ref_file()):
        ref = torch.load(ref_file(), weights_only=True)    # (tensors only: nothing in the file is executed)
        for case in ("b0", "fused"):
            out[case]["vs_default"] = closeness(saved[case], ref[case])    # [grad_value, ...]
    print("RESULT " + json.dumps(out), flush=True)


def child_records():
    """the fused encoder call of the train step (N = 4, 800x1333, bf16, model-like projection rows): product route (quad gather
    forward, cell_backward_kernel + patch_dest_kernel) against msda.records_route, with both operand orders of the 4x4x4 products"""
    import torch
    from rlipv2_amd import msda
    from tools.msda_inputs import PYRAMID_800x1333, make_inputs
    from tools.patch_check import timed
    from tools.r03_experiments import fused_problem
    inp = make_inputs(4, mode="model", dtype=torch.bfloat16, seed=3)
    msda.attach_host_shapes(inp["shapes"], PYRAMID_800x1333)
    qproj, ref = fused_problem(4, inp)
    hs = msda.host_shapes(inp["shapes"])
    fwd = lambda: msda.ms_deform_attn_fused_forward(inp["value"], inp["shapes"], inp["starts"], qproj, ref, True)   # noqa: E731
    out, res, grads = {}, {}, {}
    for name, route, swap, cell in (("product", False, False, False), ("cell_forward", False, False, True),
                                    ("records", True, False, False), ("records_swap", True, True, False)):
        msda.records_route, msda.records_swap, msda.fused_forward_cell = route, swap, cell
        try:
            o, loc, aw = fwd()
            records = msda.take_records()
            bwd = lambda: msda.ms_deform_attn_fused_backward(inp["value"], inp["shapes"], inp["starts"], loc, aw, ref,   # noqa: E731
                                                             inp["grad_out"], hs, records)
            g = bwd()
            torch.cuda.synchronize()
            res[name] = o.float()
            if name == "product":
                grads[name] = [t.clone() for t in g]
            else:
                out.setdefault(name, {})["vs_product"] = closeness(g, grads["product"])      # [grad_value, grad_qproj]
            out.setdefault(name, {}).update({"fwd_variant": msda.last_variant["fwd"], "bwd_variant": msda.last_variant["bwd"],
                                             "digest": [digest(t) for t in g], "finite": all(bool(torch.isfinite(t.float()).all()) for t in g),
                                             "fwd_us": round(timed(fwd, iters=20), 1)})
            if name != "cell_forward":
                out[name]["bwd_us"] = round(timed(bwd, iters=20), 1)
            if records is not None:
                out[name]["records_MB"] = round(records.numel() / 1e6, 1)
                out[name]["far_flag"] = int(records[:256].view(torch.int32)[60])
            del o, loc, aw, records, g
        finally:
            msda.records_route, msda.records_swap, msda.fused_forward_cell = False, True, False
    base = out["product"].pop("digest")
    scale = float(res["product"].abs().max())
    for name in ("cell_forward", "records", "records_swap"):
        out[name]["equal_bits"] = out[name].pop("digest") == base
        v = out[name]["vs_product"]
        # grad_value bit-equal (same patch pass, same operands), the projection rows' gradient within a bfloat16 rounding
        out[name]["accepted"] = bool(out[name]["finite"] and v[0]["equal_bits"] and v[1]["max_diff_rel_to_max"] <= 2.0 ** -7
                                     and v[1]["differing_share"] <= 0.05)
        out[name]["out_max_diff_rel_to_max"] = float((res[name] - res["product"]).abs().max()) / scale
    print("RESULT " + json.dumps(out), flush=True)


def child_forward():
    import torch
    from rlipv2_amd import msda
    from tools.msda_inputs import make_inputs
    from tools.patch_check import timed
    out = {}
    for mode in ("model", "init"):
        inp = make_inputs(4, mode=mode, dtype=torch.bfloat16, seed=3)
        a = (inp["value"], inp["shapes"], inp["starts"], inp["loc"], inp["aw"])
        msda.attach_host_shapes(inp["shapes"], [(100, 167), (50, 84), (25, 42), (13, 21)])
        res, t = {}, {}
        for v in ("quad", "cell"):
            msda.set_variant(v, "quad")
            try:
                res[v] = msda.ms_deform_attn_forward(*a, 64).float()
                torch.cuda.synchronize()
                t[v] = round(timed(lambda: msda.ms_deform_attn_forward(*a, 64), iters=20), 1)
            finally:
                msda.set_variant("auto")
        d = (res["cell"] - res["quad"]).abs()
        scale = float(res["quad"].abs().max())
        out[mode] = {"quad_us": t["quad"], "cell_us": t["cell"], "max_diff_rel_to_max": float(d.max()) / scale,
                     "mean_diff_rel_to_max": float(d.mean()) / scale, "non_finite": int((~torch.isfinite(res["cell"])).sum())}
    print("RESULT " + json.dumps(out), flush=True)


def child_swin():
    """the Swin routes of round 5 at stage 0 of Swin-L on 800 x 1333 (batch 2: [2, 200, 334, 192], 6 heads, window 7): one
    BasicLayer (two blocks, the second shifted; padded to 203 x 336) forward + backward with fused_window_attention (image-order
    kernel: pad / shift / partition as addressing) and fused_wide_layer_norm against the PyTorch op sequence -- agreement and
    HIP-event times; then the stage's weight gradients"""
    import torch
    from rlipv2_amd import norm, swin
    from tools.patch_check import timed
    out = {}
    torch.manual_seed(0)
    dev = "cuda:0"
    C, heads, B = 192, 6, 2
    layer = swin.BasicLayer(C, 2, heads, window_size=7, downsample=False).to(dev).to(torch.bfloat16)
    for n, p in layer.named_parameters():
        if "norm" in n or "relative_position_bias_table" in n:
            p.requires_grad_(False)                                 # the reference's freezing rule (models/swin/backbone.py:66-69)
    x0 = torch.randn(B, 200, 334, C, device=dev).to(torch.bfloat16)
    gy = torch.randn_like(x0)
    res, t = {}, {}

    def step():
        for p in layer.parameters():
            p.grad = None
        x = x0.clone().requires_grad_(True)
        y, _, _ = layer(x)
        y.backward(gy)
        return y.detach().float(), x.grad.float()
    for fused in (False, True):
        norm.fused_wide_layer_norm = swin.fused_window_attention = fused
        try:
            res[fused] = step()
            torch.cuda.synchronize()
            t[fused] = round(timed(step, iters=5), 1)
        finally:
            norm.fused_wide_layer_norm = swin.fused_window_attention = False
    rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-30))                                     # noqa: E731
    out["stage0_two_blocks_fwd_bwd"] = {"ops_us": t[False], "fused_us": t[True], "rel_l2_out": rel(res[True][0], res[False][0]),
                                        "rel_l2_dx": rel(res[True][1], res[False][1]),
                                        "finite": bool(torch.isfinite(res[True][0]).all() and torch.isfinite(res[True][1]).all()),
                                        "note": "both timings include a clone of the 51 MB input"}
    del layer, res
    # weight gradients of stage 0 (192 / 576 channels: 64-multiples the MFMA weight-gradient kernel does not take): the library
    # GEMM against the kernel on zero-padded operands (linear.pad_wgrad_to_128)
    from rlipv2_amd import linear
    T = B * 200 * 334
    wg = {}
    for name, M, K in (("qkv 576x192", 576, 192), ("proj 192x192", 192, 192), ("fc1 768x192", 768, 192), ("fc2 192x768", 192, 768)):
        dy, x = torch.randn(T, M, device=dev).to(torch.bfloat16), torch.randn(T, K, device=dev).to(torch.bfloat16)
        lib_us = round(timed(lambda: dy.t().mm(x), iters=10), 1)
        linear.pad_wgrad_to_128 = True
        try:
            dw, _ = linear.linear_wgrad(dy, x, with_bias=True, out_dtype=torch.bfloat16)
            pad_us = round(timed(lambda: linear.linear_wgrad(dy, x, with_bias=True, out_dtype=torch.bfloat16), iters=10), 1)
        finally:
            linear.pad_wgrad_to_128 = False
        wg[name] = {"library_us": lib_us, "padded_kernel_us": pad_us, "rel_l2": rel(dw.float(), dy.t().mm(x).float())}
        del dy, x
    out["stage0_weight_gradients"] = wg
    print("RESULT " + json.dumps(out), flush=True)


def child_stp():
    """the decoders' cross-attention at the bench shape (N = 4, 300 queries, 88 892 memory tokens, bf16), forward + backward of one
    MSDeformAttn call: the standard order (project all memory tokens, then sample) against deform_attn.sample_then_project
    (sample the unprojected memory -- generic forward with M' = 1, D' = 256, csrc/msda_rows.hip backward --, then project the
    2 400 sampled rows per image)"""
    import torch
    from rlipv2_amd import deform_attn
    from rlipv2_amd.msda import attach_host_shapes
    from tools.msda_inputs import PYRAMID_800x1333, level_tensors
    from tools.patch_check import timed
    torch.manual_seed(0)
    dev = "cuda:0"
    shapes, starts = level_tensors(PYRAMID_800x1333, dev)
    attach_host_shapes(shapes, PYRAMID_800x1333)
    S, N, Lq, C = int(shapes.prod(1).sum()), 4, 300, 256
    m = deform_attn.MSDeformAttn(C, 4, 8, 4).to(dev).to(torch.bfloat16)
    with torch.no_grad():
        m.sampling_offsets.weight.copy_((0.05 * torch.randn_like(m.sampling_offsets.weight.float())).to(torch.bfloat16))
    q0 = torch.randn(N, Lq, C, device=dev).to(torch.bfloat16)
    src0 = torch.randn(N, S, C, device=dev).to(torch.bfloat16)
    c = torch.rand(N, Lq, 1, 2, device=dev) * 0.8 + 0.1
    wh = torch.rand(N, Lq, 1, 2, device=dev) * 0.45 + 0.05
    ref = torch.cat([c, wh], -1).expand(N, Lq, 4, 4).contiguous()
    go = torch.randn(N, Lq, C, device=dev).to(torch.bfloat16)
    res, t = {}, {}

    def step():
        for p in m.parameters():
            p.grad = None
        q, src = q0.clone().requires_grad_(True), src0.clone().requires_grad_(True)
        out = m(q, ref, src, shapes, starts, None)
        out.backward(go)
        return out.detach().float(), src.grad.float(), m.value_proj.weight.grad.float()
    for stp in (False, True):
        deform_attn.sample_then_project = stp
        try:
            res[stp] = step()
            torch.cuda.synchronize()
            t[stp] = round(timed(step, iters=10), 1)
        finally:
            deform_attn.sample_then_project = False
    rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-30))                                     # noqa: E731
    print("RESULT " + json.dumps({"standard_us": t[False], "sample_then_project_us": t[True],
                                  "rel_l2_out": rel(res[True][0], res[False][0]), "rel_l2_d_src": rel(res[True][1], res[False][1]),
                                  "rel_l2_d_value_proj_weight": rel(res[True][2], res[False][2]),
                                  "note": "both timings include a clone of the 45 MB memory; sample-then-project = generic forward "
                                          "(M' = 1, D' = 256) + csrc/msda_rows.hip backward (ownership scatter, no atomics)"}),
          flush=True)


UNIFORM_ARMS = [("default", {}), ("far return", {"RLIPV2_CELL_FAR_RETURN": "1"}),
                ("far return + queue-fed fallback", {"RLIPV2_CELL_FAR_RETURN": "1", "RLIPV2_DEST_QUEUE": "1"})]


def child_uniform():
    """SURVEY 8d input A -- UNIFORM sampling locations (the reference's test recipe, models/ops/test.py:38), where every call has
    "far" samples and takes the sorting fallback: the whole backward of the encoder shape (N = 4, bf16, B0 signature) on the
    ablation library, once per process with the arm's switches in the environment (round 3: 1 339 us; round 2's kernels 975-1 130).
    Also timed on model-like locations: what the arm's idle launches cost the normal case."""
    import torch
    from rlipv2_amd import msda
    from tools.msda_inputs import PYRAMID_800x1333, make_inputs
    from tools.patch_check import timed
    out = {}
    for mode in ("uniform", "model"):
        inp = make_inputs(4, mode=mode, dtype=torch.bfloat16, seed=3)
        msda.attach_host_shapes(inp["shapes"], PYRAMID_800x1333)
        a = (inp["value"], inp["shapes"], inp["starts"], inp["loc"], inp["aw"], inp["grad_out"])
        res = msda.ms_deform_attn_backward(*a, 64)
        torch.cuda.synchronize()
        out[mode] = {"digest": [digest(t) for t in res], "finite": all(bool(torch.isfinite(t.float()).all()) for t in res),
                     "us": round(timed(lambda: msda.ms_deform_attn_backward(*a, 64), iters=10), 1),
                     "again_equal_bits": [digest(t) for t in msda.ms_deform_attn_backward(*a, 64)] == [digest(t) for t in res]}
        if ref_file():
            path = ref_file() + "." + mode
            if not os.path.exists(path):
                torch.save([t.cpu() for t in res], path)
            else:
                out[mode]["vs_default"] = closeness([t.cpu() for t in res], torch.load(path, weights_only=True))
    print("RESULT " + json.dumps(out), flush=True)


def main_uniform(per_child_timeout=90):
    """`--uniform-arms`: the three arms of the uniform-location case, each in a child process on the ablation library"""
    if not os.path.exists(ABLATION_LIB):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "rlipv2_amd", "csrc"), "-j8", "ablation"], timeout=900)
    tmp_dir = tempfile.mkdtemp(prefix="rlipv2_experiments_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        base_env = {k: v for k, v in os.environ.items() if k not in KEYS and k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
        base_env.update({REF_ENV: os.path.join(tmp_dir, "default_arm.pt"), "RLIPV2_LIB_PATH": ABLATION_LIB})
        rep = {}
        for name, env in UNIFORM_ARMS:
            r = run_child(["--uniform"], dict(base_env, **env), per_child_timeout)
            for mode in ("uniform", "model"):
                if isinstance(r.get(mode), dict):
                    r[mode].pop("digest", None)
                    v = r[mode].get("vs_default")
                    # the location / weight gradients come from K1 instead of the cell kernel under "far return" (uniform): within a
                    # float32 rounding of the formulas; grad_value from the same sorting pass: bit-equal
                    r[mode]["accepted"] = bool(r[mode]["finite"] and r[mode]["again_equal_bits"] and (
                        v is None or (v[0]["equal_bits"] and all(x["max_diff_rel_to_max"] <= 2e-5 for x in v[1:]))))
            rep[name] = r
        return {"uniform_location_arms": rep}
    finally:
        shutil.rmtree(tmp_dir, ignore_errors=True)


def run_child(args, env, timeout):
    t0 = time.time()
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), *args], capture_output=True, text=True, timeout=timeout,
                           env=env, cwd=ROOT)
    except subprocess.TimeoutExpired:
        return {"error": f"timed out after {timeout} s (child killed)"}
    for line in r.stdout.splitlines():
        if line.startswith("RESULT "):
            out = json.loads(line[7:])
            out["wall_s"] = round(time.time() - t0, 1)
            return out
    return {"error": f"rc {r.returncode}: {(r.stderr or r.stdout)[-300:]}"}


def run_step_child(flags, env, timeout):
    """one short bench.py run (train step, graphed, no CPU baseline, no experiments) with extra flags; -> the numbers of its line that
    matter for an A/B against the parent's own measurement"""
    t0 = time.time()
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--no-experiments", *flags]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    except subprocess.TimeoutExpired:
        return {"error": f"timed out after {timeout} s (child killed)"}
    for line in reversed(r.stdout.splitlines()):
        if line.startswith("{"):
            d = json.loads(line)
            roof = d.get("roofline", {})
            return {"flags": " ".join(flags), "ms_per_step": d.get("ms_per_step"), "images_per_s": d.get("value"),
                    "roofline_kernel": roof.get("kernel"), "roofline_frac": roof.get("frac"), "mean_launch_us": roof.get("mean_launch_us"),
                    "msda_ms_per_step": roof.get("msda_ms_per_step"),
                    "encoder_kernels_us": {k: v.get("mean_us") for k, v in roof.get("all_kernels", {}).items() if k.startswith("enc")},
                    "host_routes": d.get("config", {}).get("host_routes"), "wall_s": round(time.time() - t0, 1)}
    return {"error": f"rc {r.returncode}: {(r.stderr or r.stdout)[-300:]}"}


def main(per_child_timeout=45, budget_s=150):
    tmp_dir = tempfile.mkdtemp(prefix="rlipv2_experiments_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        return _main(tmp_dir, per_child_timeout, budget_s)
    finally:
        shutil.rmtree(tmp_dir, ignore_errors=True)            # (also when the parent is interrupted: ~300 MB of memory-backed file)


def main_arms(per_child_timeout=60):
    """`--arms`: every backward arm of the ablation build against the default arm, nothing else (tools/gpu_triage_r06.py's
    family `backward_arms`).  Builds the ablation library first if it is missing (hipcc is on the GPU box too)."""
    if not os.path.exists(ABLATION_LIB):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "rlipv2_amd", "csrc"), "-j8", "ablation"], timeout=900)
    tmp_dir = tempfile.mkdtemp(prefix="rlipv2_experiments_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        return _main(tmp_dir, per_child_timeout, 3600, arms_only=True)
    finally:
        shutil.rmtree(tmp_dir, ignore_errors=True)


def _main(tmp_dir, per_child_timeout, budget_s, arms_only=False):
    """parent: one child per arm / kernel, most informative first; nothing is started after `budget_s` seconds and no child may
    run past the deadline (bench.py's default run must stay within minutes); `--all`: no budget, every arm"""
    everything = "--all" in sys.argv
    if everything:
        per_child_timeout, budget_s = 120, 3600
    t0 = time.time()
    left = lambda: budget_s - (time.time() - t0)                                                              # noqa: E731
    report = {"what": "unmeasured kernel arms, A/B in child processes (tools/experiments_r05.py); evidence only, product path unchanged",
              "shape": "encoder N=4, 800x1333 pyramid, bf16, model-like locations; us = HIP events around 20 whole calls"}
    base_env = {k: v for k, v in os.environ.items() if k not in KEYS and k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    base_env[REF_ENV] = os.path.join(tmp_dir, "default_arm.pt")
    have_arms = os.path.exists(ABLATION_LIB)
    arms, base = {}, [None]

    def child(args, env=None):
        if left() < 15:
            return {"error": "not started: time budget used up"}
        return run_child(args, env or base_env, int(min(per_child_timeout, left())))

    def arm(k):
        name, env = ARMS[k]
        if not have_arms or (k > 0 and base[0] is None):
            return
        out = child(["--arm", str(k)], dict(base_env, RLIPV2_LIB_PATH=ABLATION_LIB, **env))
        if "error" not in out:
            digests = {case: out[case].pop("digest") for case in ("b0", "fused")}
            if base[0] is None:
                base[0] = digests
            for case in ("b0", "fused"):
                out[case]["equal_bits"] = digests[case] == base[0][case]
                v = out[case].get("vs_default")
                if v is not None:        # grad_value bit-equal; float32 gradients to 2e-5, the bfloat16 projection-row gradient to a rounding
                    lim = 2e-5 if case == "b0" else 2.0 ** -7
                    out[case]["accepted"] = bool(out[case]["finite"] and v[0]["equal_bits"] and all(r["max_diff_rel_to_max"] <= lim for r in v[1:]))
                elif k == 0:
                    out[case]["accepted"] = True
        arms[name] = out
    if arms_only:
        for k in range(len(ARMS)):
            arm(k)
        report["encoder_backward_arms"] = arms if have_arms else {"error": "no ablation build (make -C rlipv2_amd/csrc ablation)"}
        if have_arms:
            # the records route on the ablation library with the CELLG arm: its backward kernel leaves the cell-major grad_out copy
            # itself (no copy kernel), the patch pass reads it -- `product` in this object = product cell kernel + copy kernel + CELLG
            report["encoder_records_route_cellg"] = child(["--records"], dict(base_env, RLIPV2_LIB_PATH=ABLATION_LIB, RLIPV2_PATCH_CELLG="1"))
        report["wall_s"] = round(time.time() - t0, 1)
        return report
    # order = value of the evidence: the default pair, the records route (kernel level, then -- only if its gradients are the product
    # kernels' bit for bit -- the whole train step with it), the decoders' route, the most complete cell arm, the other kernels
    arm(0)
    rec = report["encoder_records_route"] = child(["--records"])
    report["decoder_cross_attention_sample_then_project"] = child(["--stp"])
    good = [n for n in ("records", "records_swap") if isinstance(rec.get(n), dict) and rec[n].get("accepted")]
    if good and (left() >= 75 or everything):
        best = min(good, key=lambda n: rec[n]["fwd_us"] + rec[n]["bwd_us"])
        flags = ["--set", "msda.records_route=1", "--set", "msda.records_swap=" + ("1" if best == "records_swap" else "0")]
        report["train_step_with_records_route"] = run_step_child(flags, base_env, int(min(120, left())))
    elif good:
        report["train_step_with_records_route"] = {"error": "not started: time budget used up"}
    arm(1)
    report["encoder_backward_arms"] = arms if have_arms else {"error": "no ablation build (make -C rlipv2_amd/csrc ablation)"}
    report["encoder_forward_cell"] = child(["--fwd"])
    report["swin_routes"] = child(["--swin"])
    for k in range(2, len(ARMS) if everything else 4):
        arm(k)
    report["wall_s"] = round(time.time() - t0, 1)
    return report


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--arm":
        child_backward(int(sys.argv[2]))
    elif len(sys.argv) > 1 and sys.argv[1] == "--records":
        child_records()
    elif len(sys.argv) > 1 and sys.argv[1] == "--fwd":
        child_forward()
    elif len(sys.argv) > 1 and sys.argv[1] == "--swin":
        child_swin()
    elif len(sys.argv) > 1 and sys.argv[1] == "--stp":
        child_stp()
    elif len(sys.argv) > 1 and sys.argv[1] == "--uniform":
        child_uniform()
    elif "--uniform-arms" in sys.argv:
        rep = main_uniform()
        print(json.dumps(rep))
        ok = all("error" not in r and all(r[m].get("accepted") for m in ("uniform", "model")) for r in rep["uniform_location_arms"].values())
        sys.exit(0 if ok else 1)
    else:
        rep = main_arms() if "--arms" in sys.argv else main()
        for name, v in rep.get("encoder_backward_arms", {}).items():
            print(f"{name:36s} {json.dumps(v)}", file=sys.stderr)
        print(json.dumps(rep))
        if "--arms" in sys.argv:       # (the triage reads the exit code: every arm ran and was accepted against the default arm)
            arms = rep.get("encoder_backward_arms", {})
            ok = bool(arms) and "error" not in arms and all(
                "error" not in v and all(v[c].get("accepted") for c in ("b0", "fused")) for v in arms.values())
            sys.exit(0 if ok else 1)
