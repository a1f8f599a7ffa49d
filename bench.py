#!/usr/bin/env python3
"""bench.py -- the driver's benchmark contract for the RLIPv2-ParSeDA hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch of synthetic input.  The workload is
BASELINE.json config 2 (RLIP_ParSeDA_v2 R50, 4 levels, 300 queries, bf16, batch 4 per GPU,
800x1333 images -> pyramid 100x167 / 50x84 / 25x42 / 13x21, 64 text tokens).

Workloads (``--workload``)
  train_step (default): the full RLIP_ParSeDA_v2 R50 train step -- ResNet-50 (frozen BN) -> input
             projections -> 6-layer ALIF-fused deformable encoder (3 VLFuse + 3 RoBERTa layers) -> 3-layer
             DAB human-object decoder -> verb decoder -> heads -> SetCriterionHOI (Hungarian matching)
             -> backward -> clip_grad_norm_(0.1) -> AdamW, bf16 autocast with float32 master weights,
             RoBERTa-base-shaped text encoder with random weights in the step; data parallel over
             images with RCCL gradient all-reduce overlapped with backward (rlipv2_amd/train.py).
  msda_step: every multi-scale-deformable-attention call of one train step
             -- 6 encoder self-attention (Lq = S = 22223), 3 human-object decoder (Lq = 300) and
             3 verb decoder (Lq = 150) cross-attention calls, forward then backward, i.e. the
             12 + 12 launches SURVEY.md section 1 counts per step -- on synthetic model-like
             sampling locations (tools/msda_inputs.py), without the dense blocks around them.

Multi-GPU: the path shards over images with no exchange inside the MSDA op, so
every rank processes its own batch (weak scaling); ranks only meet in the timing barrier.

Output: ONE JSON line on rank 0 (fields per the driver contract) plus
  roofline     -- dominant kernel: algorithmic bytes per launch / mean launch duration measured
                  with HIP events on the launch stream inside the timed region; peak 8 TB/s HBM.
  cpu_baseline -- the CPU oracle (OpenMP C restatement, kind "port") timed on this box's host
                  cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch
import ctypes

import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from rlipv2_amd import _lib  # noqa: E402
from tools.msda_inputs import PYRAMID_800x1333, make_inputs  # noqa: E402

HBM_PEAK_GBPS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec peak
# BASELINE.json's metric, verbatim.  `value` is the WHOLE-JOB rate (the driver's contract), i.e. the per-GPU figure the
# metric's name speaks of times the number of GPUs; the MSDeformAttn GB/s half of the metric is the `roofline` object.
METRIC = "images/sec/GPU RLIPv2-ParSeDA R50 train step; MSDeformAttn HBM GB/s"


class MsdaCall:
    """One MSDA call site of the train step with preallocated operands and results."""

    def __init__(self, name, N, Lq, mode, dtype, seed, device):
        self.name = name
        inp = make_inputs(N, pyramid=PYRAMID_800x1333, Lq=Lq, mode=mode, dtype=dtype, device=device, seed=seed)
        self.inp = inp
        self.dims = inp["dims"]
        self.code = _lib.MSDA_BF16 if dtype == torch.bfloat16 else _lib.MSDA_F32
        N, S, M, D, L, Lq, P = self.dims
        self.out = torch.empty(N, Lq, M * D, dtype=dtype, device=device)
        # the library's backward with a workspace and the host copy of the level shapes (msda_backward_ws): grad_value
        # comes back in value's dtype, every row written exactly once (no zero-fill, no float atomics)
        self.host_shapes = (ctypes.c_int64 * (2 * L))(*[int(v) for hw in PYRAMID_800x1333 for v in hw])
        self.ws_bytes = int(_lib.lib().msda_backward_workspace_bytes(self.code, self.host_shapes, *self.dims))
        self.ws = torch.empty(max(self.ws_bytes, 16), dtype=torch.uint8, device=device)
        self.bf16_grad = self.ws_bytes > 0 and dtype == torch.bfloat16
        self.g_value = torch.empty(N, S, M, D, dtype=dtype if self.ws_bytes else torch.float32, device=device)
        self.g_loc = torch.empty_like(inp["loc"])
        self.g_aw = torch.empty_like(inp["aw"])
        self.bytes_fwd = _lib.algorithmic_bytes(self.code, False, *self.dims)
        self.bytes_bwd = _lib.algorithmic_bytes(self.code, True, *self.dims) - (N * S * M * D * 2 if self.bf16_grad else 0)
        self.variant_bwd = "dest" if self.ws_bytes else None
        self.ev = {"fwd": [], "bwd": []}

    def forward(self, lib, stream, timed):
        i = self.inp
        if timed:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
        st = lib.msda_forward(self.code, i["value"].data_ptr(), i["shapes"].data_ptr(), i["starts"].data_ptr(),
                              i["loc"].data_ptr(), i["aw"].data_ptr(), *self.dims, self.out.data_ptr(), stream)
        if timed:
            b.record()
            self.ev["fwd"].append((a, b))
        if st:
            raise RuntimeError(_lib.strerror(st))

    def backward(self, lib, stream, timed):
        i = self.inp
        if timed:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
        flags = _lib.FLAG_GRAD_VALUE_BF16 if self.bf16_grad else 0
        st = lib.msda_backward_ws(_lib.VARIANT_AUTO | flags, self.code, i["value"].data_ptr(),
                                  i["shapes"].data_ptr(), i["starts"].data_ptr(), self.host_shapes, i["loc"].data_ptr(),
                                  i["aw"].data_ptr(), i["grad_out"].data_ptr(), *self.dims,
                                  self.g_value.data_ptr(), self.g_loc.data_ptr(), self.g_aw.data_ptr(),
                                  self.ws.data_ptr(), self.ws_bytes, stream)
        if timed:
            b.record()
            self.ev["bwd"].append((a, b))
        if st:
            raise RuntimeError(_lib.strerror(st))


def build_msda_step(batch, dtype, device, rank):
    calls = []
    for k in range(6):
        calls.append(MsdaCall(f"enc{k}", batch, None, "model", dtype, 100 * rank + k, device))
    for k in range(3):
        calls.append(MsdaCall(f"ho_dec{k}", batch, 300, "decoder", dtype, 100 * rank + 10 + k, device))
    for k in range(3):
        calls.append(MsdaCall(f"verb_dec{k}", batch, 150, "decoder", dtype, 100 * rank + 20 + k, device))
    return calls


def run_step(calls, lib, stream, timed):
    for c in calls:
        c.forward(lib, stream, timed)
    for c in reversed(calls):
        c.backward(lib, stream, timed)


def b0_signature_kernels(batch, dtype, device, iters=10):
    """The B0-signature kernels (float32 sampling_loc / attn_weight operands, msda_forward / msda_backward_ws) at the
    encoder shape on synthetic model-like inputs: HIP events on the launch stream, mean of `iters` launches.  Reported
    next to the fused kernels the train step actually runs (both lines)."""
    lib = _lib.lib()
    call = MsdaCall("enc", batch, None, "model", dtype, 7, device)
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        call.forward(lib, stream, False)
        call.backward(lib, stream, False)
    for _ in range(iters):
        call.forward(lib, stream, True)
    for _ in range(iters):
        call.backward(lib, stream, True)
    torch.cuda.synchronize()
    res = {}
    for d, nbytes in (("fwd", call.bytes_fwd), ("bwd", call.bytes_bwd)):
        us = sum(a.elapsed_time(b) for a, b in call.ev[d]) / len(call.ev[d]) * 1e3
        res[f"enc_{d}"] = {"mean_us": round(us, 2), "algorithmic_bytes_per_launch": nbytes,
                           "GBps": round(nbytes / us / 1e3, 1), "frac": round(nbytes / us / 1e3 / HBM_PEAK_GBPS, 4)}
    return res


def cpu_baseline(calls, scope="msda_step"):
    """Oracle (OpenMP C port) on whole images through all 24 MSDA calls of a step."""
    import numpy as np

    from oracle import msda_oracle as O
    O.build()
    t_total, images = 0.0, 0
    nb = calls[0].dims[0]
    while t_total < 10.0 and images < 32:              # bounded: ~10 s of CPU work, at most 32 images
        k = images % nb
        for c in calls:
            i = c.inp
            a = [i["value"][k:k + 1].float().cpu().numpy(), i["shapes"].cpu().numpy(), i["starts"].cpu().numpy(),
                 i["loc"][k:k + 1].cpu().numpy(), i["aw"][k:k + 1].cpu().numpy()]
            go = i["grad_out"][k:k + 1].float().cpu().numpy()
            t0 = time.perf_counter()
            O.forward(*a, omp=True)
            O.backward(*a, go, omp=True)
            t_total += time.perf_counter() - t0
            del a, go
        images += 1
    torch_path = cpu_baseline_torch(calls)
    return {"torch_grid_sample": torch_path,
            "value": round(images / t_total, 4), "unit": "images/s", "cores": O.threads(True), "kind": "port",
            "sample": "%d image(s) through all 12 fwd + 12 bwd MSDA calls of one step, float32, "
                      "oracle/msda_oracle.c built with OpenMP; %.1f s of CPU work%s" % (
                          images, t_total, "" if scope == "msda_step" else
                          "; covers ONLY the MSDeformAttn calls of the train step (the oracle restates that op; the "
                          "reference's dense blocks have no CPU port here), i.e. an upper bound on a CPU step rate")}


def cpu_baseline_torch(calls):
    """The reference's own CPU formulation (per-level F.grid_sample + autograd, ms_deform_attn_func.py:45-65), restated
    in oracle/msda_oracle.py::forward_torch, timed on the host cores: forward + backward of ONE call of each kind
    (encoder, ho-decoder, verb-decoder; one image, float32), weighted by the number of such calls in a step."""
    from oracle import msda_oracle as O
    nthreads = torch.get_num_threads()
    kinds, t_kind, t_total = {}, {}, 0.0
    for c in calls:
        kinds.setdefault(c.name.rstrip("0123456789"), []).append(c)
    for kind, cs in kinds.items():
        i = cs[0].inp
        shapes = [tuple(int(v) for v in hw) for hw in i["shapes"].cpu().tolist()]
        best = None
        for _ in range(2):                                   # second run: warm allocator / thread pool
            value = i["value"][:1].float().cpu().requires_grad_(True)
            loc = i["loc"][:1].float().cpu().requires_grad_(True)
            aw = i["aw"][:1].float().cpu().requires_grad_(True)
            go = i["grad_out"][:1].float().cpu()
            t0 = time.perf_counter()
            out = O.forward_torch(value, shapes, loc, aw)
            out.backward(go.reshape(out.shape))
            dt = time.perf_counter() - t0
            t_total += dt
            best = dt if best is None else min(best, dt)
        t_kind[kind] = best
    step_s = sum(t_kind[k] * len(cs) for k, cs in kinds.items())
    return {"value": round(1.0 / step_s, 4), "unit": "images/s", "cores": nthreads, "kind": "port",
            "sample": "forward + autograd backward of one call of each kind for ONE image (float32, per-level "
                      "F.grid_sample restatement of the reference's CPU path, %d torch threads), weighted by the calls "
                      "per step (%s); seconds per call: %s; %.1f s of CPU work" % (
                          nthreads, ", ".join(f"{len(cs)} x {k}" for k, cs in kinds.items()),
                          {k: round(v, 3) for k, v in t_kind.items()}, t_total)}


class KernelTimer:
    """HIP-event brackets around the MSDA entry points while the train step runs (same stream)."""

    def __init__(self):
        from rlipv2_amd import msda
        self.msda = msda
        self.records = []          # (direction, dims, dtype_code, start, end)
        self.enabled = False
        self._fwd, self._bwd = msda.ms_deform_attn_forward, msda.ms_deform_attn_backward
        msda.ms_deform_attn_forward = self._wrap(self._fwd, "fwd")
        msda.ms_deform_attn_backward = self._wrap(self._bwd, "bwd")
        # the fused geometry + sampling entry points (what the module calls when the reference points need no gradient)
        msda.ms_deform_attn_fused_forward = self._wrap_fused(msda.ms_deform_attn_fused_forward, "fwd")
        msda.ms_deform_attn_fused_backward = self._wrap_fused(msda.ms_deform_attn_fused_backward, "bwd")

    def _wrap_fused(self, fn, direction):
        def timed(value, shapes, starts, *rest):
            if not self.enabled:
                return fn(value, shapes, starts, *rest)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            out = fn(value, shapes, starts, *rest)
            b.record()
            N, S, M, D = value.shape
            code = _lib.MSDA_BF16 if value.dtype == torch.bfloat16 else _lib.MSDA_F32
            if direction == "fwd":       # (value, shapes, starts, qproj, ref, save) -> (out, loc, aw)
                Lq = rest[0].shape[1]
                moved = [value, rest[0], rest[1], *[t for t in out if t is not None], self.msda._records_out[0]]
            else:                        # (value, shapes, starts, loc, aw, ref, grad_out, host[, records]) -> [g_value, g_qproj]
                Lq = rest[3].shape[1]    # (loc / aw are None under msda.records_route: the records buffer replaces them)
                moved = [value, rest[0], rest[1], rest[2], rest[3], *rest[5:], *out]
            nbytes = sum(t.numel() * t.element_size() for t in moved if torch.is_tensor(t))
            dims = (N, S, M, D, shapes.shape[0], Lq, 4)
            # grad_value written as bfloat16: 2 bytes per element less than the float32 grad_value of msda_algorithmic_bytes
            saved = N * S * M * D * 2 if (direction == "bwd" and out[0].dtype == torch.bfloat16) else 0
            self.records.append((direction, dims, code, a, b, (nbytes, saved), self.msda.last_variant.get(direction), True))
            return out
        return timed

    def _wrap(self, fn, direction):
        def timed(value, shapes, starts, loc, aw, *rest, **kw):
            if not self.enabled:
                return fn(value, shapes, starts, loc, aw, *rest, **kw)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            out = fn(value, shapes, starts, loc, aw, *rest, **kw)
            b.record()
            N, S, M, D = value.shape
            dims = (N, S, M, D, shapes.shape[0], loc.shape[1], loc.shape[4])
            code = _lib.MSDA_BF16 if value.dtype == torch.bfloat16 else (_lib.MSDA_F64 if value.dtype == torch.float64 else _lib.MSDA_F32)
            # grad_value written directly as bfloat16 (destination-stationary backward): 2 bytes per element less
            # than the float32 grad_value msda_algorithmic_bytes assumes
            saved = N * S * M * D * 2 if (direction == "bwd" and out[0].dtype == torch.bfloat16) else 0
            self.records.append((direction, dims, code, a, b, saved, self.msda.last_variant.get(direction), False))
            return out
        return timed

    def summary(self):
        kern = {}
        for direction, dims, code, a, b, saved, variant, fused in self.records:
            kind = "enc" if dims[5] == dims[1] else f"dec{dims[5]}"
            # `bytes` = the ALGORITHMIC bytes of SURVEY.md 8d for this call (value, sampling_loc, attn_weight, output and,
            # backward, their gradients, each once in the dtypes of the call) -- also for the fused kernels, which do
            # the same job (plus the module's sampling geometry) on fewer operand bytes: those are `operand_bytes`
            operand = None
            if fused:
                operand, saved = saved
            nbytes = _lib.algorithmic_bytes(code, direction == "bwd", *dims) - saved
            k = kern.setdefault(f"{kind}_{direction}" + ("_fused" if fused else ""),
                                {"ms": 0.0, "n": 0, "dims": dims, "code": code, "bwd": direction == "bwd",
                                 "variant": variant, "bytes": nbytes, "operand_bytes": operand})
            k["ms"] += a.elapsed_time(b)
            k["n"] += 1
        return kern


def run_train_step_bench(args, world, rank, local_rank, device):
    from rlipv2_amd import parseda, train
    margs = parseda.default_args(num_queries=args.queries)
    if os.environ.get("RLIPV2_MIOPEN_FIND", "0") == "1":
        torch.backends.cudnn.benchmark = True       # MIOpen Find on first use of every convolution shape
    torch.manual_seed(0 + rank)                                       # reference main.py:505
    model, criterion = train.build_training(margs, device=device, with_text_encoder=True, backbone_name=args.backbone)
    sizes = [(800, 1333), (736, 1100)] if args.padded else None
    master = args.dtype == "bf16" and args.precision == "master"
    if master:      # bf16 parameters / activations / gradients, float32 master weights in the optimiser
        train.to_bf16(model)

    def make_batch(triplets, seed):
        b = train.synthetic_batch(args.batch, 800, 1333, n_obj=43, n_verb=21, triplets=triplets, device=device,
                                  seed=seed, sizes=sizes)
        if master:
            b[0].tensors = b[0].tensors.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        return b

    batch = make_batch(8, rank)
    # variable-target variant: the batches of the timed loop rotate through three target counts
    rotation = [batch] + ([make_batch(6, rank + 100), make_batch(11, rank + 200)] if args.var_targets else [])
    step_module = train.ParSeDATrainStep(model)
    model.train()
    dtype = torch.bfloat16 if (args.dtype == "bf16" and not master) else None
    graphed = False
    synchronizer = None
    eager_step = step_module
    force_dp = os.environ.get("RLIPV2_FORCE_DP") == "1" and dist.is_initialized()   # 1-rank plumbing check
    schedule = "overlapped" if args.dp_overlap else args.dp_schedule        # auto | flat | overlapped (data-parallel runs only)
    overlap, schedule_reason = schedule == "overlapped", None
    frozen = False
    if world > 1 or force_dp:
        train.broadcast_parameters(model, 0)
        # static unused-parameter mask from a dry run (identical on every rank), instead of per-step graph
        # searches (find_unused_parameters=True in the reference, main.py:517)
        train.freeze_parameters_without_gradient(step_module, criterion, batch, autocast_dtype=dtype)
        frozen = True
    # GPU-only host routes (rlipv2_amd/routes.py) are off in the package.  `--host-routes auto`: main() ran the self-check in a CHILD
    # process before this process touched the GPU (routes_verdicts_from_child); here the verdicts are only applied -- this process
    # never executes a route that has not just reproduced the plain step's loss and gradients on the same model and batch.  A child
    # that crashed, hung or printed nothing leaves every route off.  Data-parallel runs agree with one MIN all-reduce.
    from rlipv2_amd import routes
    if args.host_routes == "auto":
        host_routes = apply_route_verdicts(getattr(args, "route_verdicts", None), world, device)
    else:
        routes.set_all(args.host_routes == "on")
        host_routes = {k: ("on (forced, no self-check)" if v else "off (forced)") for k, v in routes.state().items()}
    if args.graph and dtype is None:
        # HIP-graph the two model phases (forward graph + backward graph); the criterion's host-side
        # assignment and the optimiser stay eager.  Data-parallel runs average the gradients with one flat
        # RCCL all-reduce after the backward replay (train.GradientSynchronizer).
        try:
            if world == 1 and not force_dp:
                if not frozen:
                    train.freeze_parameters_without_gradient(step_module, criterion, batch)
            else:
                def make_sync():
                    s_ = train.GradientSynchronizer([p for p in step_module.parameters() if p.requires_grad])
                    # the 1 / world of the gradient average is applied inside the fused optimiser's kernels (no extra pass
                    # over the 425 MB gradient buffer); the float32-parameter optimiser path scales in the synchronizer
                    s_.scale_in_optimizer = bool(master)
                    return s_
                crit_in_graph = criterion if args.graph_criterion else None
                chosen = None
                if schedule == "auto" and crit_in_graph is not None:
                    # overlapped iff captured collectives replay on all ranks AND the first overlapped step equals the flat
                    # step (loss, gradient norm to 1e-3) on all ranks -- train.choose_dp_schedule; every rank decides alike
                    chosen, picked, schedule_reason = train.choose_dp_schedule(
                        lambda ov: train.graph_step_module(step_module, model, batch, make_sync(), criterion=crit_in_graph, overlap=ov),
                        batch, device, log=lambda m: print(m, file=sys.stderr))
                    overlap = picked == "overlapped"
                    print(f"[bench] gradient schedule (auto): {picked} -- {schedule_reason}", file=sys.stderr)
                    import gc
                    gc.collect()                                    # (the step that was not chosen: its graphs and flat buffer)
                    torch.cuda.empty_cache()
                elif schedule == "auto":
                    overlap, schedule_reason = False, "auto needs the criterion inside the graphs (--no-graph-criterion): flat"
                elif overlap and not train.captured_collective_selftest(device):
                    # (RCCL only; every other backend is answered "no" without a capture attempt; all ranks agree)
                    print("[bench] collectives cannot be captured into the backward graph here (not RCCL, or the self-test "
                          "failed): one flat all-reduce after the backward graph", file=sys.stderr)
                    overlap, schedule_reason = False, "forced overlapped, but the capture self-test failed: flat"
                if chosen is not None and not args.var_targets:
                    synchronizer = chosen.synchronizer
                else:
                    chosen = None                                   # (variable targets: the cache captures per bucket, below)
                    torch.cuda.empty_cache()
                    synchronizer = make_sync()
            if args.var_targets:
                step_module = train.GraphedStepCache(step_module, model, synchronizer, criterion=criterion, overlap=overlap)
                step_module.register(rotation)                  # capture every bucket before the timed region
            elif world > 1 or force_dp:
                step_module = chosen if chosen is not None else train.graph_step_module(
                    step_module, model, batch, synchronizer, criterion=criterion if args.graph_criterion else None, overlap=overlap)
            else:
                step_module = train.graph_step_module(step_module, model, batch, synchronizer,
                                                      criterion=criterion if args.graph_criterion else None, overlap=overlap)
            graphed = True
        except Exception as e:                                  # noqa: BLE001 -- fall back to eager, say so
            import traceback
            tb = "".join(traceback.format_exception(type(e), e, e.__traceback__)[-8:])
            print(f"[bench] graph capture failed, running eager: {type(e).__name__}\n{tb}", file=sys.stderr)
            torch.cuda.synchronize()
            step_module = eager_step
        if world > 1:
            # every rank must run the same collective schedule (one flat all-reduce vs DDP buckets): if the capture
            # failed anywhere, all ranks fall back to the eager / DDP path together
            ok = torch.tensor([1 if graphed else 0], device=device, dtype=torch.int32)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0 and graphed:
                print("[bench] graph capture failed on another rank, running eager here too", file=sys.stderr)
                graphed, step_module, synchronizer = False, eager_step, None
    if (world > 1 or force_dp) and not graphed:
        step_module = torch.nn.parallel.DistributedDataParallel(
            eager_step, device_ids=[local_rank], find_unused_parameters=False, gradient_as_bucket_view=True,
            bucket_cap_mb=64)
        eager_step = step_module
    optimizer = train.FusedMasterAdamW(model) if master else train.build_optimizer(model)
    timer = KernelTimer()
    guard = train.NonFiniteGuard()                       # engine.py:123-128, without a sync of its own
    for k in range(args.warmup):
        train.train_step(step_module, criterion, optimizer, rotation[k % len(rotation)], autocast_dtype=dtype, guard=guard)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    timer.enabled = True
    t0 = time.perf_counter()
    for k in range(args.steps):
        loss = train.train_step(step_module, criterion, optimizer, rotation[k % len(rotation)], autocast_dtype=dtype,
                                guard=guard)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    guard.check(wait=True)
    timer.enabled = False
    barrier()
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if graphed:                          # every rank: the criterion holds a collective
        # kernels inside a replayed graph cannot be bracketed with events: probe them in two eager steps
        # of the very same train step (same model, batch, optimiser), right after the timed region
        # (one untimed eager step first: the first eager backward after the graph replays re-grows the caching
        #  allocator's pools, which showed up as 4x longer decoder launches in the probe)
        # (data-parallel runs: forward + backward only -- `eager_step` has no synchroniser, an optimiser step on local
        #  gradients would let the ranks' weights drift apart before the process group is torn down)
        def probe_step():
            if world > 1:
                train._forward_backward(eager_step, criterion, optimizer, batch, dtype)
            else:
                train.train_step(eager_step, criterion, optimizer, batch, autocast_dtype=dtype)
        probe_step()
        torch.cuda.synchronize()
        timer.enabled = True
        for _ in range(2):
            probe_step()
        torch.cuda.synchronize()
        timer.enabled = False
    n_params = sum(p.numel() for p in model.parameters() if p.requires_grad)
    # step-level roofline: algorithmic bytes / matrix FLOPs of one eager step of the same model + batch
    step_roofline = None
    if rank == 0 and world == 1:       # (the probe runs a whole train step, collectives included: single-process runs only)
        from rlipv2_amd import roofline
        try:
            step_roofline = roofline.probe(
                lambda: train.train_step(eager_step, criterion, optimizer, batch, autocast_dtype=dtype))
        except Exception as e:                                  # noqa: BLE001 -- accounting only, never fatal
            print(f"[bench] step roofline probe failed: {type(e).__name__}: {e}", file=sys.stderr)
            step_roofline = {"error": f"{type(e).__name__}: {e}"}   # the line says so instead of dropping the key
    dp = {"graphed": graphed, "overlap": bool(graphed and overlap and (world > 1 or force_dp)), "dp_group": bool(world > 1 or force_dp),
          "schedule": schedule, "schedule_reason": schedule_reason}
    return elapsed, timer.summary(), float(loss), n_params, dp, step_roofline, host_routes


def parallelism_text(world, graphed, overlap, dp_group, schedule=None, reason=None):
    """`config.parallelism` of the train-step line: what carried the gradients between the ranks and how the model ran; for a
    data-parallel run also how the schedule was chosen (`--dp-schedule auto`: train.choose_dp_schedule's verdict)."""
    how = ""
    if dp_group and schedule is not None:
        how = f"; schedule {schedule}" + (f": {reason}" if reason else "")
    if graphed and dp_group and overlap:
        return (f"dp{world} (bf16 RCCL gradient all-reduce in buckets of arrival order, captured inside the backward graph on a "
                "communication stream: bucket k travels while autograd computes the earlier layers); model forward/backward "
                "replayed as HIP graphs" + how)
    if graphed:
        return (f"dp{world} (one flat bf16 RCCL all-reduce of the gradients after the backward graph); "
                "model forward/backward replayed as HIP graphs" + how)
    return f"dp{world} (DDP: bucketed RCCL gradient all-reduce overlapped with backward); eager launches"


def newest_traffic_file():
    """profiles/r0N_final_traffic.json of the highest round (written by tools/gpu_final_r0N.sh + tools/pmc_final_summary.py)"""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_final_traffic.json")))
    return found[-1] if found else None


def pmc_traffic(kernel_key, args):
    """HBM bytes per launch (fetch + write) of the dominant kernel from the committed rocprofv3 --pmc passes of THIS round's
    kernels (the newest profiles/r0N_final_traffic.json, written by tools/gpu_profiles_r06.sh: separate FETCH_SIZE / WRITE_SIZE runs with
    --kernel-trace only, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes).  A recorded value, not measured by this
    run: it is only reported when the file names the same kernels the call ran (a kernel revision that renames or
    replaces them makes the key `null` instead of quoting stale bytes) and the configuration matches."""
    path = newest_traffic_file()
    if args.dtype != "bf16" or args.batch != 4 or path is None:
        return None
    return traffic_from_table(json.load(open(path)), kernel_key)


def traffic_from_table(t, kernel_key):
    parts = {"enc_bwd_fused": ["msda:cell_backward_kernel+geometry", "msda:patch_dest_kernel"],
             "enc_bwd": ["b0:cell_backward_kernel", "b0:patch_dest_kernel"],
             "enc_fwd_fused": ["fwd:quad_forward_fused_kernel"]}.get(kernel_key)
    if not parts or any(k not in t for k in parts):
        return None
    return int(sum(t[k]["fetch_bytes"] + t[k]["write_bytes"] for k in parts))


def emit(args, world, elapsed, kern, lib, workload_text, parallelism, cpu_calls, probe_steps=None, probe_note=None,
         step_roofline=None, b0=None, host_routes=None):
    probe_steps = args.steps if probe_steps is None else probe_steps
    dominant = max(kern, key=lambda n: kern[n]["ms"])
    kd = kern[dominant]
    mean_s = kd["ms"] / kd["n"] * 1e-3
    achieved = kd["bytes"] / mean_s / 1e9
    variant = kd.get("variant") or lib.msda_variant_name(lib.msda_pick_variant(int(kd["bwd"]), kd["code"], *kd["dims"])).decode()
    images = args.batch * world * args.steps
    line = {
        "metric": METRIC,
        "value": round(images / elapsed, 3),
        "per_gpu": round(images / elapsed / world, 3),
        "unit": "images/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": args.dtype,
        "data": "synthetic",
        "config": {
            "workload": workload_text,
            "global_batch": args.batch * world,
            "batch_per_gpu": args.batch,
            "pyramid": PYRAMID_800x1333,
            "parallelism": parallelism,
            "library_tuning": {
                "miopen": miopen_tuning_report(),
                "hipblaslt": "recorded solution table, lookup only (rlipv2_amd/tuned/gemm_gfx950.csv)"
                             if os.environ.get("RLIPV2_TUNED_GEMMS", "1") != "0" else "library default",
            },
        },
        "roofline": {
            "bound": "hbm",
            "kernel": f"msda_{variant}_{'backward' if kd['bwd'] else 'forward'} ({dominant}: N={kd['dims'][0]}, "
                      f"Lq={kd['dims'][5]}, S={kd['dims'][1]})",
            "achieved": round(achieved, 2),
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 4),
            "traffic": pmc_traffic(dominant, args),
            "traffic_source": "recorded: %s (rocprofv3 --pmc passes of the same kernels, tools/gpu_profiles_r06.sh); null when "
                              "no profile of the kernels that ran is committed" % (
                                  os.path.relpath(newest_traffic_file(), ROOT) if newest_traffic_file() else "no profiles/r0N_final_traffic.json"),
            "algorithmic_bytes_per_launch": kd["bytes"],
            "bytes_definition": "SURVEY.md 8d: value + sampling_loc + attn_weight + grad_out read, grad_value + "
                                "grad_sampling_loc + grad_attn_weight written, each once (bf16 value / grad_out / grad_value, "
                                "float32 locations / weights)" + ("; the kernel that ran is the fused form (geometry backward "
                                "as its epilogue), whose own operands are `operand_bytes_per_launch`" if kd.get("operand_bytes") else ""),
            "operand_bytes_per_launch": kd.get("operand_bytes"),
            "mean_launch_us": round(mean_s * 1e6, 2),
            "timing": probe_note or "HIP events on the launch stream around every call inside the timed region",
            "msda_ms_per_step": round(sum(k["ms"] for k in kern.values()) / max(1, probe_steps), 3),
            "all_kernels": {n: {"mean_us": round(k["ms"] / k["n"] * 1e3, 2), "launches": k["n"],
                                "GBps": round(k["bytes"] / (k["ms"] / k["n"] * 1e-3) / 1e9, 1)}
                            for n, k in sorted(kern.items())},
        },
    }
    if b0 is not None:
        line["roofline"]["b0_signature_kernels"] = b0
    if host_routes is not None:
        line["config"]["host_routes"] = host_routes
    if getattr(args, "overrides", None):
        line["config"]["overrides"] = list(args.overrides)
    if step_roofline is not None and "error" in step_roofline:
        line["step_roofline"] = step_roofline
    elif step_roofline is not None:
        t_mem, t_mfma = step_roofline["T_mem_s"], step_roofline["T_mfma_s"]
        step_s = elapsed / args.steps
        line["step_roofline"] = {
            "T_mem_ms": round(t_mem * 1e3, 3), "T_mfma_ms": round(t_mfma * 1e3, 3),
            "bound": "hbm" if t_mem >= t_mfma else "mfma",
            "achieved_frac_of_max": round(max(t_mem, t_mfma) / step_s, 4),
            "algorithmic_GB_per_step": round(step_roofline["bytes"] / 1e9, 3),
            "matrix_TFLOP_per_step": round(step_roofline["flops"] / 1e12, 3),
            "achieved_TFLOPs": round(step_roofline["flops"] / step_s / 1e12, 1),
            "peaks": {"hbm_GBps": HBM_PEAK_GBPS, "mfma_bf16_TFLOPs": 2500.0},
            "note": "one eager step of the same model/batch under rlipv2_amd.roofline (ATen dispatch byte counter + "
                    "FlopCounterMode + the library kernels' own operand accounting): inputs + outputs of every op "
                    "once, in the step's dtypes; frac = max(T_mem, T_mfma) / measured ms_per_step",
        }
    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(cpu_calls(), "msda_step" if "msda_step" in workload_text[:12] else "train_step")
    # The ONE stdout line the driver parses, as soon as the measurement exists.  Nothing that could lose it runs before it: the
    # experiments leg (opt-in, `--experiments`) starts only after this line has left the process (experiments_leg below).
    print(json.dumps(line), flush=True)
    return line


def miopen_tuning_report():
    """What the line may claim about MIOpen's Find results: "recorded find-db" only if the db directory is ours AND its
    files were recorded for the MIOpen version that is loaded (MIOpen ignores files of another version without a word)."""
    import rlipv2_amd
    if os.environ.get("RLIPV2_MIOPEN_FIND", "0") == "1":
        return "find at first use (RLIPV2_MIOPEN_FIND=1)"
    st = rlipv2_amd.miopen_db_status()
    if not st["path"] or "rlipv2_miopen_db" not in st["path"]:
        return "library default (no recorded find-db in use)"
    if st["version_match"]:
        return f"recorded find-db, lookup only (rlipv2_amd/tuned/miopen, recorded for MIOpen {st['recorded_for']} = the loaded library)"
    return (f"library default: the recorded find-db is for MIOpen {st['recorded_for']}, the loaded library is {st['library']} "
            "-- MIOpen ignores it (expect ~2 ms/step of split-K workspace kernels)")


def under_profiler():
    return any("rocprof" in (os.environ.get(k) or "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB",
                                                                           "ROCPROFILER_REGISTER_LIBRARY"))


EXPERIMENTS_BUDGET_S = 150          # tools/experiments_r05.py starts no child after this many seconds ...
EXPERIMENTS_GRACE_S = 60            # ... and the whole leg (a process group of its own) is killed this long after that


def experiments_applicable(args, world):
    """the leg is opt-in (`--experiments`) and only meaningful next to the full default evidence run: 1 GPU, R50, bf16, batch 4,
    no overrides / variants; never under rocprofv3 (its preload follows children)"""
    return bool(args.experiments and not args.no_cpu_baseline and world == 1 and args.backbone == "resnet50" and args.dtype == "bf16"
                and args.batch == 4 and not args.overrides and not args.padded and not args.var_targets and not under_profiler())


def experiments_leg(args, world, out_path=None, cmd=None, budget_s=None):
    """AFTER the stdout line (nothing of this run is measured any more, nothing printed here is parsed by the driver): the A/B
    table of the kernel arms that have never been timed on hardware -- tools/experiments_r05.py as a child PROCESS GROUP (every arm
    a grandchild with its own timeout), killed as a group when it overruns.  The object goes to stderr as one `EXPERIMENTS {json}`
    line and to gpurun_out/experiments_last.json; tools/promote_r05.py reads either.  Never raises."""
    import signal
    import subprocess
    if not experiments_applicable(args, world):
        return None
    budget_s = (EXPERIMENTS_BUDGET_S + EXPERIMENTS_GRACE_S) if budget_s is None else budget_s
    cmd = cmd or [sys.executable, os.path.join(ROOT, "tools", "experiments_r05.py")]
    out_path = out_path or os.path.join(ROOT, "gpurun_out", "experiments_last.json")
    try:
        if torch.cuda.is_initialized():
            torch.cuda.synchronize()
            torch.cuda.empty_cache()                              # the grandchildren allocate on the same device
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
        proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env, cwd=ROOT,
                                start_new_session=True)
        try:
            out, _ = proc.communicate(timeout=budget_s)
            rep = None
            for ln in reversed(out.splitlines()):
                if ln.startswith("{"):
                    rep = json.loads(ln)
                    break
            if rep is None:
                rep = {"error": f"rc {proc.returncode}: no JSON object on the leg's stdout"}
        except subprocess.TimeoutExpired:
            try:
                os.killpg(proc.pid, signal.SIGKILL)               # the leg and every arm it started
            except ProcessLookupError:
                pass
            proc.communicate()
            rep = {"error": f"timed out after {budget_s} s (process group killed)"}
    except Exception as e:                                        # noqa: BLE001 -- evidence only, never fatal
        rep = {"error": f"{type(e).__name__}: {e}"}
    try:
        print("EXPERIMENTS " + json.dumps(rep), file=sys.stderr, flush=True)
        os.makedirs(os.path.dirname(out_path), exist_ok=True)
        with open(out_path, "w") as f:
            json.dump(rep, f)
    except Exception as e:                                        # noqa: BLE001
        print(f"[bench] could not record the experiments object: {type(e).__name__}: {e}", file=sys.stderr)
    return rep


# ---------------------------------------------------------------------------------------------------------------------------------
# host-route self-check in a child process (rlipv2_amd/routes.py; VERDICT round 5, "next round" item 1)

ROUTES_CHILD_TIMEOUT_S = 300
_DIST_ENV = ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "ROLE_WORLD_SIZE", "GROUP_WORLD_SIZE",
             "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT", "TORCHELASTIC_MAX_RESTARTS",
             "TORCHELASTIC_USE_AGENT_STORE", "TORCHELASTIC_ERROR_FILE", "RLIPV2_FORCE_DP")


def applicable_routes(backbone):
    """the Swin routes cannot apply to a ResNet-50 step: not exercised there (two eager steps saved), reported as such"""
    from rlipv2_amd import routes
    swin_only = ("fused_wide_layer_norm", "fused_window_attention")
    return [n for n in routes.GPU_ONLY_ROUTES if backbone.startswith("swin") or n not in swin_only]


def routes_child_main(args):
    """`bench.py --routes-child`: build the very model and batch the timed run will use (same seeds: rank 0's weights -- what every
    rank holds after the broadcast --, this rank's images), run routes.validate on the applicable routes and print one
    `ROUTES {json}` line.  A single process on one device: no process group, no optimiser, nothing timed."""
    from rlipv2_amd import parseda, routes, train
    rank = args.routes_child_rank
    if args.routes_child_device < 0:                 # (CPU dry run of the plumbing, tests/test_bench_host.py: every route "off (not applicable)")
        device = "cpu"
    else:
        torch.cuda.set_device(args.routes_child_device)
        device = f"cuda:{args.routes_child_device}"
    margs = parseda.default_args(num_queries=args.queries)
    if os.environ.get("RLIPV2_MIOPEN_FIND", "0") == "1":
        torch.backends.cudnn.benchmark = True
    torch.manual_seed(0)
    model, criterion = train.build_training(margs, device=device, with_text_encoder=True, backbone_name=args.backbone)
    master = args.dtype == "bf16" and args.precision == "master"
    if master:
        train.to_bf16(model)
    sizes = [(800, 1333), (736, 1100)] if args.padded else None
    batch = train.synthetic_batch(args.batch, 800, 1333, n_obj=43, n_verb=21, triplets=8, device=device, seed=rank, sizes=sizes)
    if master:
        batch[0].tensors = batch[0].tensors.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    step_module = train.ParSeDATrainStep(model)
    model.train()
    dtype = torch.bfloat16 if (args.dtype == "bf16" and not master) else None
    if device != "cpu":                              # (the dry run of the freeze is a whole forward + backward: nothing for it to decide on the CPU)
        train.freeze_parameters_without_gradient(step_module, criterion, batch, autocast_dtype=dtype)
    verdict = routes.validate(step_module, criterion, batch, autocast_dtype=dtype, names=applicable_routes(args.backbone),
                              log=lambda m: print(m, file=sys.stderr))
    if device != "cpu":
        torch.cuda.synchronize()
    print("ROUTES " + json.dumps({n: verdict[n] for n in applicable_routes(args.backbone)}), flush=True)


def routes_verdicts_from_child(argv, backbone, rank=0, device_index=0, timeout=ROUTES_CHILD_TIMEOUT_S, cmd=None):
    """Parent side, called BEFORE this process initialises the GPU: start `bench.py --routes-child` with the same arguments, wait at
    most `timeout` s, read its `ROUTES {json}` line.  -> {route: "on" | "off (...)"} for every route of routes.GPU_ONLY_ROUTES.
    A child that dies on a signal (a GPU memory fault is SIGSEGV / SIGABRT, not a Python exception), hangs, or prints no verdict
    leaves every route off; so does a verdict that names an unknown route or says anything but "on" / "off (...)"."""
    import signal
    import subprocess
    from rlipv2_amd import routes
    names = list(routes.GPU_ONLY_ROUTES)
    todo = applicable_routes(backbone)
    result = {n: "off (not applicable: no such block in a %s step)" % backbone for n in names if n not in todo}

    def all_off(why):
        result.update({n: f"off (self-check child: {why})" for n in todo})
        return result
    cmd = cmd or [sys.executable, os.path.abspath(__file__), *argv, "--routes-child", "--routes-child-rank", str(rank),
                  "--routes-child-device", str(device_index)]
    env = {k: v for k, v in os.environ.items() if k not in _DIST_ENV}
    try:
        proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=ROOT, start_new_session=True)
    except OSError as e:
        return all_off(f"not started, {type(e).__name__}: {e}")
    try:
        out, err = proc.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(proc.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        proc.communicate()
        return all_off(f"timed out after {timeout} s, killed")
    for ln in (err or "").splitlines():
        if ln.startswith("[routes]"):
            print(ln, file=sys.stderr)
    if proc.returncode != 0:
        how = f"signal {signal.Signals(-proc.returncode).name}" if proc.returncode < 0 else f"exit code {proc.returncode}"
        return all_off(f"{how}; {(err or '').strip().splitlines()[-1][:160] if (err or '').strip() else 'no message'}")
    verdict = None
    for ln in reversed(out.splitlines()):
        if ln.startswith("ROUTES "):
            try:
                verdict = json.loads(ln[7:])
            except ValueError:
                verdict = None
            break
    if not isinstance(verdict, dict) or set(verdict) != set(todo) or not all(
            isinstance(v, str) and (v == "on" or v.startswith("off (")) for v in verdict.values()):
        return all_off("no well-formed verdict on its stdout")
    result.update(verdict)
    return {n: result[n] for n in names}


def apply_route_verdicts(verdicts, world, device):
    """switch on exactly the routes whose verdict is "on" -- on every rank (MIN all-reduce), or on none"""
    from rlipv2_amd import routes
    names = list(routes.GPU_ONLY_ROUTES)
    if verdicts is None:
        verdicts = {n: "off (no self-check was run)" for n in names}
    ok = [1 if verdicts.get(n) == "on" else 0 for n in names]
    if world > 1 and dist.is_initialized():
        t = torch.tensor(ok, dtype=torch.int32, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        agreed = t.tolist()
    else:
        agreed = ok
    out = {}
    for n, mine, both in zip(names, ok, agreed):
        routes.set_route(n, bool(both))
        out[n] = verdicts.get(n, "off (no verdict)") if (both or not mine) else "off (self-check failed on another rank)"
    return out


def apply_overrides(overrides):
    """`--set module.attr=value`: flip an existing bool / int switch of a rlipv2_amd module (A/B runs)."""
    import importlib
    done = {}
    for item in overrides:
        name, _, value = item.partition("=")
        mod, _, attr = name.rpartition(".")
        m = importlib.import_module("rlipv2_amd." + mod)
        old = getattr(m, attr, None)
        if not isinstance(old, (bool, int)) or value == "":
            raise SystemExit(f"--set {item}: rlipv2_amd.{mod}.{attr} is not a bool / int switch")
        new = (value.lower() in ("1", "true", "on")) if isinstance(old, bool) else int(value)
        setattr(m, attr, new)
        done[name] = new
    return done


def self_launch(n_gpus):
    """`python bench.py --gpus N` without a launcher: run `python -m torch.distributed.run --nproc-per-node N bench.py <same
    arguments>` as a child process and return its exit code.  Counting devices does not initialise the GPU on this
    image (torch.cuda.device_count() reads the driver's device list)."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n_gpus and os.environ.get("RLIPV2_SINGLE_DEVICE") != "1":
        print(f"[bench] --gpus {n_gpus} but this node exposes {have} GPU(s)", file=sys.stderr)
        return 2
    with socket.socket() as sock:                        # a free rendezvous port on the loopback interface
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC: what RCCL needs between the ranks on this host
    print("[bench] launching", " ".join(cmd), file=sys.stderr)
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--no-graph-criterion", dest="graph_criterion", action="store_false",
                    help="keep the criterion's device work eager (outside the HIP graphs)")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4, help="images per GPU (BASELINE config 2: 4)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--workload", default="train_step", choices=["train_step", "msda_step"])
    ap.add_argument("--queries", type=int, default=300)
    ap.add_argument("--backbone", default="resnet50", choices=["resnet50", "swin_tiny", "swin_large"],
                    help="resnet50 = BASELINE config 2 (the metric's configuration); swin_large = config 4 (use --batch 2)")
    ap.add_argument("--no-graph", dest="graph", action="store_false",
                    help="do not capture the model phases as HIP graphs (single-GPU runs capture by default)")
    ap.add_argument("--precision", default="master", choices=["master", "autocast"],
                    help="bf16 policy: bf16 parameters + float32 master weights (default) or torch.autocast")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--experiments", action="store_true",
                    help="AFTER the JSON line has been printed: the A/B table of the kernel arms that have never been timed on "
                         "hardware (tools/experiments_r05.py in a process group of its own, at most ~3.5 min) -> one `EXPERIMENTS "
                         "{json}` line on stderr + gpurun_out/experiments_last.json.  Off by default: the metric run touches product "
                         "kernels only")
    ap.add_argument("--no-experiments", dest="experiments", action="store_false", help="(the default; kept for older scripts)")
    ap.add_argument("--routes-child", action="store_true", help=argparse.SUPPRESS)     # internal: the host-route self-check process
    ap.add_argument("--routes-child-rank", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--routes-child-device", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--set", action="append", default=[], metavar="MODULE.ATTR=VALUE", dest="overrides",
                    help="A/B runs: set a bool / int attribute of a rlipv2_amd module for this run, e.g. --set decoder.fused_glue=0 "
                         "(the switches INTEGRATION.md lists; recorded in config.overrides)")
    ap.add_argument("--host-routes", default="auto", choices=["auto", "off", "on"],
                    help="GPU-only host routes of the train step (rlipv2_amd/routes.py): auto = switch on the ones that pass the "
                         "start-up self-check against the plain step, run in a child process before this process touches the "
                         "GPU (default), off / on = forced")
    ap.add_argument("--msda-fwd-cell", action="store_true",
                    help="EXPERIMENT: the encoder's fused MSDA forward through cell_forward_kernel (LDS windows + matrix cores); "
                         "not validated on hardware, never the default")
    ap.add_argument("--dp-schedule", default="auto", choices=["auto", "flat", "overlapped"],
                    help="data-parallel runs, how the gradients travel: overlapped = bucketed all-reduce captured inside the "
                         "backward graph (reference: DDP, main.py:515-517), flat = one all-reduce after it, auto (default) = "
                         "overlapped iff captured collectives replay on all ranks and the first overlapped step's loss and "
                         "gradient norm equal the flat step's to 1e-3, else flat; the choice is reported in config.parallelism")
    ap.add_argument("--dp-overlap", action="store_true", help="the same as --dp-schedule overlapped (older scripts)")
    ap.add_argument("--deterministic", action="store_true",
                    help="MIOpen restricted to deterministic convolution solvers (torch.backends.cudnn.deterministic): the one "
                         "source of run-to-run noise in the step is a MIOpen convolution (tools/nondet_modules.py)")
    ap.add_argument("--padded", action="store_true",
                    help="non-best-case variant: images of (800,1333) and (736,1100) padded into one batch, mask path live")
    ap.add_argument("--var-targets", action="store_true",
                    help="non-best-case variant: 6 / 8 / 11 triplets per image in rotation (one graph capture per bucket)")
    args = ap.parse_args()

    if args.routes_child:
        # internal: the host-route self-check of one rank (started by routes_verdicts_from_child with the parent's own arguments,
        # torch.distributed.run's variables removed): a single process, whatever --gpus says
        if args.msda_fwd_cell:
            from rlipv2_amd import msda as _msda
            _msda.fused_forward_cell = True
        apply_overrides(args.overrides)
        if args.deterministic:
            torch.backends.cudnn.deterministic = True
        if torch.cuda.device_count() == 0 and args.routes_child_device >= 0:
            raise SystemExit("bench.py --routes-child needs a GPU")
        routes_child_main(args)
        return
    if args.gpus > 1 and "RANK" not in os.environ:
        # Plain `python bench.py --gpus N`: launch the N ranks ourselves (the reference launches with
        # torch.distributed.launch --nproc_per_node, scripts/RLIP_ParSeDA/train_RLIP_ParSeDA_v2_mixed_vgcoco_resnet.sh:1-2).
        # The launcher is a CHILD process started before this process touches the GPU (no HIP call, no
        # torch.cuda.is_available() above this line): a process that has initialised the GPU must never exec or be
        # replaced.  Rank 0's JSON line passes through on stdout; the exit code is the launcher's.
        raise SystemExit(self_launch(args.gpus))
    if args.msda_fwd_cell:
        from rlipv2_amd import msda as _msda
        _msda.fused_forward_cell = True
    apply_overrides(args.overrides)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run "
                         f"--nproc-per-node {args.gpus}, or without RANK in the environment to let bench.py launch its ranks")
    if torch.cuda.device_count() == 0:            # (reads the driver's device list; does not initialise the GPU on this image)
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    if args.workload == "train_step" and args.host_routes == "auto":
        # the self-check runs in a child BEFORE this process creates its GPU context: a never-run route that faults or hangs
        # takes the child with it, never the timed region or the JSON line (the child is a plain subprocess, not an exec)
        # (not under rocprofv3: its preloaded tool has initialised the GPU in this process already and follows children into the
        #  trace -- the profiled run is the plain step, and says so)
        args.route_verdicts = ({n: "off (self-check child not started under a profiler)" for n in applicable_routes(args.backbone)}
                               if under_profiler() else None) or routes_verdicts_from_child(
            list(sys.argv[1:]), args.backbone, rank=rank, device_index=0 if os.environ.get("RLIPV2_SINGLE_DEVICE") == "1" else local_rank)
        print("[bench] host routes (self-check in a child process): " + json.dumps(args.route_verdicts), file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    if args.deterministic:
        torch.backends.cudnn.deterministic = True
    # Development aid for boxes with ONE GPU: RLIPV2_SINGLE_DEVICE=1 puts every rank on cuda:0 and
    # RLIPV2_DIST_BACKEND=gloo carries the collectives over the host (RCCL refuses two ranks per device), so that
    # the multi-rank control flow (broadcast, capture, synchroniser, collectives) runs for real.  Not a benchmark.
    if os.environ.get("RLIPV2_SINGLE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = f"cuda:{local_rank}"
    if world > 1 or (os.environ.get("RLIPV2_FORCE_DP") == "1" and "RANK" in os.environ):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("RLIPV2_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(device))
        else:
            dist.init_process_group(backend)

    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    lib = _lib.lib()
    if args.workload == "train_step":
        elapsed, kern, loss, n_params, dp, step_roofline, host_routes = run_train_step_bench(args, world, rank, local_rank, device)
        graphed = dp["graphed"]
        if rank == 0:
            emit(args, world, elapsed, kern, lib, workload_text=(
                "train_step: RLIP_ParSeDA_v2 " + {"resnet50": "R50", "swin_large": "Swin-L", "swin_tiny": "Swin-T"}[args.backbone]
                + " 4-scale %d-query train step (fwd phase A+B, SetCriterionHOI, bwd, "
                "clip 0.1, AdamW), batch %d/GPU, 800x1333, 64 relation/object texts, RoBERTa-base-shaped text "
                "encoder in the step, %s; %.1f M trainable parameters; "
                "final loss %.4f" % (args.queries, args.batch,
                                     ("bf16 parameters/activations/gradients with float32 master weights" if args.precision == "master"
                                      else "bf16 autocast over float32 weights") if args.dtype == "bf16" else "float32",
                                     n_params / 1e6, loss)
                + ("; VARIANT padded batch: images of 800x1333 and 736x1100 padded together, padding masks live" if args.padded else "")
                + ("; VARIANT variable targets: 8 / 6 / 11 triplets per image in rotation, one graph capture per bucket" if args.var_targets else "")
                + ("; VARIANT eager launches (no HIP graphs)" if not args.graph else "")),
                 parallelism=parallelism_text(world, dp["graphed"], dp["overlap"], dp["dp_group"], dp["schedule"], dp["schedule_reason"]),
                 cpu_calls=lambda: build_msda_step(1, torch.float32, device, 0),
                 probe_steps=2 if graphed else None,
                 probe_note=("HIP events around every MSDA call in 2 eager steps of the same train step run right after "
                             "the timed region (the timed steps replay HIP graphs, whose kernels cannot be bracketed)")
                 if graphed else None, step_roofline=step_roofline, host_routes=host_routes,
                 b0=b0_signature_kernels(args.batch, dtype, device) if args.backbone == "resnet50" else None)
        if world > 1:
            dist.destroy_process_group()
        if rank == 0:
            experiments_leg(args, world)             # opt-in; after the line, in a process group of its own
        return
    calls = build_msda_step(args.batch, dtype, device, rank)
    stream = torch.cuda.current_stream().cuda_stream

    for _ in range(args.warmup):
        run_step(calls, lib, stream, timed=False)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run_step(calls, lib, stream, timed=True)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        kern = {}
        for c in calls:
            kind = "enc" if c.name.startswith("enc") else ("dec300" if c.name.startswith("ho") else "dec150")
            for d in ("fwd", "bwd"):
                ms = [a.elapsed_time(b) for a, b in c.ev[d]]
                k = kern.setdefault(f"{kind}_{d}", {"ms": 0.0, "n": 0, "bytes": c.bytes_fwd if d == "fwd" else c.bytes_bwd,
                                                    "dims": c.dims, "code": c.code, "bwd": d == "bwd",
                                                    "variant": c.variant_bwd if d == "bwd" else None})
                k["ms"] += sum(ms)
                k["n"] += len(ms)
        emit(args, world, elapsed, kern, lib, workload_text=(
            "msda_step: the 12 fwd + 12 bwd MSDeformAttn launches of one RLIP_ParSeDA_v2 R50 train step "
            "(6 encoder Lq=S=22223, 3 ho-decoder Lq=300, 3 verb-decoder Lq=150), 4-level 800x1333 pyramid, "
            "M8 D32 L4 P4; dense blocks not included"),
             parallelism=f"dp{world} (independent image shards, no collective in this workload)",
             cpu_calls=lambda: calls)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
