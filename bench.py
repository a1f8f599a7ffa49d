#!/usr/bin/env python3
"""bench.py -- the driver's benchmark contract for the RLIPv2-ParSeDA hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch of synthetic input.  The workload is
BASELINE.json config 2 (RLIP_ParSeDA_v2 R50, 4 levels, 300 queries, bf16, batch 4 per GPU,
800x1333 images -> pyramid 100x167 / 50x84 / 25x42 / 13x21, 64 text tokens).

Workloads (``--workload``)
  msda_step  (default this round): every multi-scale-deformable-attention call of one train step
             -- 6 encoder self-attention (Lq = S = 22223), 3 human-object decoder (Lq = 300) and
             3 verb decoder (Lq = 150) cross-attention calls, forward then backward, i.e. the
             12 + 12 launches SURVEY.md section 1 counts per step -- on synthetic model-like
             sampling locations (tools/msda_inputs.py).  The dense blocks around them are not
             in this workload; `config.workload` says so.

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
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from rlipv2_amd import _lib  # noqa: E402
from tools.msda_inputs import PYRAMID_800x1333, make_inputs  # noqa: E402

HBM_PEAK_GBPS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec peak
METRIC = "images/sec RLIPv2-ParSeDA R50 train step (MSDeformAttn HBM GB/s in `roofline`)"


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
        self.g_value = torch.empty(N, S, M, D, dtype=torch.float32, device=device)
        self.g_loc = torch.empty_like(inp["loc"])
        self.g_aw = torch.empty_like(inp["aw"])
        self.bytes_fwd = _lib.algorithmic_bytes(self.code, False, *self.dims)
        self.bytes_bwd = _lib.algorithmic_bytes(self.code, True, *self.dims)
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
        self.g_value.zero_()      # the reference's at::zeros_like; kept outside the event bracket
        if timed:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
        st = lib.msda_backward_ex(_lib.VARIANT_AUTO | 0x100, self.code, i["value"].data_ptr(),
                                  i["shapes"].data_ptr(), i["starts"].data_ptr(), i["loc"].data_ptr(),
                                  i["aw"].data_ptr(), i["grad_out"].data_ptr(), *self.dims,
                                  self.g_value.data_ptr(), self.g_loc.data_ptr(), self.g_aw.data_ptr(), stream)
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


def cpu_baseline(calls):
    """Oracle (OpenMP C port) on one image of the batch through all 24 calls of a step."""
    import numpy as np

    from oracle import msda_oracle as O
    O.build()
    t_total, images = 0.0, 0
    nb = calls[0].dims[0]
    while t_total < 10.0 and images < 4 * nb:          # bounded: at least ~10 s of CPU work or 4 batches
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
    return {"value": round(images / t_total, 4), "unit": "images/s", "cores": O.threads(True), "kind": "port",
            "sample": "%d image(s) through all 12 fwd + 12 bwd MSDA calls of one step, float32, "
                      "oracle/msda_oracle.c built with OpenMP; %.1f s of CPU work" % (images, t_total)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4, help="images per GPU (BASELINE config 2: 4)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--workload", default="msda_step", choices=["msda_step"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = f"cuda:{local_rank}"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device(device))

    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    lib = _lib.lib()
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
        # per-kernel means from the HIP events recorded inside the timed region
        kern = {}
        for c in calls:
            kind = "enc" if c.name.startswith("enc") else ("ho_dec" if c.name.startswith("ho") else "verb_dec")
            for d in ("fwd", "bwd"):
                ms = [a.elapsed_time(b) for a, b in c.ev[d]]
                k = kern.setdefault(f"{kind}_{d}", {"ms": 0.0, "n": 0, "bytes": c.bytes_fwd if d == "fwd" else c.bytes_bwd,
                                                    "dims": c.dims, "code": c.code, "bwd": d == "bwd"})
                k["ms"] += sum(ms)
                k["n"] += len(ms)
        dominant = max(kern, key=lambda n: kern[n]["ms"])
        kd = kern[dominant]
        mean_s = kd["ms"] / kd["n"] * 1e-3
        achieved = kd["bytes"] / mean_s / 1e9
        variant = lib.msda_variant_name(lib.msda_pick_variant(int(kd["bwd"]), kd["code"], *kd["dims"])).decode()
        images = args.batch * world * args.steps
        line = {
            "metric": METRIC,
            "value": round(images / elapsed, 3),
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
                "workload": "msda_step: the 12 fwd + 12 bwd MSDeformAttn launches of one RLIP_ParSeDA_v2 R50 "
                            "train step (6 encoder Lq=S=22223, 3 ho-decoder Lq=300, 3 verb-decoder Lq=150), "
                            "4-level 800x1333 pyramid, M8 D32 L4 P4; dense blocks not included",
                "global_batch": args.batch * world,
                "batch_per_gpu": args.batch,
                "pyramid": PYRAMID_800x1333,
                "parallelism": f"dp{world} (independent image shards, no collective in this workload)",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": f"msda_{variant}_{'backward' if kd['bwd'] else 'forward'} (encoder shape, {dominant})",
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4),
                "traffic": None,
                "algorithmic_bytes_per_launch": kd["bytes"],
                "mean_launch_us": round(mean_s * 1e6, 2),
                "all_kernels": {n: {"mean_us": round(k["ms"] / k["n"] * 1e3, 2),
                                    "GBps": round(k["bytes"] / (k["ms"] / k["n"] * 1e-3) / 1e9, 1)}
                                for n, k in sorted(kern.items())},
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(calls)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
