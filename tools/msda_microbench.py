"""Kernel-level timing of the MSDA variants on one GPU (not the judged bench; see bench.py).

usage: python tools/msda_microbench.py [--iters 20] [--out gpurun_out/microbench.json]
Prints, per (shape, dtype, input mode, variant, direction): mean kernel time (HIP events on the
launch stream), algorithmic GB/s and the fraction of the 8 TB/s HBM peak.
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlipv2_amd import _lib, msda  # noqa: E402
from tools.msda_inputs import make_inputs  # noqa: E402

HBM_PEAK = 8.0e12


def time_call(fn, iters, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--out", default="gpurun_out/microbench.json")
    ap.add_argument("--variants", default="generic,quad,window")
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--skip-quad-bwd", action="store_true", default=True)
    args = ap.parse_args()
    L = _lib.lib()
    rows = []
    cases = [("enc", 4, None, "model"), ("enc", 4, None, "uniform"), ("dec300", 4, 300, "decoder"),
             ("dec150", 4, 150, "decoder")]
    if not args.quick:
        cases += [("enc", 1, None, "model"), ("enc", 8, None, "model")]
    for name, N, Lq, mode in cases:
        for dtype in (torch.float32, torch.bfloat16):
            inp = make_inputs(N, Lq=Lq, mode=mode, dtype=dtype)
            dims = inp["dims"]
            code = _lib.MSDA_F32 if dtype == torch.float32 else _lib.MSDA_BF16
            for variant in args.variants.split(","):
                if variant == "generic" and name == "enc" and N > 4:
                    continue
                if variant == "quad" and name == "enc" and args.skip_quad_bwd:
                    pass
                msda.set_variant(variant)
                a = (inp["value"], inp["shapes"], inp["starts"], inp["loc"], inp["aw"])
                times = {}
                try:
                    try:
                        times["fwd"] = time_call(lambda: msda.ms_deform_attn_forward(*a, 64), args.iters)
                    except RuntimeError:
                        pass
                    if not (variant == "quad" and name == "enc" and args.skip_quad_bwd):
                        try:
                            times["bwd"] = time_call(lambda: msda.ms_deform_attn_backward(*a, inp["grad_out"], 64),
                                                     args.iters)
                        except RuntimeError:
                            pass
                finally:
                    msda.set_variant("auto")
                for direction, t in times.items():
                    nbytes = _lib.algorithmic_bytes(code, direction == "bwd", *dims)
                    row = dict(case=name, N=N, mode=mode, dtype=str(dtype).split(".")[-1], variant=variant,
                               dir=direction, us=t * 1e6, alg_MB=nbytes / 1e6, GBps=nbytes / t / 1e9,
                               frac_hbm=nbytes / t / HBM_PEAK)
                    rows.append(row)
                    print("{case:7s} N={N} {mode:8s} {dtype:9s} {variant:8s} {dir} {us:10.1f} us  {GBps:8.1f} GB/s  "
                          "{frac_hbm:6.1%}".format(**row), flush=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(rows, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
