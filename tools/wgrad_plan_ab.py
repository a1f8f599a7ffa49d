"""Decides brief item 6 (single-launch weight gradient for the small-token Linears) with ONE measurement: for the decoder / text
shapes of the train step, linear_wgrad as the plan runs it today (2-4 chunks + the reduction launch) against ONE chunk with the
direct-store instantiation (no second launch; exists and is validated for token counts that are multiples of 32 -- so the token
counts here are, and what a tail-masking variant would gain is what this prints).  100 calls captured into one HIP graph per
setting, so that launch overheads count the way they do in the train step.
usage (GPU box, ablation build: the plan's knob is read from the environment there):
    for m in 8 100000; do RLIPV2_WGRAD_MINSTEPS=$m RLIPV2_LIB_PATH=tools/_build/librlipv2_msda_ablation.so python tools/wgrad_plan_ab.py; done"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlipv2_amd import linear  # noqa: E402

SHAPES = [(608, 256, 256), (1216, 256, 256), (1216, 2048, 256), (1216, 256, 2048), (608, 2048, 256), (320, 768, 768),
          (1088, 256, 256), (1216, 384, 256)]
REPS = 100


def main():
    tag = os.environ.get("RLIPV2_WGRAD_MINSTEPS", "(default 8)")
    print(f"# RLIPV2_WGRAD_MINSTEPS = {tag}   ({REPS} calls per graph replay, us per call)")
    for T, M, K in SHAPES:
        dy = torch.randn(T, M, device="cuda").to(torch.bfloat16)
        x = torch.randn(T, K, device="cuda").to(torch.bfloat16)
        for _ in range(3):
            dw, db = linear.linear_wgrad(dy, x)
        torch.cuda.synchronize()
        ref = dy.float().t() @ x.float()
        err = float((dw.float() - ref).abs().max() / ref.abs().max())
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            with torch.cuda.graph(g):
                for _ in range(REPS):
                    linear.linear_wgrad(dy, x)
        torch.cuda.synchronize()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            g.replay()
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / 10 / REPS * 1e6
        print(f"T={T:5d} M={M:5d} K={K:5d}   {us:7.2f} us per call   max rel err {err:.1e}", flush=True)


if __name__ == "__main__":
    main()
