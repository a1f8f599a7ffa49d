"""A/B of the matrix-core patch pass (csrc/msda_patch.hip) against the sorting pass (csrc/msda_dest.hip) on the encoder
shape: same inputs, grad_value compared, both timed.  Needs the ablation build (RLIPV2_MSDA_PATCH is read there only):
    make -C rlipv2_amd/csrc ablation && RLIPV2_LIB_PATH=$PWD/tools/_build/librlipv2_msda_ablation.so python tools/patch_check.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlipv2_amd import msda  # noqa: E402
from tools.msda_inputs import make_inputs  # noqa: E402


def timed(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    for N, mode in ((1, "model"), (4, "model"), (4, "uniform")):
        inp = make_inputs(N, mode=mode, dtype=torch.bfloat16, seed=3)
        a = (inp["value"], inp["shapes"], inp["starts"], inp["loc"], inp["aw"], inp["grad_out"])
        res = {}
        for patch in ("0", "1"):
            os.environ["RLIPV2_MSDA_PATCH"] = patch
            gv = msda.ms_deform_attn_backward(*a, 64)[0].float()
            torch.cuda.synchronize()
            res[patch] = gv
            t = timed(lambda: msda.ms_deform_attn_backward(*a, 64))
            print(f"N={N} {mode:8s} patch={patch}: whole backward {t:8.1f} us", flush=True)
        d = (res["0"] - res["1"]).abs()
        ref = res["0"].abs().max().item()
        print(f"   grad_value: max |diff| / max |ref| = {d.max().item() / ref:.3e}, mean |diff| / max = {d.mean().item() / ref:.3e}, "
              f"rows differing {int((d > 0).any(-1).sum())} of {d.shape[0] * d.shape[1] * d.shape[2]}, equal bits: {bool((d == 0).all())}")
        os.environ["RLIPV2_MSDA_PATCH"] = "1"
        again = msda.ms_deform_attn_backward(*a, 64)[0].float()
        print(f"   patch pass repeatable bit for bit: {bool(torch.equal(again, res['1']))}")


if __name__ == "__main__":
    main()
