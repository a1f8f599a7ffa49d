"""A/B of the patches-per-wave experiment of patch_dest_kernel (RLIPV2_PATCH_REPS, ablation build only): grad_value of the
whole backward compared bit for bit with the default (1 patch per wave), both timed.
    RLIPV2_LIB_PATH=$PWD/tools/_build/librlipv2_msda_ablation.so python tools/reps_check.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlipv2_amd import msda  # noqa: E402
from tools.msda_inputs import make_inputs  # noqa: E402
from tools.patch_check import timed  # noqa: E402


def main():
    for N, mode in ((4, "model"), (4, "init"), (1, "uniform")):
        inp = make_inputs(N, mode=mode, dtype=torch.bfloat16, seed=5)
        a = (inp["value"], inp["shapes"], inp["starts"], inp["loc"], inp["aw"], inp["grad_out"])
        base = None
        for reps in ("1", "2", "3", "4", "8"):
            os.environ["RLIPV2_PATCH_REPS"] = reps
            out = [t.float() for t in msda.ms_deform_attn_backward(*a, 64)]
            torch.cuda.synchronize()
            t = timed(lambda: msda.ms_deform_attn_backward(*a, 64), iters=20)
            if base is None:
                base = out
            same = all(torch.equal(x, y) for x, y in zip(out, base))
            print(f"N={N} {mode:8s} reps={reps}: whole backward {t:8.1f} us, equal bits with reps=1: {same}", flush=True)
        os.environ["RLIPV2_PATCH_REPS"] = "1"


if __name__ == "__main__":
    main()
