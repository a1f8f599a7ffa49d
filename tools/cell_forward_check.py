"""The experimental forward variant "cell" (LDS windows + matrix cores, csrc/msda_patch.hip: cell_forward_kernel) against
the product forward kernel on the encoder shape: results compared, both timed with HIP events.
    python tools/cell_forward_check.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlipv2_amd import msda  # noqa: E402
from tools.msda_inputs import make_inputs  # noqa: E402
from tools.patch_check import timed  # noqa: E402


def main():
    for N, mode in ((1, "model"), (4, "model"), (4, "init"), (4, "uniform")):
        inp = make_inputs(N, mode=mode, dtype=torch.bfloat16, seed=3)
        a = (inp["value"], inp["shapes"], inp["starts"], inp["loc"], inp["aw"])
        res, t = {}, {}
        for v in ("quad", "cell"):
            msda.set_variant(v, "quad")
            try:
                res[v] = msda.ms_deform_attn_forward(*a, 64).float()
                torch.cuda.synchronize()
                t[v] = timed(lambda: msda.ms_deform_attn_forward(*a, 64), iters=20)
            except RuntimeError as e:
                print(f"N={N} {mode}: {v} failed: {e}")
            finally:
                msda.set_variant("auto")
        if "cell" in res:
            d = (res["cell"] - res["quad"]).abs()
            ref = res["quad"].abs().max().item()
            print(f"N={N} {mode:8s} quad {t['quad']:7.1f} us | cell {t['cell']:7.1f} us | max |diff| / max |ref| = "
                  f"{d.max().item() / ref:.3e}, mean {d.mean().item() / ref:.3e}, non-finite "
                  f"{int((~torch.isfinite(res['cell'])).sum())}", flush=True)


if __name__ == "__main__":
    main()
