"""A/B of cell_backward_kernel (K1's work from LDS windows on 4x4x4 MFMAs + binning) against round 2's K1 + bin2_kernel
on the encoder shape (ablation build: RLIPV2_MSDA_CELL is read there only)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlipv2_amd import msda  # noqa: E402
from tools.msda_inputs import make_inputs  # noqa: E402
from tools.patch_check import timed  # noqa: E402


def main():
    cases = [(1, "model", [(25, 34), (13, 17), (7, 9), (4, 5)]), (4, "init", None), (4, "model", None), (4, "uniform", None)]
    for N, mode, pyr in cases:
        kw = dict(pyramid=pyr) if pyr else {}
        inp = make_inputs(N, mode=mode, dtype=torch.bfloat16, seed=3, **kw)
        a = (inp["value"], inp["shapes"], inp["starts"], inp["loc"], inp["aw"], inp["grad_out"])
        res = {}
        for cell in ("0", "1"):
            os.environ["RLIPV2_MSDA_CELL"] = cell
            res[cell] = [t.float() for t in msda.ms_deform_attn_backward(*a, 64)]
            torch.cuda.synchronize()
            t = timed(lambda: msda.ms_deform_attn_backward(*a, 64))
            print(f"N={N} {mode:8s} {'small' if pyr else 'full '} cell={cell}: whole backward {t:8.1f} us", flush=True)
        for name, x, y in zip(("grad_value", "grad_loc", "grad_aw"), res["1"], res["0"]):
            d = (x - y).abs()
            print(f"   {name:10s}: max |diff| / max |ref| = {d.max().item() / y.abs().max().item():.3e}  "
                  f"non-finite {int((~torch.isfinite(x)).sum())}")
        os.environ["RLIPV2_MSDA_CELL"] = "1"
        again = msda.ms_deform_attn_backward(*a, 64)
        print("   repeatable bit for bit:", all(torch.equal(u.float(), v) for u, v in zip(again, res["1"])))


if __name__ == "__main__":
    main()
