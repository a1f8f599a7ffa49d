"""Ablation timing of the window backward kernel (profiling aid, not a benchmark)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlipv2_amd import msda
from tools.msda_inputs import make_inputs
from tools.msda_microbench import time_call

inp = make_inputs(4, mode="model", dtype=torch.bfloat16)
a = (inp["value"], inp["shapes"], inp["starts"], inp["loc"], inp["aw"], inp["grad_out"])
msda.set_variant("quad", "window")
for grid in ("512", "256", "1024"):
    os.environ["RLIPV2_MSDA_GRID"] = grid
    for dbg, what in ((0, "full"), (1, "no LDS adds"), (2, "no flush"), (3, "no adds, no flush"), (4, "value gathers -> one address"),
                      (8, "no band loop at all"), (12, "no bands, trivial gathers")):
        os.environ["RLIPV2_MSDA_DEBUG"] = str(dbg)
        t = time_call(lambda: msda.ms_deform_attn_backward(*a, 64), 10)
        print(f"grid {grid:5s} dbg {dbg:2d} {what:32s} {t*1e6:10.1f} us", flush=True)
