"""Ablation timing of the window backward kernel (profiling aid, not a benchmark)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlipv2_amd import msda
from tools.msda_inputs import make_inputs
from tools.msda_microbench import time_call

inp = make_inputs(4, mode="model", dtype=torch.bfloat16)
a = (inp["value"], inp["shapes"], inp["starts"], inp["loc"], inp["aw"], inp["grad_out"])
msda.set_variant("quad", "window")
for grid in ("512",):
    os.environ["RLIPV2_MSDA_GRID"] = grid
    for dbg, what in ((0, "full (chunk 64)"), (32, "chunk 32"), (2, "no global atomics"), (4, "no walk"), (16, "unsorted records")):
        os.environ["RLIPV2_MSDA_DEBUG"] = str(dbg)
        t = time_call(lambda: msda.ms_deform_attn_backward(*a, 64), 10)
        print(f"grid {grid:5s} dbg {dbg:2d} {what:32s} {t*1e6:10.1f} us", flush=True)
