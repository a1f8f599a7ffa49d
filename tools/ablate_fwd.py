"""Ablation timing of the quad forward kernel (profiling aid)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlipv2_amd import msda
from tools.msda_inputs import make_inputs
from tools.msda_microbench import time_call
for dtype in (torch.bfloat16, torch.float32):
    inp = make_inputs(4, mode="model", dtype=dtype)
    a = (inp["value"], inp["shapes"], inp["starts"], inp["loc"], inp["aw"])
    for dbg, what in ((0, "full"), (1, "gathers from cache-resident rows")):
        os.environ["RLIPV2_MSDA_DEBUG"] = str(dbg)
        t = time_call(lambda: msda.ms_deform_attn_forward(*a, 64), 20)
        print(f"{str(dtype):16s} dbg {dbg} {what:36s} {t*1e6:9.1f} us", flush=True)
os.environ["RLIPV2_MSDA_DEBUG"] = "0"
