"""Ablation arms of the window-staged tile forward (csrc/msda_quad.hip: tile_forward_kernel; ablation build): base loop, bounding
boxes, staging, sampling -- the phase times quoted in DESIGN.md section 4 (a measured negative result for bf16)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlipv2_amd import msda
from tools.msda_inputs import make_inputs
from tools.msda_microbench import time_call
inp = make_inputs(4, mode="model", dtype=torch.bfloat16)
a = (inp["value"], inp["shapes"], inp["starts"], inp["loc"], inp["aw"])
msda.set_variant("window", "auto")
for dbg, what in ((0, "full"), (16, "no staging DMA"), (32, "no sampling"), (48, "no staging, no sampling"), (112, "+ no bbox atomics")):
    os.environ["RLIPV2_MSDA_DEBUG"] = str(dbg)
    t = time_call(lambda: msda.ms_deform_attn_forward(*a, 64), 20)
    print(f"TILE_H={os.environ.get('RLIPV2_MSDA_TILE_H','16'):3s} GRID={os.environ.get('RLIPV2_MSDA_GRID','-'):5s} dbg {dbg:3d} {what:28s} {t*1e6:8.1f} us", flush=True)
