"""Window-staged tile forward vs the direct-gather quad forward: agreement and timing (profiling aid).
RLIPV2_MSDA_TILE_H selects the tile height (8 / 12 / 16) at library load."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlipv2_amd import msda
from tools.msda_inputs import make_inputs
from tools.msda_microbench import time_call
for dtype in (torch.bfloat16, torch.float32):
    for mode in ("model", "uniform"):
        inp = make_inputs(4, mode=mode, dtype=dtype)
        a = (inp["value"], inp["shapes"], inp["starts"], inp["loc"], inp["aw"])
        res = {}
        for name in ("quad", "window"):
            msda.set_variant(name, "auto")
            out = msda.ms_deform_attn_forward(*a, 64)
            t = time_call(lambda: msda.ms_deform_attn_forward(*a, 64), 20)
            res[name] = (out.float(), t)
        msda.set_variant("auto", "auto")
        d = (res["window"][0] - res["quad"][0]).abs().max().item()
        ref = res["quad"][0].abs().max().item()
        print(f"{str(dtype):15s} {mode:8s} TILE_H={os.environ.get('RLIPV2_MSDA_TILE_H', '16'):3s} quad {res['quad'][1]*1e6:7.1f} us   "
              f"window {res['window'][1]*1e6:7.1f} us   max|diff| {d:.3e} (max|out| {ref:.3e})", flush=True)
