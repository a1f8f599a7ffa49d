"""Geometry-sharing forward (RLIPV2_MSDA_FWD_SHARED=1, default) vs the original quad forward (=0): the env
var is read once per process, so this script is run once per setting and prints timing + an output checksum."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlipv2_amd import msda
from tools.msda_inputs import make_inputs
from tools.msda_microbench import time_call
msda.set_variant("quad", "auto")
for dtype in (torch.bfloat16, torch.float32):
    for mode in ("model", "uniform", "decoder"):
        inp = make_inputs(4, mode=mode, dtype=dtype)
        a = (inp["value"], inp["shapes"], inp["starts"], inp["loc"], inp["aw"])
        out = msda.ms_deform_attn_forward(*a, 64)
        t = time_call(lambda: msda.ms_deform_attn_forward(*a, 64), 20)
        print(f"shared={os.environ.get('RLIPV2_MSDA_FWD_SHARED', '1')} {str(dtype):15s} {mode:8s} {t*1e6:8.1f} us  "
              f"checksum {out.double().sum().item():.10e} {out.double().abs().sum().item():.10e}", flush=True)
