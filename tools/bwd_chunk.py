import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlipv2_amd import msda
from tools.msda_inputs import make_inputs
from tools.msda_microbench import time_call
inp = make_inputs(4, mode="model", dtype=torch.bfloat16)
a = (inp["value"], inp["shapes"], inp["starts"], inp["loc"], inp["aw"], inp["grad_out"])
t = time_call(lambda: msda.ms_deform_attn_backward(*a, 64), 20)
print(f"DEBUG={os.environ.get('RLIPV2_MSDA_DEBUG', '0'):6s} GRID={os.environ.get('RLIPV2_MSDA_GRID', '-'):5s} encoder backward {t*1e6:8.1f} us", flush=True)
