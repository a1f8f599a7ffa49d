"""Run the encoder-shape backward a few times (for rocprofv3 --kernel-trace --stats)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlipv2_amd import msda
from tools.msda_inputs import make_inputs
dtype = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else torch.float32
mode = sys.argv[2] if len(sys.argv) > 2 else "model"
inp = make_inputs(4, mode=mode, dtype=dtype)
a = (inp["value"], inp["shapes"], inp["starts"], inp["loc"], inp["aw"])
for _ in range(5):
    msda.ms_deform_attn_forward(*a, 64)
    msda.ms_deform_attn_backward(*a, inp["grad_out"], 64)
torch.cuda.synchronize()
