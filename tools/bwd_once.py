"""Runs the MSDA backward (one variant, encoder shape) a few times: target of rocprofv3 kernel traces.
usage: python tools/bwd_once.py [variant] [dtype] [mode] [iters] [N]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlipv2_amd import msda  # noqa: E402
from tools.msda_inputs import make_inputs  # noqa: E402

variant = sys.argv[1] if len(sys.argv) > 1 else "dest"
dtype = torch.bfloat16 if (len(sys.argv) < 3 or sys.argv[2] == "bf16") else torch.float32
mode = sys.argv[3] if len(sys.argv) > 3 else "model"
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 10
N = int(sys.argv[5]) if len(sys.argv) > 5 else 4
inp = make_inputs(N, mode=mode, dtype=dtype)
a = (inp["value"], inp["shapes"], inp["starts"], inp["loc"], inp["aw"])
if variant == "fwd":
    for _ in range(iters):
        msda.ms_deform_attn_forward(*a, 64)
else:
    msda.set_variant("quad", variant)
    for _ in range(iters):
        msda.ms_deform_attn_backward(*a, inp["grad_out"], 64)
torch.cuda.synchronize()
