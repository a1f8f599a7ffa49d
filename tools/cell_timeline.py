"""Cycle stamps of cell_backward_kernel's phases, summed over the workgroups (thread 0 of each; ablation build):
binning | mask write-out | window staging | per task: operand wait, samples | phase 3 as a whole."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RLIPV2_CELL_DBG"] = sys.argv[2] if len(sys.argv) > 2 else "16"
from rlipv2_amd import _lib, msda  # noqa: E402
from tools.msda_inputs import make_inputs  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "init"
inp = make_inputs(4, mode=mode, dtype=torch.bfloat16, seed=3)
a = (inp["value"], inp["shapes"], inp["starts"], inp["loc"], inp["aw"], inp["grad_out"])
L = _lib.lib()
msda.ms_deform_attn_backward(*a, 64)
torch.cuda.synchronize()
L.msda_debug_cell_timeline(None, 1)
msda.ms_deform_attn_backward(*a, 64)
torch.cuda.synchronize()
buf = (ctypes.c_uint64 * 16)()
L.msda_debug_cell_timeline(buf, 0)
wgs = 4 * 8 * 77
names = ["binning (rest)", "mask write-out", "staging (+plan)", "task operand wait (sum over 3 tasks)", "task samples (sum over 3 tasks)", "phase 3 tail",
         "-", "-", "bin: patch ranges", "bin: table zero", "bin: loads arrive", "bin: arithmetic + ORs (thread 0)", "bin: closing barrier"]
for n, v in zip(names, buf):
    print(f"{n:40s} {v / wgs:10.0f} cycles per workgroup")
