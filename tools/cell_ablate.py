"""Ablation arms of cell_backward_kernel (ablation build): RLIPV2_CELL_DBG 1 = binning only, 2 = binning + staging."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlipv2_amd import msda  # noqa: E402
from tools.msda_inputs import make_inputs  # noqa: E402
from tools.patch_check import timed  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "init"
inp = make_inputs(4, mode=mode, dtype=torch.bfloat16, seed=3)
a = (inp["value"], inp["shapes"], inp["starts"], inp["loc"], inp["aw"], inp["grad_out"])
os.environ["RLIPV2_PATCH_DBG"] = "1"          # the patch pass reduced to its enumeration: a constant background
for dbg in ["0", "1", "2", "4"]:
    os.environ["RLIPV2_CELL_DBG"] = dbg
    print(f"{mode} cell dbg={dbg}: whole backward {timed(lambda: msda.ms_deform_attn_backward(*a, 64)):8.1f} us", flush=True)
