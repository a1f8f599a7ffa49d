"""Ablation arms of patch_dest_kernel (ablation build): RLIPV2_PATCH_DBG bits 1 = enumeration only, 2 = cache-resident
operands, 4 = loads only.  Prints HIP-event time of the whole backward (K1 ~ 200 us and bin2 are included)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlipv2_amd import msda  # noqa: E402
from tools.msda_inputs import make_inputs  # noqa: E402
from tools.patch_check import timed  # noqa: E402

inp = make_inputs(4, mode="model", dtype=torch.bfloat16, seed=3)
a = (inp["value"], inp["shapes"], inp["starts"], inp["loc"], inp["aw"], inp["grad_out"])
for dbg in sys.argv[1:] or ["0", "1", "2", "4", "6"]:
    os.environ["RLIPV2_PATCH_DBG"] = dbg
    print(f"dbg={dbg}: whole backward {timed(lambda: msda.ms_deform_attn_backward(*a, 64)):8.1f} us", flush=True)
