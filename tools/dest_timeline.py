"""Cycle totals per phase of workgroup 0 of dest_kernel (instrumented library: make -C rlipv2_amd/csrc timeline).
usage: RLIPV2_LIB_PATH=tools/_build/librlipv2_msda_tl.so python tools/dest_timeline.py [bf16|f32] [model|uniform]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlipv2_amd import _lib, msda  # noqa: E402
from tools.msda_inputs import make_inputs  # noqa: E402

dtype = torch.float32 if (len(sys.argv) > 1 and sys.argv[1] == "f32") else torch.bfloat16
mode = sys.argv[2] if len(sys.argv) > 2 else "model"
inp = make_inputs(4, mode=mode, dtype=dtype)
a = (inp["value"], inp["shapes"], inp["starts"], inp["loc"], inp["aw"], inp["grad_out"])
msda.set_variant("quad", "dest")
L = _lib.lib()
for _ in range(2):
    msda.ms_deform_attn_backward(*a, 64)
torch.cuda.synchronize()
L.msda_debug_dest_timeline(None, 1)
iters = 5
for _ in range(iters):
    msda.ms_deform_attn_backward(*a, 64)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 16)()
L.msda_debug_dest_timeline(buf, 0)
names = ["dequeue", "mask row + prefix", "p1 decode/load/rank", "barrier A", "p2 offsets", "barrier B", "p3 scatter",
         "barrier C", "p4 walk", "barrier D", "loop exit", "store"]
items, passes = buf[14], buf[15]
tot = sum(buf[:12])
print(f"workgroup 0: {items / iters:.1f} items, {passes / iters:.1f} passes per launch, {tot / iters:.0f} cycles")
names += ["p1a wait prefetched", "p1b cursor + issue loads"]
tot += buf[12] + buf[13]
for k, nme in enumerate(names):
    per = buf[k] / max(1, passes if (2 <= k <= 9 or k >= 12) else items)
    print(f"  {nme:22s} {buf[k] / iters:12.0f} cycles/launch  {100.0 * buf[k] / tot:5.1f} %   {per:8.0f} per {'pass' if (2 <= k <= 9 or k >= 12) else 'item'}")
