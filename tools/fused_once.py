"""Runs the fused geometry + sampling kernels (msda_fused_forward / msda_fused_backward_ws) at the encoder shape a few
times: target of rocprofv3 kernel traces / counter passes.  usage: python tools/fused_once.py fwd|bwd [iters] [N]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlipv2_amd import msda  # noqa: E402
from tools.msda_inputs import PYRAMID_800x1333, make_inputs  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "bwd"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
N = int(sys.argv[3]) if len(sys.argv) > 3 else 4
inp = make_inputs(N, mode="model", dtype=torch.bfloat16)
value, shapes, starts = inp["value"], inp["shapes"], inp["starts"]
msda.attach_host_shapes(shapes, PYRAMID_800x1333)
S, M, L, P = value.shape[1], 8, 4, 4
g = torch.Generator(device="cuda").manual_seed(1)
# raw projection rows whose geometry reproduces model-like locations: reference = pixel centre, offsets of a few pixels
ref = []
for (H, W) in PYRAMID_800x1333:
    ys, xs = torch.meshgrid((torch.arange(H, device="cuda") + 0.5) / H, (torch.arange(W, device="cuda") + 0.5) / W, indexing="ij")
    ref.append(torch.stack([xs.reshape(-1), ys.reshape(-1)], -1))
ref = torch.cat(ref, 0)[None, :, None, :].expand(N, S, L, 2).contiguous()
qproj = torch.randn(N, S, M * L * P * 3, device="cuda", generator=g)
qproj[..., :M * L * P * 2] *= 2.5
qproj = qproj.bfloat16()
if what == "fwd":
    for _ in range(iters):
        msda.ms_deform_attn_fused_forward(value, shapes, starts, qproj, ref, True)
else:
    out, loc, aw = msda.ms_deform_attn_fused_forward(value, shapes, starts, qproj, ref, True)
    hs = msda.host_shapes(shapes)
    for _ in range(iters):
        msda.ms_deform_attn_fused_backward(value, shapes, starts, loc, aw, ref, inp["grad_out"], hs)
torch.cuda.synchronize()
