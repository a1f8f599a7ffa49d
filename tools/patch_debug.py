"""Small-pyramid debug of the patch pass against the sorting pass (ablation build)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlipv2_amd import msda  # noqa: E402
from tools.msda_inputs import make_inputs  # noqa: E402

pyr = [(25, 34), (13, 17), (7, 9), (4, 5)]
if len(sys.argv) > 1 and sys.argv[1] == "big":
    pyr = [(100, 167), (50, 84), (25, 42), (13, 21)]
inp = make_inputs(1, pyramid=pyr, mode="model", dtype=torch.bfloat16, seed=3)
a = (inp["value"], inp["shapes"], inp["starts"], inp["loc"], inp["aw"], inp["grad_out"])
res = {}
for patch in ("0", "1"):
    os.environ["RLIPV2_MSDA_PATCH"] = patch
    res[patch] = msda.ms_deform_attn_backward(*a, 64)[0].float()
    torch.cuda.synchronize()
ref, got = res["0"][0], res["1"][0]          # [S, M, D]
start = 0
for l, (H, W) in enumerate(pyr):
    r, g = ref[start:start + H * W], got[start:start + H * W]
    bad = (~torch.isfinite(g)).any(-1)
    d = (r - g).abs().amax(-1)
    print(f"level {l}: rows {H * W * r.shape[1]}, non-finite rows {int(bad.sum())}, rows with |diff| > 1e-2 max: "
          f"{int((d > 1e-2 * r.abs().max()).sum())}, max diff {float(d[~bad].max()) if (~bad).any() else -1:.3e} (ref max {float(r.abs().max()):.3e})")
    if bad.any():
        idx = bad.nonzero()[:8]
        print("   first bad (pixel, head):", [(int(i[0]) // W, int(i[0]) % W, int(i[1])) for i in idx])
    start += H * W
