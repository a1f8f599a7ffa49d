"""How many (query, level) groups touch how many 4x4-pixel patches (the work list of csrc/msda_patch.hip), from the inputs
alone (torch, any device): python tools/patch_stats.py [mode] [N]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.msda_inputs import make_inputs, PYRAMID_800x1333  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "model"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = "cuda:0" if torch.cuda.is_available() else "cpu"
inp = make_inputs(N, mode=mode, dtype=torch.float32, device=dev, seed=3)
loc = inp["loc"]                                     # [N, Lq, M, L, P, 2]
tot_groups = 0
tot_inc = 0
for l, (H, W) in enumerate(PYRAMID_800x1333):
    x = loc[:, :, :, l, :, 0] * W - 0.5
    y = loc[:, :, :, l, :, 1] * H - 0.5
    inside = (y > -1) & (x > -1) & (y < H) & (x < W)
    x0, y0 = torch.floor(x), torch.floor(y)
    ids = []
    for dy in (0, 1):
        for dx in (0, 1):
            cy, cx = y0 + dy, x0 + dx
            ok = inside & (cy >= 0) & (cy < H) & (cx >= 0) & (cx < W)
            pid = (torch.div(cy, 4, rounding_mode="floor") * 1000 + torch.div(cx, 4, rounding_mode="floor")).long()
            ids.append(torch.where(ok, pid, torch.full_like(pid, -1)))
    ids = torch.cat(ids, -1)                          # [N, Lq, M, 16]
    s, _ = ids.sort(-1)
    distinct = ((s[..., 1:] != s[..., :-1]) & (s[..., 1:] >= 0)).sum(-1) + (s[..., 0] >= 0).long()
    groups = ids.shape[0] * ids.shape[1] * ids.shape[2]
    inc = int(distinct.sum())
    npatch = ((H + 3) // 4) * ((W + 3) // 4)
    print(f"level {l}: {groups} groups, {inc} (group, patch) incidences = {inc / groups:.2f} per group, "
          f"{inc / (npatch * N * 8):.0f} per patch -> {inc / (npatch * N * 8) / 32:.1f} MFMA steps per patch")
    tot_groups += groups
    tot_inc += inc
print(f"total: {tot_inc} incidences ({tot_inc / tot_groups:.2f} per group), {tot_inc / 32:.0f} steps of 32")
