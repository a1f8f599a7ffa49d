"""Workload census of the encoder backward's patch pass (csrc/msda_patch.hip) on the bench's locations, CPU only (numpy):
how many (query, level) groups touch a 4 x 4 patch, how full the 32-group MFMA steps are, how the steps split over the
levels, and the same for the cell pass's tasks.  No kernels, no oracle: plain index arithmetic on the synthetic locations of
tools/msda_inputs.py (modes "init" and "model").   usage: python tools/patch_census.py [init|model]
"""
import math
import sys

import numpy as np

PYR = [(100, 167), (50, 84), (25, 42), (13, 21)]
M, P = 8, 4


def locations(mode, rng):
    th = np.arange(M) * (2 * math.pi / M)
    dirs = np.stack([np.cos(th), np.sin(th)], -1)
    dirs = dirs / np.abs(dirs).max(-1, keepdims=True)
    ref = []
    for (H, W) in PYR:
        ys, xs = np.meshgrid((np.arange(H) + 0.5) / H, (np.arange(W) + 0.5) / W, indexing="ij")
        ref.append(np.stack([xs.reshape(-1), ys.reshape(-1)], -1))
    ref = np.concatenate(ref, 0)                                         # [S, 2]
    S = ref.shape[0]
    off = dirs[None, :, None, None, :] * np.arange(1, P + 1)[None, None, None, :, None]      # [1, M, 1, P, 2]
    off = np.broadcast_to(off, (S, M, len(PYR), P, 2)).copy()
    if mode == "model":
        off += rng.standard_normal(off.shape)
    norm = np.asarray([(W, H) for (H, W) in PYR], dtype=np.float64)
    return ref[:, None, None, None, :] + off / norm[None, None, :, None, :]                 # [S, M, L, P, 2]


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "init"
    loc = locations(mode, np.random.default_rng(0))
    S = loc.shape[0]
    tot_groups = tot_touch = tot_steps = 0
    print(f"mode {mode}: S = {S}, per image; groups = (query, head, level)")
    print("level  patches  groups/patch(mean, p50, max)  touches/group  steps/patch  fill   share of steps")
    rows, enum = [], []
    # cell and bit of every query (csrc/msda_patch.hip: cells = pyramid columns over 16 x 16 level-0 pixels, 340 bits)
    CY, CX = (PYR[0][0] + 15) // 16, (PYR[0][1] + 15) // 16
    qcell, qword = [], []
    for lq, (H, W) in enumerate(PYR):
        sh = 4 - lq
        iy, ix = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
        qcell.append(((iy >> sh) * CX + (ix >> sh)).reshape(-1))
        j = (0, 256, 320, 336)[lq] + ((iy & ((1 << sh) - 1)) << sh) + (ix & ((1 << sh) - 1))
        qword.append((j >> 5).reshape(-1))
    qcell, qword = np.concatenate(qcell), np.concatenate(qword)
    for l, (H, W) in enumerate(PYR):
        PY, PX = (H + 3) // 4, (W + 3) // 4
        x = loc[:, :, l, :, 0] * W - 0.5
        y = loc[:, :, l, :, 1] * H - 0.5                                 # [S, M, P]
        inside = (y > -1) & (x > -1) & (y < H) & (x < W)
        y0 = np.clip(np.floor(y).astype(int), 0, None); y1 = np.clip(np.floor(y).astype(int) + 1, None, H - 1)
        x0 = np.clip(np.floor(x).astype(int), 0, None); x1 = np.clip(np.floor(x).astype(int) + 1, None, W - 1)
        ids = []
        for yy in (y0, y1):
            for xx in (x0, x1):
                pid = (yy >> 2) * PX + (xx >> 2)
                ids.append(np.where(inside, pid, -1))
        ids = np.stack(ids, -1).reshape(S, M, 4 * P)                     # 16 patch ids per group
        ids.sort(-1)
        first = np.concatenate([np.ones((S, M, 1), bool), ids[:, :, 1:] != ids[:, :, :-1]], -1) & (ids >= 0)
        head = np.broadcast_to(np.arange(M)[None, :, None], ids.shape)
        cnt = np.bincount((head * (PY * PX) + ids)[first], minlength=M * PY * PX).reshape(M, PY * PX)
        touches = first.sum(-1)                                          # patches per group
        # the enumeration's work: (patch, source cell) slots with at least one group, and their non-zero 32-query mask words
        key = (head.astype(np.int64) * (PY * PX) + ids)[first]
        gq = np.broadcast_to(np.arange(S)[:, None, None], ids.shape)[first]
        cells_nonempty = np.unique(key * (CY * CX) + qcell[gq]).size
        words_nonzero = np.unique((key * (CY * CX) + qcell[gq]) * 11 + qword[gq]).size
        rad = (1, 2, 3, 6)[l]
        nby, nbx = min(2 * rad + 1, CY), min(2 * rad + 1, CX)
        enum.append((M * PY * PX * nby * nbx, cells_nonempty, words_nonzero))
        steps = (cnt + 31) // 32
        rows.append((l, M * PY * PX, cnt, touches, steps))
        tot_groups += S * M; tot_touch += int(cnt.sum()); tot_steps += int(steps.sum())
    for l, npatch, cnt, touches, steps in rows:
        print(f"{l:5d}  {npatch:7d}  {cnt.mean():8.1f} {np.median(cnt):6.0f} {cnt.max():6d}          {touches.mean():6.2f}      "
              f"{steps.mean():8.2f}   {cnt.sum() / (32.0 * steps.sum()):.3f}   {steps.sum() / tot_steps:.3f}")
    print(f"all levels: {tot_groups} groups, {tot_touch} (group, patch) pairs = {tot_touch / tot_groups:.2f} per group, "
          f"{tot_steps} steps of 32, fill {tot_touch / (32.0 * tot_steps):.3f}")
    print("enumeration per image: level  slots scanned  non-empty slots  non-zero mask words   (per 32-group step)")
    for (l, npatch, cnt, touches, steps), (sl, ne, wz) in zip(rows, enum):
        st = float(steps.sum())
        print(f"                       {l:5d}  {sl:13d}  {ne:15d}  {wz:19d}   ({sl / st:.2f} / {ne / st:.2f} / {wz / st:.2f})")
    print(f"per batch-4 call: {4 * tot_steps} steps = {4 * tot_steps * 8} v_mfma_f32_16x16x32_bf16 (hi + lo weights x 4 channel blocks)")


if __name__ == "__main__":
    main()
