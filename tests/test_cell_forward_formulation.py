"""Lane-level model of cell_forward_kernel (csrc/msda_patch.hip), checked on the CPU against the oracle.  No GPU and no
product code: the kernel's DATA FLOW is restated in numpy with the hardware semantics measured in round 3
(profiles/r03_probe_mfma_tr_rates.txt) --

  ds_read_b64_tr_b16   in a 16-lane group lane p ADDRESSES row p >> 2 (8-byte piece p & 3 of it); lane i RECEIVES column i
                       of the 4 x 16 block: element e = row e
  v_mfma_f32_4x4x4_16B block = 4 consecutive lanes; A: lane r holds row r (4 k-values), B: lane j holds column j,
                       D: lane j register i = D[i][j]

-- and must reproduce the reference's forward sum (ms_deform_im2col_cuda.cuh:237-299): windows with a zero border and two
zero rows instead of validity masks, a 32-byte record per (query, sample) written by lane (query, point), weights split into
bfloat16 hi (A row 0) + lo (A row 1), channel p of the half in lane p of the query's 16-lane group.  What this pins is the
index arithmetic and operand placement of the kernel's design; the HIP code itself still needs its GPU run."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import msda_oracle as O  # noqa: E402  (tests may use the oracle)

PYR = np.array([(9, 13), (5, 7), (3, 4), (2, 2)], dtype=np.int64)        # one cell: level 0 within 16 x 16
M, D, L, P = 2, 32, 4, 4


def bf16_bits(x):
    """float32 -> bfloat16 bit pattern, round to nearest even (v_cvt_pk_bf16_f32)"""
    u = np.asarray(x, dtype=np.float32).view(np.uint32)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def bf16_val(bits):
    return (np.asarray(bits, dtype=np.uint32) << 16).view(np.float32)


def problem(seed, spread):
    rng = np.random.default_rng(seed)
    starts = np.concatenate(([0], np.cumsum(PYR[:, 0] * PYR[:, 1])[:-1])).astype(np.int64)
    S = int((PYR[:, 0] * PYR[:, 1]).sum())
    ref = []
    for H, W in PYR:
        ys, xs = np.meshgrid((np.arange(H) + 0.5) / H, (np.arange(W) + 0.5) / W, indexing="ij")
        ref.append(np.stack([xs.ravel(), ys.ravel()], -1))
    ref = np.concatenate(ref, 0)
    off = rng.normal(0.0, spread, (1, S, M, L, P, 2))
    loc = (ref[None, :, None, None, None, :] + off / np.stack([PYR[:, 1], PYR[:, 0]], -1)[None, None, None, :, None, :])
    loc[0, 0, 0, :, 0] = (-0.7, 0.5)                                        # out of range: must read the zero rows
    loc[0, 1, 1, :, 1] = (0.0, 0.0)                                         # corner pixel: three corners outside the level
    loc[0, 2, 0, :, 2] = (1.0, 1.0)
    aw = rng.random((1, S, M, L, P))
    aw /= aw.sum((-1, -2), keepdims=True)
    value = bf16_val(bf16_bits(rng.standard_normal((1, S, M, D))))          # bfloat16 values
    return value.astype(np.float32), starts, loc.astype(np.float32), aw.astype(np.float32), S


def model_forward(value, starts, loc, aw, S):
    """out [S, M, D] float32, computed the way the kernel moves data"""
    out = np.zeros((S, M, D), dtype=np.float32)
    vbits = bf16_bits(value[0])                                              # [S, M, D] uint16
    for m in range(M):
        acc = np.zeros((S, D), dtype=np.float32)                             # one register per (query, channel): lane p16 / 16 + p16
        for l, (H, W) in enumerate(PYR):
            H, W = int(H), int(W)
            x, y = loc[0, :, m, l, :, 0], loc[0, :, m, l, :, 1]              # [S, P]
            h_im = np.float32(y) * np.float32(H) - np.float32(0.5)
            w_im = np.float32(x) * np.float32(W) - np.float32(0.5)
            inside = (h_im > -1) & (w_im > -1) & (h_im < H) & (w_im < W)
            iy = np.floor(np.where(inside, h_im, 0)).astype(int)
            ix = np.floor(np.where(inside, w_im, 0)).astype(int)
            # window: every corner of every in-range sample, NOT clipped (bbox over ix .. ix + 1, iy .. iy + 1)
            if inside.any():
                x0, y0 = int(ix[inside].min()), int(iy[inside].min())
                cols, rows = int(ix[inside].max()) + 1 - x0 + 1, int(iy[inside].max()) + 1 - y0 + 1
            else:
                x0 = y0 = 0
                cols, rows = 2, 0
            pitch = cols + ((2 - cols) & 3)
            assert pitch % 4 == 2 and pitch >= cols
            win = np.full(((rows + 2) * pitch * 64) // 2, 0x7FC0, dtype=np.uint16)      # NaN: unwritten bytes must never matter
            for wy in range(rows + 2):
                for wx in range(cols):
                    gy, gx = y0 + wy, x0 + wx
                    real = wy < rows and 0 <= gy < H and 0 <= gx < W
                    dst = ((wy * pitch + wx) * 64) // 2
                    win[dst:dst + 32] = vbits[starts[l] + gy * W + gx, m] if real else 0
            zero_base = rows * pitch * 64
            # phase A: lane (query, point) -> record
            lh = np.where(inside, h_im, 0) - iy
            lw = np.where(inside, w_im, 0) - ix
            hh, hw = 1 - lh, 1 - lw
            a = np.where(inside, aw[0, :, m, l, :], 0).astype(np.float32)
            w4 = np.stack([hh * hw * a, hh * lw * a, lh * hw * a, lh * lw * a], -1).astype(np.float32)   # TL TR BL BR
            hi = bf16_bits(w4)
            lo = bf16_bits(w4 - bf16_val(hi))
            base = np.where(inside, ((iy - y0) * pitch + (ix - x0)) * 64, zero_base)
            # phase B: 4 queries per wave step, one 16-lane group each
            for q0 in range(0, S, 4):
                for s in range(P):
                    for g4 in range(min(4, S - q0)):
                        q = q0 + g4
                        for half in range(2):
                            # transposing read of the group: lane p addresses corner p >> 2, piece p & 3
                            rows_read = np.zeros((4, 16), dtype=np.uint16)
                            for p in range(16):
                                crn, r4 = p >> 2, p & 3
                                addr = base[q, s] + ((crn >> 1) * pitch + (crn & 1)) * 64 + r4 * 8 + half * 32
                                rows_read[crn, r4 * 4:(r4 + 1) * 4] = win[addr // 2:addr // 2 + 4]
                            bcol = bf16_val(rows_read)                       # lane i receives column i: bcol[:, i]
                            for blk in range(4):                             # the group's 4 MFMA blocks
                                A = np.zeros((4, 4), dtype=np.float32)       # lane r of the block holds row r
                                for r in range(4):
                                    A[r] = bf16_val(lo[q, s] if (r & 1) else hi[q, s])    # rec + (r & 1) * 8
                                B = bcol[:, blk * 4:blk * 4 + 4]             # lane j holds column j: channel 4 blk + j
                                Dm = A.astype(np.float64) @ B.astype(np.float64)
                                for j in range(4):                           # lane j: registers D[0..3][j]; rows 0 + 1 are kept
                                    acc[q, half * 16 + blk * 4 + j] += np.float32(Dm[0, j] + Dm[1, j])
        out[:, m] = acc
    return out


@pytest.mark.parametrize("spread", [0.7, 2.5])
def test_lane_level_model_reproduces_the_reference_forward(spread):
    value, starts, loc, aw, S = problem(7, spread)
    ref = O.forward(value.astype(np.float64), PYR, starts, loc.astype(np.float64), aw.astype(np.float64))[0].reshape(S, M, D)
    got = model_forward(value, starts, loc, aw, S)
    assert np.isfinite(got).all()                                           # the NaN filler of the window was never read
    # float32 geometry + weights to 2^-16: far inside one bfloat16 rounding of the result
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-4 * float(np.abs(ref).max()))
