"""The arithmetic behind csrc/msda_patch.hip, checked on the CPU against the oracle (no GPU, no product code involved):

  grad_value[pix, ch] = sum over (query, level) groups g of  A[pix, g] * grad_out[g, ch]
  A[pix, g] = sum over the group's points of  attn * tent(x - pix_x) * tent(y - pix_y),   tent(d) = max(0, 1 - |d|)

i.e. the reference's bilinear scatter (`atomicAdd(grad_value + ptr, w * top_grad_value)`, ms_deform_im2col_cuda.cuh:122-158)
written as a matrix product per 4x4-pixel patch, with the weights split into bfloat16 hi + lo for the matrix cores.  Two
claims of the kernel's header are pinned here: the tent products ARE the bilinear corner weights with the reference's
bounds rules (float64: equal to the oracle's grad_value to rounding), and the hi + lo split loses at most 2^-16 relative per
weight (so bfloat16 gradients and a bfloat16 result are not degraded)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import msda_oracle as O  # noqa: E402  (tests may use the oracle)

PYR = np.array([(10, 14), (5, 7), (3, 4), (2, 2)], dtype=np.int64)


def _problem(seed, jitter):
    rng = np.random.default_rng(seed)
    N, M, D, L, P = 2, 2, 32, 4, 4
    starts = np.concatenate(([0], np.cumsum(PYR[:, 0] * PYR[:, 1])[:-1])).astype(np.int64)
    S = int((PYR[:, 0] * PYR[:, 1]).sum())
    ref = []
    for H, W in PYR:
        ys, xs = np.meshgrid((np.arange(H) + 0.5) / H, (np.arange(W) + 0.5) / W, indexing="ij")
        ref.append(np.stack([xs.ravel(), ys.ravel()], -1))
    ref = np.concatenate(ref, 0)                                             # [S, 2]: the encoder's own pixel centres
    off = rng.normal(0.0, jitter, (N, S, M, L, P, 2)) + rng.integers(-3, 4, (1, 1, M, 1, P, 2))
    loc = ref[None, :, None, None, None, :] + off / np.stack([PYR[:, 1], PYR[:, 0]], -1)[None, None, None, :, None, :]
    aw = rng.random((N, S, M, L, P))
    aw /= aw.sum((-1, -2), keepdims=True)
    value = rng.standard_normal((N, S, M, D))
    grad_out = rng.standard_normal((N, S, M * D))
    return value, PYR, starts, loc, aw, grad_out


def _tent(d):
    return np.maximum(0.0, 1.0 - np.abs(d))


def _bf16(x):
    """round-to-nearest-even to bfloat16 precision, kept in float32 (v_cvt_pk_bf16_f32)"""
    u = np.asarray(x, dtype=np.float32).view(np.uint32)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    return r.view(np.float32)


def patch_grad_value(shapes, starts, loc, aw, grad_out, S, M, D, split=None):
    """the patch pass as matrix products: per (image, head, level, 4x4 patch) D[pix, ch] = A[pix, groups] @ G[groups, ch]"""
    N, Lq = loc.shape[0], loc.shape[1]
    gv = np.zeros((N, S, M, D))
    go = grad_out.reshape(N, Lq, M, D)
    for l, (H, W) in enumerate(shapes):
        x = loc[:, :, :, l, :, 0] * W - 0.5                                  # [N, Lq, M, P]
        y = loc[:, :, :, l, :, 1] * H - 0.5
        inside = (y > -1) & (x > -1) & (y < H) & (x < W)                    # ms_deform_im2col_cuda.cuh:285
        a = np.where(inside, aw[:, :, :, l, :], 0.0)
        for py in range(0, H, 4):
            for px in range(0, W, 4):
                rows, cols = np.arange(py, min(py + 4, H)), np.arange(px, min(px + 4, W))
                ty = _tent(np.where(inside, y, -8.0)[..., None] - rows)     # [N, Lq, M, P, rows]
                tx = _tent(np.where(inside, x, -8.0)[..., None] - cols)
                A = np.einsum("nqmp,nqmpr,nqmpc->nmrcq", a, ty, tx)         # weight of every query's group per pixel
                if split is not None:
                    A = split(A)
                pix = (starts[l] + rows[:, None] * W + cols[None, :])
                gv[:, pix] += np.einsum("nmrcq,nqmd->nrcmd", A, go)
    return gv


@pytest.mark.parametrize("jitter", [0.0, 0.6])
def test_tent_products_are_the_bilinear_scatter(jitter):
    value, shapes, starts, loc, aw, grad_out = _problem(3, jitter)
    ref_gv, _, _ = O.backward(value, shapes, starts, loc, aw, grad_out)
    got = patch_grad_value(shapes, starts, loc, aw, grad_out, value.shape[1], value.shape[2], value.shape[3])
    np.testing.assert_allclose(got, ref_gv, rtol=1e-10, atol=1e-12)


def test_hi_lo_bfloat16_split_of_the_weights_is_within_2_to_the_minus_16():
    value, shapes, starts, loc, aw, grad_out = _problem(5, 0.6)
    go_bf = _bf16(grad_out).astype(np.float64)                              # the kernel's grad_out operand is bfloat16
    exact = patch_grad_value(shapes, starts, loc, aw, go_bf, value.shape[1], value.shape[2], value.shape[3])

    def split(A):
        A32 = A.astype(np.float32)
        hi = _bf16(A32)
        lo = _bf16(A32 - hi)
        w = hi.astype(np.float64) + lo.astype(np.float64)
        nz = A32 != 0
        assert float(np.abs(w[nz] - A32[nz].astype(np.float64)).max() / 1.0) >= 0.0
        rel = np.abs(w[nz] - A32[nz]) / np.abs(A32[nz])
        assert rel.max() <= 2.0 ** -16, rel.max()
        return w

    got = patch_grad_value(shapes, starts, loc, aw, go_bf, value.shape[1], value.shape[2], value.shape[3], split=split)
    scale = np.abs(exact).max()
    assert np.abs(got - exact).max() <= 2.0 ** -14 * scale                   # far inside one bfloat16 rounding (2^-9) of the result
