"""The "records" route of the encoder backward (csrc/msda_cell_forward.inc with EMIT, csrc/msda_cell_records.inc; round 5, never
run on hardware) on the lane-level model of tools/emu/, through the C ABI (msda_records_bytes / msda_records_forward /
msda_records_backward): the forward pass leaves a 2-byte record per sample (round 6's record diet; the bilinear fractions and the
weight come from the group records), the cells' window tables and the patch pass's masks and group records; the backward pass
then runs no window placement, no corner clipping and no binning.

The bar is the verdict's: BIT-EQUAL to the product kernels (msda_backward_ws / msda_fused_backward_ws: cell_backward_kernel +
patch_dest_kernel, both validated on hardware in round 3) on the same call, both operand orders of the 4x4x4 products; the
oracle is only the cross-check that the product route itself is right on these inputs."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import msda_oracle as O  # noqa: E402  (tests may use the oracle)
from test_cell_forward_emulated import CLANG, bf16_bits, bf16_val, make_problem  # noqa: E402

pytestmark = pytest.mark.skipif(not os.path.exists(CLANG), reason="needs the ROCm clang++ (ext_vector_type) as host compiler")

BF16, FLAG_BF16_GV, FLAG_SWAP, CELL = 2, 0x200, 0x400, 6
FULL = os.environ.get("RLIPV2_TEST_EMU_FULL", "0") == "1"     # the default suite runs one problem per code path (~2.5 min)


@pytest.fixture(scope="module")
def lib(emu_library):
    return _bind(ctypes.CDLL(emu_library()))


@pytest.fixture(scope="module")
def ablation_lib(emu_library):
    return _bind(ctypes.CDLL(emu_library("-DMSDA_ABLATION")))


def _bind(L):
    vp, i, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
    d = [i] * 7
    L.msda_records_bytes.argtypes = [i, vp, *d]
    L.msda_records_bytes.restype = sz
    L.msda_records_forward.argtypes = [i, vp, vp, vp, vp, vp, vp, i, vp, vp, *d, vp, vp, sz, vp]
    L.msda_records_backward.argtypes = [i, i, vp, vp, vp, vp, vp, vp, vp, i, vp, *d, vp, vp, vp, vp, vp, sz, vp, sz, vp]
    L.msda_forward_hs.argtypes = [i, i, vp, vp, vp, vp, vp, vp, *d, vp, vp]
    L.msda_backward_ws.argtypes = [i, i, vp, vp, vp, vp, vp, vp, vp, *d, vp, vp, vp, vp, sz, vp]
    L.msda_backward_workspace_bytes.argtypes = [i, vp, *d]
    L.msda_backward_workspace_bytes.restype = sz
    L.msda_fused_forward_hs.argtypes = [i, i, vp, vp, vp, vp, vp, vp, i, *d, vp, vp, vp, vp]
    L.msda_fused_backward_ws.argtypes = [i, i, vp, vp, vp, vp, vp, vp, vp, i, vp, *d, vp, vp, vp, sz, vp]
    return L



def p(a):
    return a.ctypes.data if a is not None else None


CASES = [
    ("windows of every level in LDS, 2 x 2 cells, 2 heads", [(20, 27), (10, 14), (5, 7), (3, 4)], 2, (1.5, 1.5, 1.0, 0.7)),
    ("level 0 too wide for the window budget: the buffer-load route", [(30, 40), (15, 20), (8, 10), (4, 5)], 1, (25.0, 2.0, 1.0, 0.7)),
] + ([("ragged pyramid, wide offsets on the coarse levels", [(25, 34), (13, 17), (7, 9), (4, 5)], 1, (2.0, 3.0, 3.0, 3.0))] if FULL else [])


@pytest.mark.parametrize("name,pyr,M,spread", CASES, ids=[c[0] for c in CASES])
def test_op_signature_bit_equal_to_the_product_route(lib, name, pyr, M, spread):
    pyr, starts, S, value, loc, aw = make_problem(pyr, M, spread, seed=7)
    rng = np.random.default_rng(3)
    grad_out = bf16_val(bf16_bits(rng.standard_normal((1, S, M * 32)))).astype(np.float32)
    vb, gob = np.ascontiguousarray(bf16_bits(value)), np.ascontiguousarray(bf16_bits(grad_out))
    sh, st = np.ascontiguousarray(pyr, dtype=np.int64), np.ascontiguousarray(starts, dtype=np.int64)
    dims = (1, S, M, 32, 4, S, 4)
    # the product route: forward (cell kernel, the same sums as the records forward) and backward
    out_ref = np.zeros((1, S, M * 32), dtype=np.uint16)
    assert lib.msda_forward_hs(CELL, BF16, p(vb), p(sh), p(st), p(sh), p(loc), p(aw), *dims, p(out_ref), None) == 0
    ws_bytes = lib.msda_backward_workspace_bytes(BF16, p(sh), *dims)
    assert ws_bytes > 0
    gv_ref, gl_ref, ga_ref = np.zeros(vb.shape, dtype=np.uint16), np.full(loc.shape, np.nan, np.float32), np.full(aw.shape, np.nan, np.float32)
    ws = np.zeros(ws_bytes + 64, dtype=np.uint8)
    assert lib.msda_backward_ws(4 | FLAG_BF16_GV, BF16, p(vb), p(sh), p(st), p(sh), p(loc), p(aw), p(gob), *dims, p(gv_ref),
                                p(gl_ref), p(ga_ref), p(ws), ws_bytes, None) == 0
    # ... is the oracle's (so "bit-equal to it" means something)
    a = (value.astype(np.float64), pyr, starts, loc.astype(np.float64), aw.astype(np.float64))
    o_gv, o_gl, o_ga = O.backward(*a, grad_out.astype(np.float64))
    assert np.abs(bf16_val(gv_ref) - o_gv).max() <= 2.0 ** -7 * np.abs(o_gv).max()
    np.testing.assert_allclose(ga_ref, o_ga, rtol=1e-4, atol=1e-5 * float(np.abs(o_ga).max()))

    rec_bytes = lib.msda_records_bytes(BF16, p(sh), *dims)
    assert rec_bytes > 0
    records = np.full(rec_bytes + 64, 0xA5, dtype=np.uint8)                     # garbage: whatever is read must have been written
    out = np.zeros_like(out_ref)
    assert lib.msda_records_forward(BF16, p(vb), p(sh), p(st), p(sh), None, None, 0, p(loc), p(aw), *dims, p(out), p(records),
                                    rec_bytes, None) == 0
    assert np.array_equal(out, out_ref)
    assert np.all(records[rec_bytes:] == 0xA5)
    far = int(records[:256].view(np.int32)[60])
    assert far == 0                                                             # (the patch pass, not the sorting pass, is what runs below)
    cells = -(-int(pyr[0][0]) // 16) * -(-int(pyr[0][1]) // 16)
    wtab = records[256:256 + M * cells * 4 * 32].view(np.int32).reshape(M * cells, 4, 8)      # x0, y0, cols, rows, pitch, staged
    assert np.all(wtab[..., 4] % 4 == 2) and np.all(wtab[..., 4] >= wtab[..., 2])           # pitch = 2 (mod 4), >= cols
    if "every level in LDS" in name:                                            # which route the levels took
        assert np.all(wtab[..., 5] == 1), wtab[..., 5]
    elif "buffer-load" in name:
        assert np.sum(wtab[..., 5] == 0) >= cells, wtab[..., 5]
    # (both operand orders of the 4x4x4 products with RLIPV2_TEST_EMU_FULL=1; by default one per problem, alternating)
    orders = (FLAG_BF16_GV, FLAG_BF16_GV | FLAG_SWAP) if FULL else ((FLAG_BF16_GV | FLAG_SWAP,) if "every level" in name else (FLAG_BF16_GV,))
    for flags in orders:
        gv, gl, ga = np.zeros_like(gv_ref), np.full_like(gl_ref, np.nan), np.full_like(ga_ref, np.nan)
        ws = np.zeros(ws_bytes + 64, dtype=np.uint8)
        assert lib.msda_records_backward(flags, BF16, p(vb), p(sh), p(st), p(sh), p(loc), p(aw), None, 0, p(gob), *dims, p(gv),
                                         p(gl), p(ga), None, p(records), rec_bytes, p(ws), ws_bytes, None) == 0
        assert np.array_equal(gl.view(np.uint32), gl_ref.view(np.uint32)), f"grad_sampling_loc differs (flags {flags:#x}, far {far})"
        assert np.array_equal(ga.view(np.uint32), ga_ref.view(np.uint32)), f"grad_attn_weight differs (flags {flags:#x})"
        assert np.array_equal(gv, gv_ref), f"grad_value differs (flags {flags:#x}, far {far})"
    if "every level" in name:
        # run-to-run: the same bits again (no atomics, one writer per element, fixed summation order) -- under a different
        # interleaving of the 512 lane threads
        gv2, gl2, ga2 = np.zeros_like(gv_ref), np.full_like(gl_ref, np.nan), np.full_like(ga_ref, np.nan)
        ws = np.zeros(ws_bytes + 64, dtype=np.uint8)
        assert lib.msda_records_backward(flags, BF16, p(vb), p(sh), p(st), p(sh), p(loc), p(aw), None, 0, p(gob), *dims, p(gv2),
                                         p(gl2), p(ga2), None, p(records), rec_bytes, p(ws), ws_bytes, None) == 0
        assert np.array_equal(gv2, gv) and np.array_equal(gl2.view(np.uint32), gl.view(np.uint32)) and np.array_equal(ga2.view(np.uint32), ga.view(np.uint32))


def test_records_route_with_the_cell_major_grad_out_copy_fused_in(ablation_lib, monkeypatch):
    """Round 6, ablation build + RLIPV2_PATCH_CELLG=1: cell_records_backward_kernel -- which holds every query's grad_out row anyway
    -- also leaves the cell-major copy that patch_dest_multi_kernel<., ., CELLG> reads (no grad_out_cells_kernel launch).  Bar: the
    same bits as before: the module-operand comparison of the test below on the ablation library with the arm switched on (the product
    route inside it then runs the arm with its stand-alone copy kernel, itself bit-equal to the product kernels:
    tests/test_backward_emulated.py), and the op's signature with the switch on against off."""
    monkeypatch.setenv("RLIPV2_PATCH_CELLG", "1")
    test_module_operands_bit_equal_to_the_fused_product_route(ablation_lib, 2)      # (the REFDIM 2 instantiations: the train step's call)
    monkeypatch.delenv("RLIPV2_PATCH_CELLG")
    # the op's signature, the two settings of the switch against each other (switch off = the records route of the tests above, bit-equal
    # to the product kernels there): the same bits, from a workspace full of garbage -- the copy was written before it was read
    name, pyr, M, spread = CASES[0]
    pyr, starts, S, value, loc, aw = make_problem(pyr, M, spread, seed=7)
    gob = np.ascontiguousarray(bf16_bits(np.random.default_rng(3).standard_normal((1, S, M * 32))))
    vb = np.ascontiguousarray(bf16_bits(value))
    sh, st = np.ascontiguousarray(pyr, dtype=np.int64), np.ascontiguousarray(starts, dtype=np.int64)
    dims = (1, S, M, 32, 4, S, 4)
    L = ablation_lib
    rec_bytes, ws_bytes = L.msda_records_bytes(BF16, p(sh), *dims), L.msda_backward_workspace_bytes(BF16, p(sh), *dims)
    got = []
    for cellg in ("0", "1"):
        monkeypatch.setenv("RLIPV2_PATCH_CELLG", cellg)
        records, out = np.zeros(rec_bytes, dtype=np.uint8), np.zeros((1, S, M * 32), dtype=np.uint16)
        assert L.msda_records_forward(BF16, p(vb), p(sh), p(st), p(sh), None, None, 0, p(loc), p(aw), *dims, p(out), p(records), rec_bytes, None) == 0
        gv, gl, ga = np.zeros_like(vb), np.full(loc.shape, np.nan, np.float32), np.full(aw.shape, np.nan, np.float32)
        ws = np.full(ws_bytes + 64, 0xA5, dtype=np.uint8)                       # (garbage: the copy must have been written before it is read)
        assert L.msda_records_backward(FLAG_BF16_GV | FLAG_SWAP, BF16, p(vb), p(sh), p(st), p(sh), p(loc), p(aw), None, 0, p(gob), *dims,
                                       p(gv), p(gl), p(ga), None, p(records), rec_bytes, p(ws), ws_bytes, None) == 0
        assert np.all(ws[ws_bytes:] == 0xA5)
        got.append((gv, gl.view(np.uint32), ga.view(np.uint32)))
    for a, b in zip(*got):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("refdim", [2, 4] if FULL else [2])
def test_module_operands_bit_equal_to_the_fused_product_route(lib, refdim):
    """the train step's form: projection rows + reference points in, saved float32 locations / weights + records out; the
    backward writes grad_value and the projection rows' gradient"""
    M, L, P = 2, 4, 4
    pyr = np.asarray([(20, 27), (10, 14), (5, 7), (3, 4)], dtype=np.int64)
    starts = np.concatenate(([0], np.cumsum(pyr[:, 0] * pyr[:, 1])[:-1])).astype(np.int64)
    S = int((pyr[:, 0] * pyr[:, 1]).sum())
    rng = np.random.default_rng(23 + refdim)
    refp = []
    for H, W in pyr:
        ys, xs = np.meshgrid((np.arange(H) + 0.5) / H, (np.arange(W) + 0.5) / W, indexing="ij")
        refp.append(np.stack([xs.ravel(), ys.ravel()], -1))
    refp = np.concatenate(refp, 0)
    N, Lq = 1, S
    qproj = rng.standard_normal((N, Lq, M * L * P * 3))
    qproj[..., :M * L * P * 2] *= 2.0
    qb = np.ascontiguousarray(bf16_bits(qproj))
    if refdim == 2:
        ref = np.ascontiguousarray(np.broadcast_to(refp[None, :, None, :], (N, Lq, L, 2)), dtype=np.float32)
    else:
        wh = np.broadcast_to(np.asarray([0.2, 0.15]), (N, Lq, L, 2))
        ref = np.ascontiguousarray(np.concatenate([np.broadcast_to(refp[None, :, None, :], (N, Lq, L, 2)), wh], -1), dtype=np.float32)
    vb = np.ascontiguousarray(bf16_bits(rng.standard_normal((N, S, M, 32)) * 0.5))
    gob = np.ascontiguousarray(bf16_bits(rng.standard_normal((N, Lq, M * 32))))
    dims = (N, S, M, 32, L, Lq, P)
    out_ref = np.zeros((N, Lq, M * 32), dtype=np.uint16)
    loc_ref = np.full((N, Lq, M, L, P, 2), np.nan, dtype=np.float32)
    aw_ref = np.full((N, Lq, M, L, P), np.nan, dtype=np.float32)
    assert lib.msda_fused_forward_hs(CELL, BF16, p(vb), p(pyr), p(starts), p(pyr), p(qb), p(ref), refdim, *dims, p(out_ref),
                                     p(loc_ref), p(aw_ref), None) == 0
    ws_bytes = lib.msda_backward_workspace_bytes(BF16, p(pyr), *dims)
    ws = np.zeros(ws_bytes + 64, dtype=np.uint8)
    gv_ref, gq_ref = np.zeros(vb.shape, dtype=np.uint16), np.zeros(qb.shape, dtype=np.uint16)
    assert lib.msda_fused_backward_ws(FLAG_BF16_GV, BF16, p(vb), p(pyr), p(starts), p(pyr), p(loc_ref), p(aw_ref), p(ref), refdim,
                                      p(gob), *dims, p(gv_ref), p(gq_ref), p(ws), ws_bytes, None) == 0

    rec_bytes = lib.msda_records_bytes(BF16, p(pyr), *dims)
    assert rec_bytes > 0
    records = np.full(rec_bytes, 0xA5, dtype=np.uint8)
    out, loc, aw = np.zeros_like(out_ref), np.full_like(loc_ref, np.nan), np.full_like(aw_ref, np.nan)
    assert lib.msda_records_forward(BF16, p(vb), p(pyr), p(starts), p(pyr), p(qb), p(ref), refdim, p(loc), p(aw), *dims, p(out),
                                    p(records), rec_bytes, None) == 0
    assert np.array_equal(out, out_ref)
    assert np.array_equal(loc.view(np.uint32), loc_ref.view(np.uint32)) and np.array_equal(aw.view(np.uint32), aw_ref.view(np.uint32))
    for flags in ((FLAG_BF16_GV, FLAG_BF16_GV | FLAG_SWAP) if FULL else (FLAG_BF16_GV,)):        # (the no-loc call below runs the other order)
        gv, gq = np.zeros_like(gv_ref), np.zeros_like(gq_ref)
        ws = np.zeros(ws_bytes + 64, dtype=np.uint8)
        assert lib.msda_records_backward(flags, BF16, p(vb), p(pyr), p(starts), p(pyr), p(loc), p(aw), p(ref), refdim, p(gob), *dims,
                                         p(gv), None, None, p(gq), p(records), rec_bytes, p(ws), ws_bytes, None) == 0
        assert np.array_equal(gq, gq_ref), f"grad of the projection rows differs (flags {flags:#x})"
        assert np.array_equal(gv, gv_ref), f"grad_value differs (flags {flags:#x})"
    # the same call with the records as the WHOLE saved state: no float32 locations / weights written or read
    records2 = np.full(rec_bytes, 0x5A, dtype=np.uint8)
    out2 = np.zeros_like(out_ref)
    assert lib.msda_records_forward(BF16, p(vb), p(pyr), p(starts), p(pyr), p(qb), p(ref), refdim, None, None, *dims, p(out2),
                                    p(records2), rec_bytes, None) == 0
    assert np.array_equal(out2, out_ref)
    gv, gq = np.zeros_like(gv_ref), np.zeros_like(gq_ref)
    ws = np.zeros(ws_bytes + 64, dtype=np.uint8)
    assert lib.msda_records_backward(FLAG_BF16_GV | FLAG_SWAP, BF16, p(vb), p(pyr), p(starts), p(pyr), None, None, p(ref), refdim, p(gob),
                                     *dims, p(gv), None, None, p(gq), p(records2), rec_bytes, p(ws), ws_bytes, None) == 0
    assert np.array_equal(gq, gq_ref) and np.array_equal(gv, gv_ref)
    # one of the two pointers alone is refused
    assert lib.msda_records_forward(BF16, p(vb), p(pyr), p(starts), p(pyr), p(qb), p(ref), refdim, p(loc), None, *dims, p(out2),
                                    p(records2), rec_bytes, None) != 0


def test_far_samples_hand_grad_value_to_the_sorting_pass(lib):
    """uniform random locations: the forward's binning raises the "far" flag in the records' control block; the backward's
    patch pass returns at once and the gated sorting pass writes grad_value -- the same bits as the product route, which
    takes the same detour"""
    M = 1
    pyr = np.asarray([(16, 64), (8, 32), (4, 16), (2, 8)], dtype=np.int64)      # 1 x 4 cells: a level-0 patch sees 3 of them
    starts = np.concatenate(([0], np.cumsum(pyr[:, 0] * pyr[:, 1])[:-1])).astype(np.int64)
    S = int((pyr[:, 0] * pyr[:, 1]).sum())
    rng = np.random.default_rng(31)
    loc = rng.random((1, S, M, 4, 4, 2)).astype(np.float32)
    aw = rng.random((1, S, M, 4, 4))
    aw = (aw / aw.sum((-1, -2), keepdims=True)).astype(np.float32)
    vb = np.ascontiguousarray(bf16_bits(rng.standard_normal((1, S, M, 32)) * 0.5))
    gob = np.ascontiguousarray(bf16_bits(rng.standard_normal((1, S, M * 32))))
    dims = (1, S, M, 32, 4, S, 4)
    ws_bytes = lib.msda_backward_workspace_bytes(BF16, p(pyr), *dims)
    ws = np.zeros(ws_bytes + 64, dtype=np.uint8)
    gv_ref, gl_ref, ga_ref = np.zeros(vb.shape, dtype=np.uint16), np.full(loc.shape, np.nan, np.float32), np.full(aw.shape, np.nan, np.float32)
    assert lib.msda_backward_ws(4 | FLAG_BF16_GV, BF16, p(vb), p(pyr), p(starts), p(pyr), p(loc), p(aw), p(gob), *dims, p(gv_ref),
                                p(gl_ref), p(ga_ref), p(ws), ws_bytes, None) == 0
    rec_bytes = lib.msda_records_bytes(BF16, p(pyr), *dims)
    records = np.full(rec_bytes, 0xA5, dtype=np.uint8)
    out = np.zeros((1, S, M * 32), dtype=np.uint16)
    assert lib.msda_records_forward(BF16, p(vb), p(pyr), p(starts), p(pyr), None, None, 0, p(loc), p(aw), *dims, p(out), p(records),
                                    rec_bytes, None) == 0
    assert int(records[:256].view(np.int32)[60]) != 0                         # the "far" word
    gv, gl, ga = np.zeros_like(gv_ref), np.full_like(gl_ref, np.nan), np.full_like(ga_ref, np.nan)
    ws = np.zeros(ws_bytes + 64, dtype=np.uint8)
    assert lib.msda_records_backward(FLAG_BF16_GV, BF16, p(vb), p(pyr), p(starts), p(pyr), p(loc), p(aw), None, 0, p(gob), *dims,
                                     p(gv), p(gl), p(ga), None, p(records), rec_bytes, p(ws), ws_bytes, None) == 0
    assert np.array_equal(gl.view(np.uint32), gl_ref.view(np.uint32)) and np.array_equal(ga.view(np.uint32), ga_ref.view(np.uint32))
    # (the sorting pass orders a pixel's records by LDS-atomic arrival inside a wave: lane order on the hardware, thread
    #  scheduling on the host model -- there the product route itself differs from run to run in a handful of last bits)
    a, b = bf16_val(gv).astype(np.float64), bf16_val(gv_ref).astype(np.float64)
    assert np.abs(a - b).max() <= 2.0 ** -7 * np.abs(b).max() and np.mean(gv != gv_ref) < 1e-3


def test_unsupported_calls_are_refused(lib):
    pyr = np.asarray([(20, 27), (10, 14), (5, 7), (3, 4)], dtype=np.int64)
    S = int((pyr[:, 0] * pyr[:, 1]).sum())
    assert lib.msda_records_bytes(BF16, p(pyr), 1, S, 2, 32, 4, S, 4) > 0
    assert lib.msda_records_bytes(0, p(pyr), 1, S, 2, 32, 4, S, 4) == 0          # float32
    assert lib.msda_records_bytes(BF16, p(pyr), 1, S, 2, 32, 4, 300, 4) == 0     # not an encoder call
    assert lib.msda_records_bytes(BF16, p(pyr), 1, S, 2, 64, 4, S, 4) == 0       # D != 32
    assert lib.msda_records_bytes(BF16, None, 1, S, 2, 32, 4, S, 4) == 0         # no host shapes
    assert lib.msda_records_bytes(BF16, p(pyr), 1, S + 1, 2, 32, 4, S + 1, 4) == 0   # sum(H * W) != S


def test_autograd_function_with_the_route_on_and_off(lib, monkeypatch):
    """FusedMSDeformAttnFunction -- the product's Python: the records tensor travels from the forward call to the backward call
    as a saved tensor -- with msda.records_route on / off (+ records_swap), on the host-model library: the output and both
    gradients bit for bit.  (tests/test_zz_round5_gpu.py repeats this on the device.)"""
    import contextlib

    import torch

    from rlipv2_amd import _lib, msda

    i, vp, sz = ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t
    lib.msda_check_im2col_step.argtypes = [i, i]
    lib.msda_fused_supported.argtypes = [i, vp, i, *([i] * 7)]
    lib.msda_fused_forward.argtypes = [i, vp, vp, vp, vp, vp, i, *([i] * 7), vp, vp, vp, vp]
    monkeypatch.setattr(_lib, "lib", lambda: lib)
    monkeypatch.setattr(msda, "_on_device", lambda t: True)
    monkeypatch.setattr(msda, "_launch", lambda t: contextlib.nullcontext(None))

    M, L, P = 2, 4, 4
    pyr = [(20, 27), (10, 14), (5, 7), (3, 4)]
    shapes = torch.tensor(pyr, dtype=torch.int64)
    msda.attach_host_shapes(shapes, pyr)
    starts = torch.cat([shapes.new_zeros(1), (shapes[:, 0] * shapes[:, 1]).cumsum(0)[:-1]])
    S = int((shapes[:, 0] * shapes[:, 1]).sum())
    g = torch.Generator().manual_seed(5)
    refp = torch.cat([torch.stack(torch.meshgrid((torch.arange(H) + 0.5) / H, (torch.arange(W) + 0.5) / W, indexing="ij")[::-1], -1).reshape(-1, 2)
                      for H, W in pyr])
    ref = refp[None, :, None, :].expand(1, S, L, 2).contiguous()
    value0 = (0.5 * torch.randn(1, S, M, 32, generator=g)).to(torch.bfloat16)
    qproj0 = torch.randn(1, S, M * L * P * 3, generator=g)
    qproj0[..., :M * L * P * 2] *= 2.0
    qproj0 = qproj0.to(torch.bfloat16)
    gout = torch.randn(1, S, M * 32, generator=g).to(torch.bfloat16)

    def run(route, swap):
        monkeypatch.setattr(msda, "records_route", route)
        monkeypatch.setattr(msda, "records_swap", swap)
        value, qproj = value0.clone().requires_grad_(True), qproj0.clone().requires_grad_(True)
        out = msda.FusedMSDeformAttnFunction.apply(value, shapes, starts, qproj, ref, 64)
        variant_fwd = msda.last_variant["fwd"]
        out.backward(gout)
        return out.detach(), value.grad, qproj.grad, variant_fwd, msda.last_variant["bwd"]
    base = run(False, False)
    assert base[3] == "quad+geometry" and base[4] == "dest+geometry"
    for swap in ((False, True) if FULL else (True,)):
        got = run(True, swap)
        assert got[3] == "cell+geometry+records" and got[4] == "records+geometry"
        # (the two forward kernels sum a query's 16 samples differently: the output agrees to bfloat16 rounding, the gradients
        #  -- same records, same formulas -- bit for bit)
        assert float((got[0].float() - base[0].float()).abs().max()) <= 2.0 ** -6 * float(base[0].float().abs().max())
        assert torch.equal(got[1].view(torch.int16), base[1].view(torch.int16))
        assert torch.equal(got[2].view(torch.int16), base[2].view(torch.int16))


@pytest.mark.parametrize("N,M", [(2, 4), (3, 1)] if FULL else [(2, 4)])
def test_batches_and_the_xcd_placement(lib, N, M):
    """N * M = 8 takes the XCD-aware workgroup -> (image, head, cell) mapping (hardware block b runs on XCD b % 8), N * M = 3 the
    plain one; images beyond the first exercise the per-image offsets of the value rows, the records and the window tables"""
    pyr = [(16, 28), (8, 14), (4, 7), (2, 4)]            # 1 x 2 cells; 596 queries (<= 512: the product route is the few-query pass)
    parts = [make_problem(pyr, M, (1.5, 1.5, 1.0, 0.7), seed=40 + n) for n in range(N)]
    pyr, starts, S = parts[0][0], parts[0][1], parts[0][2]
    value = np.concatenate([q[3] for q in parts], 0)
    loc = np.ascontiguousarray(np.concatenate([q[4] for q in parts], 0))
    aw = np.ascontiguousarray(np.concatenate([q[5] for q in parts], 0))
    rng = np.random.default_rng(9)
    vb = np.ascontiguousarray(bf16_bits(value))
    gob = np.ascontiguousarray(bf16_bits(rng.standard_normal((N, S, M * 32))))
    sh, st = np.ascontiguousarray(pyr, dtype=np.int64), np.ascontiguousarray(starts, dtype=np.int64)
    dims = (N, S, M, 32, 4, S, 4)
    ws_bytes = lib.msda_backward_workspace_bytes(BF16, p(sh), *dims)
    gv_ref, gl_ref, ga_ref = np.zeros(vb.shape, dtype=np.uint16), np.full(loc.shape, np.nan, np.float32), np.full(aw.shape, np.nan, np.float32)
    ws = np.zeros(ws_bytes + 64, dtype=np.uint8)
    assert lib.msda_backward_ws(4 | FLAG_BF16_GV, BF16, p(vb), p(sh), p(st), p(sh), p(loc), p(aw), p(gob), *dims, p(gv_ref),
                                p(gl_ref), p(ga_ref), p(ws), ws_bytes, None) == 0
    rec_bytes = lib.msda_records_bytes(BF16, p(sh), *dims)
    records = np.full(rec_bytes, 0xA5, dtype=np.uint8)
    out = np.zeros((N, S, M * 32), dtype=np.uint16)
    assert lib.msda_records_forward(BF16, p(vb), p(sh), p(st), p(sh), None, None, 0, p(loc), p(aw), *dims, p(out), p(records),
                                    rec_bytes, None) == 0
    a64 = (value.astype(np.float64), pyr, starts, loc.astype(np.float64), aw.astype(np.float64))
    o_out = O.forward(*a64)
    assert np.abs(bf16_val(out) - o_out).max() <= 2.0 ** -7 * np.abs(o_out).max()
    gv, gl, ga = np.zeros_like(gv_ref), np.full_like(gl_ref, np.nan), np.full_like(ga_ref, np.nan)
    ws = np.zeros(ws_bytes + 64, dtype=np.uint8)
    assert lib.msda_records_backward(FLAG_BF16_GV | FLAG_SWAP, BF16, p(vb), p(sh), p(st), p(sh), p(loc), p(aw), None, 0, p(gob), *dims,
                                     p(gv), p(gl), p(ga), None, p(records), rec_bytes, p(ws), ws_bytes, None) == 0
    assert np.array_equal(gl.view(np.uint32), gl_ref.view(np.uint32)) and np.array_equal(ga.view(np.uint32), ga_ref.view(np.uint32))
    assert np.array_equal(gv, gv_ref)


@pytest.mark.skipif(not FULL, reason="70 s on the host model; RLIPV2_TEST_EMU_FULL=1 (green when written, round 5)")
def test_experiments_child_of_the_route_runs_on_the_model(lib, monkeypatch, capsys):
    """tools/experiments_r05.py --records (the child bench.py's `experiments` leg starts on the first hardware run) with the host
    model standing in for the device and a small pyramid for the 800 x 1333 one: its own logic -- four configurations, digests,
    the table it prints -- must not be what fails on the day."""
    import contextlib
    import json

    import torch

    from rlipv2_amd import _lib, msda
    from tools import experiments_r05 as X
    from tools import msda_inputs, patch_check, r03_experiments

    i, vp = ctypes.c_int, ctypes.c_void_p
    lib.msda_fused_forward.argtypes = [i, vp, vp, vp, vp, vp, i, *([i] * 7), vp, vp, vp, vp]
    monkeypatch.setattr(_lib, "lib", lambda: lib)
    monkeypatch.setattr(msda, "_on_device", lambda t: True)
    monkeypatch.setattr(msda, "_launch", lambda t: contextlib.nullcontext(None))
    pyr = [(20, 27), (10, 14), (5, 7), (3, 4)]
    real_inputs = msda_inputs.make_inputs
    monkeypatch.setattr(msda_inputs, "make_inputs", lambda N, **kw: real_inputs(1, pyramid=pyr, M=2, device="cpu", **{k: v for k, v in kw.items() if k != "device"}))
    monkeypatch.setattr(msda_inputs, "PYRAMID_800x1333", pyr)

    def fused_problem(N, inp):
        S = inp["value"].shape[1]
        g = torch.Generator().manual_seed(1)
        ref = torch.cat([torch.stack(torch.meshgrid((torch.arange(H) + 0.5) / H, (torch.arange(W) + 0.5) / W, indexing="ij")[::-1], -1).reshape(-1, 2)
                         for H, W in pyr])[None, :, None, :].expand(1, S, 4, 2).contiguous()
        qproj = torch.randn(1, S, 2 * 48, generator=g)
        qproj[..., :64] *= 2.5
        return qproj.bfloat16(), ref
    monkeypatch.setattr(r03_experiments, "fused_problem", fused_problem)
    monkeypatch.setattr(patch_check, "timed", lambda fn, iters=10: (fn(), 123.0)[1])
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    X.child_records()
    line = [ln for ln in capsys.readouterr().out.splitlines() if ln.startswith("RESULT ")][-1]
    out = json.loads(line[7:])
    assert set(out) == {"product", "cell_forward", "records", "records_swap"}
    assert out["records"]["equal_bits"] and out["records_swap"]["equal_bits"] and out["cell_forward"]["equal_bits"]
    assert out["records"]["accepted"] and out["records"]["vs_product"][0]["equal_bits"] and out["records"]["vs_product"][1]["differing_share"] == 0.0
    assert out["records"]["fwd_variant"] == "cell+geometry+records" and out["records"]["bwd_variant"] == "records+geometry"
    assert out["product"]["bwd_variant"] == "dest+geometry" and out["records"]["far_flag"] == 0
    assert out["records"]["out_max_diff_rel_to_max"] <= 2.0 ** -6 and "digest" not in json.dumps(out)


def test_far_samples_without_saved_locations_rebuild_them_for_the_sorting_pass(lib):
    """module operands with offsets of tens of pixels on a 1 x 4-cell pyramid, forward called WITHOUT float32 locations / weights:
    the forward raises the "far" flag, the backward rebuilds the two tensors from the group records inside its workspace
    (records_unbin_kernel, gated like the pass that reads them) and the sorting pass produces grad_value -- against the product's
    fused route, which saves them and takes the same detour"""
    M, L, P, refdim = 1, 4, 4, 2
    pyr = np.asarray([(16, 64), (8, 32), (4, 16), (2, 8)], dtype=np.int64)
    starts = np.concatenate(([0], np.cumsum(pyr[:, 0] * pyr[:, 1])[:-1])).astype(np.int64)
    S = int((pyr[:, 0] * pyr[:, 1]).sum())
    rng = np.random.default_rng(77)
    refp = np.concatenate([np.stack([g.ravel() for g in np.meshgrid((np.arange(W) + 0.5) / W, (np.arange(H) + 0.5) / H)], -1) for H, W in pyr], 0)
    ref = np.ascontiguousarray(np.broadcast_to(refp[None, :, None, :], (1, S, L, 2)), dtype=np.float32)
    qproj = rng.standard_normal((1, S, M * L * P * 3))
    qproj[..., :M * L * P * 2] *= 12.0                                           # pixels: across cells
    qb = np.ascontiguousarray(bf16_bits(qproj))
    vb = np.ascontiguousarray(bf16_bits(rng.standard_normal((1, S, M, 32)) * 0.5))
    gob = np.ascontiguousarray(bf16_bits(rng.standard_normal((1, S, M * 32))))
    dims = (1, S, M, 32, L, S, P)
    out_ref = np.zeros((1, S, M * 32), dtype=np.uint16)
    loc = np.full((1, S, M, L, P, 2), np.nan, dtype=np.float32)
    aw = np.full((1, S, M, L, P), np.nan, dtype=np.float32)
    assert lib.msda_fused_forward_hs(CELL, BF16, p(vb), p(pyr), p(starts), p(pyr), p(qb), p(ref), refdim, *dims, p(out_ref), p(loc),
                                     p(aw), None) == 0
    ws_bytes = lib.msda_backward_workspace_bytes(BF16, p(pyr), *dims)
    ws = np.zeros(ws_bytes + 64, dtype=np.uint8)
    gv_ref, gq_ref = np.zeros(vb.shape, dtype=np.uint16), np.zeros(qb.shape, dtype=np.uint16)
    assert lib.msda_fused_backward_ws(FLAG_BF16_GV, BF16, p(vb), p(pyr), p(starts), p(pyr), p(loc), p(aw), p(ref), refdim, p(gob),
                                      *dims, p(gv_ref), p(gq_ref), p(ws), ws_bytes, None) == 0
    assert int(ws[:256].view(np.int32)[60]) != 0                                 # the product route met far samples too
    rec_bytes = lib.msda_records_bytes(BF16, p(pyr), *dims)
    records = np.full(rec_bytes, 0xA5, dtype=np.uint8)
    out = np.zeros_like(out_ref)
    assert lib.msda_records_forward(BF16, p(vb), p(pyr), p(starts), p(pyr), p(qb), p(ref), refdim, None, None, *dims, p(out),
                                    p(records), rec_bytes, None) == 0
    assert np.array_equal(out, out_ref) and int(records[:256].view(np.int32)[60]) != 0
    gv, gq = np.zeros_like(gv_ref), np.zeros_like(gq_ref)
    ws = np.full(ws_bytes + 64, 0xEE, dtype=np.uint8)                            # garbage where the rebuilt tensors will live
    assert lib.msda_records_backward(FLAG_BF16_GV, BF16, p(vb), p(pyr), p(starts), p(pyr), None, None, p(ref), refdim, p(gob), *dims,
                                     p(gv), None, None, p(gq), p(records), rec_bytes, p(ws), ws_bytes, None) == 0
    assert np.all(ws[ws_bytes:] == 0xEE)
    assert np.array_equal(gq, gq_ref)
    a, b = bf16_val(gv).astype(np.float64), bf16_val(gv_ref).astype(np.float64)  # (host-model order of the sorting pass: see above)
    assert np.abs(a - b).max() <= 2.0 ** -7 * np.abs(b).max() and np.mean(gv != gv_ref) < 1e-3
