"""Host-side launch plan of the encoder backward's cell + patch route (csrc/msda_patch.hip: make_patch_plan), read through
the C ABI (msda_backward_plan_info, no device work): the index arithmetic the kernels rely on, checked on the CPU for the
pyramids of every configuration of the survey (800 x 1333 R50 / Swin strides, 640 x 640, ragged small ones)."""
import ctypes

import numpy as np
import pytest

from rlipv2_amd import _lib

LF, TF = 14, 5          # MSDA_PLAN_LEVEL_FIELDS, MSDA_PLAN_TAIL_FIELDS (include/rlipv2_msda.h)
NAMES = ("H", "W", "PY", "PX", "rad", "nby", "nbx", "invx", "sbase", "parts", "reps", "ibase", "nitems", "cell")


def plan(shapes, N=4, M=8, Lq=None, dtype=_lib.MSDA_BF16):
    L = _lib.lib()
    hs = np.asarray(shapes, dtype=np.int64).reshape(-1)
    S = int(sum(h * w for h, w in shapes))
    out = np.zeros(4 * LF + TF, dtype=np.int32)
    n = L.msda_backward_plan_info(dtype, hs.ctypes.data, N, S, M, 32, len(shapes), S if Lq is None else Lq, 4,
                                  out.ctypes.data, out.size)
    if n <= 0:
        return n, None
    lv = [dict(zip(NAMES, out[l * LF:(l + 1) * LF].tolist())) for l in range(4)]
    tail = dict(zip(("CY", "CX", "slots", "items", "bin_lds"), out[4 * LF:].tolist()))
    return n, (lv, tail)


def origin(l, pp, rad, nb, cells):          # nb_origin of msda_patch.hip, restated
    o = ((pp * 4) >> (4 - l)) - rad
    return min(max(o, 0), cells - nb)


PYRAMIDS = [
    [(100, 167), (50, 84), (25, 42), (13, 21)],      # 800 x 1333, strides 8..64 (configs 2-5)
    [(80, 80), (40, 40), (20, 20), (10, 10)],        # 640 x 640 (config 1)
    [(92, 138), (46, 69), (23, 35), (12, 18)],       # a padded batch's smaller image
    [(25, 34), (13, 17), (7, 9), (4, 5)],
    [(20, 27), (10, 14), (5, 7), (3, 4)],            # the goldens' pyramid
    [(10, 14), (5, 7), (3, 4), (2, 2)],
    [(1, 1), (1, 1), (1, 1), (1, 1)],
    [(7, 300), (4, 150), (2, 75), (1, 38)],          # very wide
]


@pytest.mark.parametrize("shapes", PYRAMIDS)
def test_plan_invariants(shapes):
    n, got = plan(shapes)
    assert n == 4 * LF + TF
    lv, t = got
    CY, CX = t["CY"], t["CX"]
    slots = items = 0
    for l, (v, (H, W)) in enumerate(zip(lv, shapes)):
        assert (v["H"], v["W"], v["cell"]) == (H, W, 16 >> l)
        assert (v["PY"], v["PX"]) == ((H + 3) // 4, (W + 3) // 4)
        # every pixel of every level belongs to a cell of the grid
        assert CY * v["cell"] >= H and CX * v["cell"] >= W
        assert 1 <= v["nby"] <= CY and 1 <= v["nbx"] <= CX and v["nby"] * v["nbx"] <= 128
        # slot -> (row, column) of the neighbourhood without a division: the kernels use (slot * invx) >> 16
        for slot in range(v["nby"] * v["nbx"]):
            assert (slot * v["invx"]) >> 16 == slot // v["nbx"]
        # a patch's neighbourhood lies inside the cell grid and contains the patch's own cell
        for pp, nb, cells in ((range(v["PY"]), v["nby"], CY), (range(v["PX"]), v["nbx"], CX)):
            prev = 0
            for q in pp:
                o = origin(l, q, v["rad"], nb, cells)
                home = (q * 4) >> (4 - l)
                assert 0 <= o <= cells - nb and o <= home < o + nb
                assert o >= prev                       # monotone: the patches that reach a cell are a contiguous range
                prev = o
        assert v["sbase"] == slots
        slots += v["PY"] * v["PX"] * v["nby"] * v["nbx"]
        assert v["parts"] in (1, 2, 4) and v["reps"] == 1          # (patches per wave: an ablation-build experiment)
    # workgroup items: coarsest level first, every patch of a level in exactly one (item, wave group)
    for l in (3, 2, 1, 0):
        v = lv[l]
        assert v["ibase"] == items
        per = 4 // v["parts"] * v["reps"]
        assert v["nitems"] == (v["PY"] * v["PX"] + per - 1) // per
        items += v["nitems"]
    assert (t["slots"], t["items"]) == (slots, items)
    assert 0 < t["bin_lds"] <= 60 * 1024 and t["bin_lds"] % 48 == 0       # 12 words per (patch in reach of a cell)


def test_mask_word_arithmetic_of_the_patch_pass():
    """word index -> (slot, word in slot) for 12-word slots without a division: (w * 683) >> 13 == w // 12, w < 2048"""
    w = np.arange(2048)
    assert np.array_equal((w * 683) >> 13, w // 12)


def test_route_is_refused_where_it_does_not_apply():
    enc = PYRAMIDS[0]
    assert plan(enc, dtype=_lib.MSDA_F32)[0] == 0            # bfloat16 only
    assert plan(enc, Lq=300)[0] == 0                         # encoder calls only (Lq == S)
    assert plan(enc[:3] + [(13, 22)], Lq=None)[0] > 0        # any consistent pyramid is fine ...
    L = _lib.lib()
    hs = np.asarray(enc, dtype=np.int64).reshape(-1)
    out = np.zeros(8, dtype=np.int32)
    S = sum(h * w for h, w in enc)
    assert L.msda_backward_plan_info(_lib.MSDA_BF16, hs.ctypes.data, 4, S, 8, 32, 4, S, 4, out.ctypes.data, out.size) == -1
    assert L.msda_backward_plan_info(_lib.MSDA_BF16, hs.ctypes.data, 4, S + 1, 8, 32, 4, S + 1, 4, out.ctypes.data, out.size) == 0


def test_closed_form_of_the_patch_ranges_equals_the_enumeration():
    """nb_range (csrc/msda_patch.hip, arms 3 / 4 of cell_backward_kernel): the patch rows (columns) whose neighbourhood contains
    a cell row (column), as an interval in closed form, against the enumeration over nb_origin the product kernel runs -- every
    level, grid sizes 1-24 cells, every radius of the plan, patch counts around the natural one (the formula restated here; the
    C code itself runs on the host model: tests/test_backward_emulated.py compares the arms with the product kernels bit for bit)"""
    def nb_origin(l, pp, rad, nb, cells):
        o = ((pp * 4) >> (4 - l)) - rad
        return 0 if o < 0 else (cells - nb if o > cells - nb else o)

    def closed(l, c, rad, nb, cells, P):
        s, v = 4 - l, c - nb + 1
        lo = 0 if v <= 0 else P if v > cells - nb else ((((v + rad) << s) + 3) >> 2)
        hi = P - 1 if c >= cells - nb else ((((c + rad + 1) << s) + 3) >> 2) - 1
        return lo, min(hi, P - 1)

    n = 0
    for l in range(4):
        for cells in range(1, 25):
            for rad in (1, 2, 3, 6):
                nb = min(2 * rad + 1, cells)
                for P in range(1, ((cells * 16) >> l) // 4 + 3):
                    for c in range(cells):
                        ts = [t for t in range(P) if nb_origin(l, t, rad, nb, cells) <= c < nb_origin(l, t, rad, nb, cells) + nb]
                        lo, hi = closed(l, c, rad, nb, cells, P)
                        assert (ts == [] and lo > hi) or (ts and (lo, hi) == (ts[0], ts[-1]) and ts == list(range(lo, hi + 1)))
                        n += 1
    assert n > 100000


@pytest.mark.parametrize("shapes", PYRAMIDS)
@pytest.mark.parametrize("N,M", [(4, 8), (1, 1), (8, 8)])
def test_records_buffer_layout_follows_the_plan(shapes, N, M):
    """msda_records_bytes (the saved state of the records route, include/rlipv2_msda.h) = control block + one window table per
    (image, head, cell) + 2-byte sample records ([level][352 query slots][point] per cell: whole 128-byte lines; round 5 stored 16
    bytes per sample) + the patch pass's masks and group records, as the plan
    sizes them; every query has a slot; the sorting fallback's rebuilt locations / weights fit the workspace the ABI asks for"""
    L = _lib.lib()
    n, p = plan(shapes, N=N, M=M)
    hs = np.asarray(shapes, dtype=np.int64).reshape(-1)
    S = int(sum(h * w for h, w in shapes))
    dims = (N, S, M, 32, 4, S, 4)
    L.msda_records_bytes.restype = ctypes.c_size_t
    got = int(L.msda_records_bytes(_lib.MSDA_BF16, hs.ctypes.data, *dims))
    if n <= 0:
        assert got == 0
        return
    lv, tail = p
    cells = tail["CY"] * tail["CX"]
    items = N * M * cells
    masks = N * M * tail["slots"] * 48
    group_records = N * M * 4 * cells * 340 * 48
    if tail["bin_lds"] > 60 * 1024:                                   # (the forward keeps the cell's mask table in its window region)
        assert got == 0
        return
    assert got == 256 + items * 128 + items * 4 * 352 * 4 * 2 + masks + group_records
    assert cells * 340 >= S and sum(min(16 >> l, 16) ** 2 for l in range(4)) == 340
    ws = int(L.msda_backward_workspace_bytes(_lib.MSDA_BF16, hs.ctypes.data, *dims))
    assert ws >= N * S * M * 16 * 12
