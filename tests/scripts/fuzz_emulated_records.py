"""Random encoder calls through the records route (csrc/msda_cell_forward.inc EMIT, csrc/msda_cell_records.inc) of the MSDA
library built for the lane-level workgroup model: ragged and degenerate pyramids (levels that do not halve, 1-pixel levels),
1-3 images, 1-8 heads, offsets from half a pixel to far outside the cells' reach (the gated sorting fallback), the op's operands
(refdim 0) and the module's (refdim 2 / 4, with and without saved float32 locations / weights).  Checks: every result against
the oracle at the tolerances of tests/test_msda_gpu.py; bit-equality with the product route (msda_backward_ws /
msda_fused_backward_ws) whenever that route is the cell + patch pair too (more than 512 queries, no far sample).
Not part of the test suite (minutes of host time).
    usage: tools/emu/build_lib.sh /tmp/libmsda_emu.so && python tests/scripts/fuzz_emulated_records.py <seed> <seconds>
(lives under tests/: it uses the oracle as the checker)"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from oracle import msda_oracle as O  # noqa: E402
from conftest import kink_samples  # noqa: E402
from test_cell_forward_emulated import bf16_bits, bf16_val  # noqa: E402

L = ctypes.CDLL("/tmp/libmsda_emu.so")
vp, i, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
d = [i] * 7
L.msda_records_bytes.argtypes = [i, vp, *d]
L.msda_records_bytes.restype = sz
L.msda_records_forward.argtypes = [i, vp, vp, vp, vp, vp, vp, i, vp, vp, *d, vp, vp, sz, vp]
L.msda_records_backward.argtypes = [i, i, vp, vp, vp, vp, vp, vp, vp, i, vp, *d, vp, vp, vp, vp, vp, sz, vp, sz, vp]
L.msda_backward_ws.argtypes = [i, i, vp, vp, vp, vp, vp, vp, vp, *d, vp, vp, vp, vp, sz, vp]
L.msda_backward_workspace_bytes.argtypes = [i, vp, *d]
L.msda_backward_workspace_bytes.restype = sz
L.msda_fused_forward_hs.argtypes = [i, i, vp, vp, vp, vp, vp, vp, i, *d, vp, vp, vp, vp]
L.msda_fused_backward_ws.argtypes = [i, i, vp, vp, vp, vp, vp, vp, vp, i, vp, *d, vp, vp, vp, sz, vp]
BF16, GV16, SWAP, CELL = 2, 0x200, 0x400, 6
p = lambda a: a.ctypes.data if a is not None else None                                        # noqa: E731
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
t_end = time.time() + float(sys.argv[2] if len(sys.argv) > 2 else 300)
n = fails = refused = 0
while time.time() < t_end:
    n += 1
    H0, W0 = int(rng.integers(2, 49)), int(rng.integers(2, 49))
    pyr = [(H0, W0)]
    for _ in range(3):
        h, w = pyr[-1]
        pyr.append((max(1, (h + 1) // 2 if rng.random() < 0.8 else int(rng.integers(1, h + 1))),
                    max(1, (w + 1) // 2 if rng.random() < 0.8 else int(rng.integers(1, w + 1)))))
    pyr = np.asarray(pyr, dtype=np.int64)
    starts = np.concatenate(([0], np.cumsum(pyr[:, 0] * pyr[:, 1])[:-1])).astype(np.int64)
    S = int((pyr[:, 0] * pyr[:, 1]).sum())
    N, M = int(rng.integers(1, 3)), int(rng.choice([1, 2, 3, 4, 8]))
    refdim = int(rng.choice([0, 2, 4]))
    save_loc = refdim == 0 or rng.random() < 0.4
    spread = float(rng.choice([0.5, 2.0, 6.0, 40.0]))
    flags = GV16 | (SWAP if rng.random() < 0.5 else 0)
    dims = (N, S, M, 32, 4, S, 4)
    desc = f"#{n} pyr={pyr.tolist()} N={N} M={M} refdim={refdim} saved_loc={save_loc} spread={spread} swap={bool(flags & SWAP)}"
    rec_bytes = L.msda_records_bytes(BF16, p(pyr), *dims)
    if rec_bytes == 0:
        refused += 1
        print("skip", desc, "(msda_records_bytes == 0)", flush=True)
        continue
    refp = np.concatenate([np.stack([g.ravel() for g in np.meshgrid((np.arange(W) + 0.5) / W, (np.arange(H) + 0.5) / H)], -1) for H, W in pyr], 0)
    norm = np.stack([pyr[:, 1], pyr[:, 0]], -1).astype(np.float64)
    vb = np.ascontiguousarray(bf16_bits(rng.standard_normal((N, S, M, 32)) * 0.5))
    gob = np.ascontiguousarray(bf16_bits(rng.standard_normal((N, S, M * 32))))
    value, gout = bf16_val(vb).astype(np.float64), bf16_val(gob).astype(np.float64)
    ws_bytes = L.msda_backward_workspace_bytes(BF16, p(pyr), *dims)
    out = np.zeros((N, S, M * 32), dtype=np.uint16)
    records = np.full(rec_bytes, 0xA5, dtype=np.uint8)
    gv = np.zeros(vb.shape, dtype=np.uint16)
    try:
        if refdim == 0:
            off = rng.standard_normal((N, S, M, 4, 4, 2)) * spread
            loc = (refp[None, :, None, None, None, :] + off / norm[None, None, None, :, None, :]).astype(np.float32)
            aw = rng.random((N, S, M, 4, 4))
            aw = (aw / aw.sum((-1, -2), keepdims=True)).astype(np.float32)
            assert L.msda_records_forward(BF16, p(vb), p(pyr), p(starts), p(pyr), None, None, 0, p(loc), p(aw), *dims, p(out), p(records), rec_bytes, None) == 0
            far = int(records[:256].view(np.int32)[60])
            gl, ga = np.full(loc.shape, np.nan, np.float32), np.full(aw.shape, np.nan, np.float32)
            ws = np.zeros(ws_bytes + 64, dtype=np.uint8)
            assert L.msda_records_backward(flags, BF16, p(vb), p(pyr), p(starts), p(pyr), p(loc), p(aw), None, 0, p(gob), *dims, p(gv), p(gl), p(ga),
                                           None, p(records), rec_bytes, p(ws), ws_bytes, None) == 0
            a = (value, pyr, starts, loc.astype(np.float64), aw.astype(np.float64))
            o_out = O.forward(*a)
            o_gv, o_gl, o_ga = O.backward(*a, gout)
            tol = 2.0 ** -7
            np.testing.assert_allclose(bf16_val(out), o_out, rtol=tol, atol=1e-3 * max(1.0, float(np.abs(o_out).max())))
            np.testing.assert_allclose(bf16_val(gv), o_gv, rtol=tol, atol=1e-3 * max(1.0, float(np.abs(o_gv).max())))
            np.testing.assert_allclose(ga, o_ga, rtol=1e-4, atol=1e-5 * max(1.0, float(np.abs(o_ga).max())))
            keep = ~kink_samples(dict(loc=loc, shapes=pyr))
            np.testing.assert_allclose(gl[keep], o_gl[keep], rtol=1e-4, atol=1e-5 * max(1.0, float(np.abs(o_gl).max())))
            equal = ""
            if S > 512 and far == 0:
                gv2, gl2, ga2 = np.zeros_like(gv), np.full_like(gl, np.nan), np.full_like(ga, np.nan)
                ws = np.zeros(ws_bytes + 64, dtype=np.uint8)
                assert L.msda_backward_ws(4 | GV16, BF16, p(vb), p(pyr), p(starts), p(pyr), p(loc), p(aw), p(gob), *dims, p(gv2), p(gl2), p(ga2), p(ws), ws_bytes, None) == 0
                assert np.array_equal(gl.view(np.uint32), gl2.view(np.uint32)) and np.array_equal(ga.view(np.uint32), ga2.view(np.uint32)) and np.array_equal(gv, gv2), "bits differ from the product route"
                equal = " bit-equal"
        else:
            qproj = rng.standard_normal((N, S, M * 48))
            qproj[..., :M * 32] *= spread
            qb = np.ascontiguousarray(bf16_bits(qproj))
            qd = bf16_val(qb).astype(np.float64)
            ref2 = np.broadcast_to(refp[None, :, None, :], (N, S, 4, 2))
            if refdim == 2:
                ref = np.ascontiguousarray(ref2, dtype=np.float32)
            else:
                ref = np.ascontiguousarray(np.concatenate([ref2, rng.uniform(0.05, 0.4, (N, S, 4, 2))], -1), dtype=np.float32)
            loc = np.full((N, S, M, 4, 4, 2), np.nan, np.float32) if save_loc else None
            aw = np.full((N, S, M, 4, 4), np.nan, np.float32) if save_loc else None
            assert L.msda_records_forward(BF16, p(vb), p(pyr), p(starts), p(pyr), p(qb), p(ref), refdim, p(loc), p(aw), *dims, p(out), p(records), rec_bytes, None) == 0
            far = int(records[:256].view(np.int32)[60])
            gq = np.zeros(qb.shape, dtype=np.uint16)
            ws = np.full(ws_bytes + 64, 0xEE, dtype=np.uint8)
            assert L.msda_records_backward(flags, BF16, p(vb), p(pyr), p(starts), p(pyr), p(loc), p(aw), p(ref), refdim, p(gob), *dims, p(gv), None, None,
                                           p(gq), p(records), rec_bytes, p(ws), ws_bytes, None) == 0
            offs = qd[..., :M * 32].reshape(N, S, M, 4, 4, 2)
            lg = qd[..., M * 32:].reshape(N, S, M, 16)
            e = np.exp(lg - lg.max(-1, keepdims=True))
            awd = (e / e.sum(-1, keepdims=True)).reshape(N, S, M, 4, 4)
            r64 = ref.astype(np.float64)
            if refdim == 2:
                scale = 1.0 / norm[None, None, None, :, None, :]
                locd = r64[:, :, None, :, None, :] + offs * scale
            else:
                scale = (r64[..., 2:] * 0.5 / 4)[:, :, None, :, None, :]
                locd = r64[:, :, None, :, None, :2] + offs * scale
            a = (value, pyr, starts, locd, awd)
            o_out = O.forward(*a)
            o_gv, o_gl, o_ga = O.backward(*a, gout)
            tol = 2.0 ** -7
            np.testing.assert_allclose(bf16_val(out), o_out, rtol=tol, atol=2e-3 * max(1.0, float(np.abs(o_out).max())))
            np.testing.assert_allclose(bf16_val(gv), o_gv, rtol=tol, atol=2e-3 * max(1.0, float(np.abs(o_gv).max())))
            g_off = o_gl * np.broadcast_to(scale, o_gl.shape)
            g_logit = awd * (o_ga - (awd * o_ga).sum((-1, -2), keepdims=True))
            ref_gq = np.concatenate([g_off.reshape(N, S, -1), g_logit.reshape(N, S, -1)], -1)
            keep = np.broadcast_to(~kink_samples({"loc": locd, "shapes": pyr}, 1e-3)[..., None], o_gl.shape).reshape(N, S, -1)
            keep = np.concatenate([keep, np.ones((N, S, M * 16), dtype=bool)], -1)
            assert np.abs(bf16_val(gq) - ref_gq)[keep].max() <= 2.0 ** -6 * max(1e-6, float(np.abs(ref_gq).max())), "grad of the projection rows"
            equal = ""
            if S > 512 and far == 0:
                loc2, aw2 = np.full((N, S, M, 4, 4, 2), np.nan, np.float32), np.full((N, S, M, 4, 4), np.nan, np.float32)
                out2, gv2, gq2 = np.zeros_like(out), np.zeros_like(gv), np.zeros_like(gq)
                assert L.msda_fused_forward_hs(CELL, BF16, p(vb), p(pyr), p(starts), p(pyr), p(qb), p(ref), refdim, *dims, p(out2), p(loc2), p(aw2), None) == 0
                ws = np.zeros(ws_bytes + 64, dtype=np.uint8)
                assert L.msda_fused_backward_ws(GV16, BF16, p(vb), p(pyr), p(starts), p(pyr), p(loc2), p(aw2), p(ref), refdim, p(gob), *dims, p(gv2), p(gq2), p(ws), ws_bytes, None) == 0
                assert np.array_equal(out, out2) and np.array_equal(gq, gq2) and np.array_equal(gv, gv2), "bits differ from the product route"
                if save_loc:
                    assert np.array_equal(loc.view(np.uint32), loc2.view(np.uint32)) and np.array_equal(aw.view(np.uint32), aw2.view(np.uint32))
                equal = " bit-equal"
        print("ok  ", desc, f"far={far}{equal}", flush=True)
    except Exception as ex:                                                                    # noqa: BLE001
        fails += 1
        print("FAIL", desc, str(ex)[:300].replace("\n", " | "), flush=True)
print(f"{n} problems, {refused} refused by the route, {fails} failures")
