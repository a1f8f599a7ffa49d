"""The MSDA part of the product library -- msda_api.hip and every MSDA kernel file, the SAME sources hipcc compiles -- built for the CPU against the lane-level workgroup model (tools/emu/build_lib.sh) and driven through its
C ABI (include/rlipv2_msda.h) with host arrays: the reference-generated goldens of tests/golden/ in float64, float32 and
bfloat16, kernel variant by kernel variant, the destination-stationary backward family (sorting pass, few-query pass, and
for bfloat16 encoder calls cell_backward_kernel + patch_dest_kernel), the fused geometry entry points.

A CPU mirror of the golden tests of tests/test_msda_gpu.py.  It exists because the GPU pool can be closed (second half of
round 3): kernel LOGIC stays checkable.  It is test infrastructure -- nothing in rlipv2_amd/ can load this library, the
product never loads this library (CPU tensors run the op's CPU twins, csrc/msda_cpu.cpp) -- and it says nothing about code generation, timing or the hardware itself."""
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
from conftest import boundary_samples, kink_samples, load_golden  # noqa: E402
from test_cell_forward_emulated import CLANG, bf16_bits, bf16_val, make_problem  # noqa: E402

pytestmark = pytest.mark.skipif(not os.path.exists(CLANG), reason="needs the ROCm clang++ (ext_vector_type) as host compiler")

F32, F64, BF16 = 0, 1, 2
VAR = {"auto": 0, "generic": 1, "quad": 2, "window": 3, "dest": 4, "coarse": 5, "cell": 6}
FLAG_BF16_GV = 0x200


class EmuLib:
    def __init__(self, path):
        L = self.L = ctypes.CDLL(path)
        vp, i = ctypes.c_void_p, ctypes.c_int
        d = [i] * 7
        L.msda_forward_ex.argtypes = [i, i, vp, vp, vp, vp, vp, *d, vp, vp]
        L.msda_forward_hs.argtypes = [i, i, vp, vp, vp, vp, vp, vp, *d, vp, vp]
        L.msda_backward_ex.argtypes = [i, i, vp, vp, vp, vp, vp, vp, *d, vp, vp, vp, vp]
        L.msda_backward_ws.argtypes = [i, i, vp, vp, vp, vp, vp, vp, vp, *d, vp, vp, vp, vp, ctypes.c_size_t, vp]
        L.msda_backward_workspace_bytes.argtypes = [i, vp, *d]
        L.msda_backward_workspace_bytes.restype = ctypes.c_size_t
        L.msda_prepare_forward.argtypes = [i, vp, vp, i, vp, i, i, i, i, vp, vp, vp]
        L.msda_fused_forward.argtypes = [i, vp, vp, vp, vp, vp, i, *d, vp, vp, vp, vp]
        L.msda_fused_forward_hs.argtypes = [i, i, vp, vp, vp, vp, vp, vp, i, *d, vp, vp, vp, vp]
        L.msda_fused_backward_ws.argtypes = [i, i, vp, vp, vp, vp, vp, vp, vp, i, vp, *d, vp, vp, vp, ctypes.c_size_t, vp]
        L.msda_fused_supported.argtypes = [i, vp, i, *d]

    @staticmethod
    def _arr(a, dt):
        if dt == BF16:
            return np.ascontiguousarray(bf16_bits(a))
        return np.ascontiguousarray(a, dtype=np.float64 if dt == F64 else np.float32)

    def run(self, fwd, bwd, dt, g, host_shapes=True):
        """-> out, g_value, g_loc, g_aw as float64 arrays (bf16 results decoded); bwd None: forward only"""
        aux = np.float64 if dt == F64 else np.float32
        v = self._arr(g["value"], dt)
        loc, aw = np.ascontiguousarray(g["loc"], dtype=aux), np.ascontiguousarray(g["aw"], dtype=aux)
        go = self._arr(g["grad_out"], dt)
        sh, st = np.ascontiguousarray(g["shapes"], dtype=np.int64), np.ascontiguousarray(g["starts"], dtype=np.int64)
        N, S, M, D = g["value"].shape
        Lq, nL, P = loc.shape[1], loc.shape[3], loc.shape[4]
        dims = (N, S, M, D, nL, Lq, P)
        p = lambda a: a.ctypes.data                                                  # noqa: E731
        out = np.zeros((N, Lq, M * D), dtype=v.dtype)
        if fwd == "cell":
            rc = self.L.msda_forward_hs(VAR[fwd], dt, p(v), p(sh), p(st), p(sh), p(loc), p(aw), *dims, p(out), None)
        else:
            rc = self.L.msda_forward_ex(VAR[fwd], dt, p(v), p(sh), p(st), p(loc), p(aw), *dims, p(out), None)
        assert rc == 0, f"forward {fwd}: status {rc}"
        dec = lambda a: bf16_val(a).astype(np.float64) if a.dtype == np.uint16 else a.astype(np.float64)   # noqa: E731
        if bwd is None:
            return dec(out), None, None, None
        g_loc, g_aw = np.full(loc.shape, np.nan, dtype=aux), np.full(aw.shape, np.nan, dtype=aux)
        ws_bytes = self.L.msda_backward_workspace_bytes(dt, p(sh), *dims) if host_shapes and bwd in ("auto", "dest") else 0
        if ws_bytes:
            flags = FLAG_BF16_GV if dt == BF16 else 0
            g_value = np.zeros(v.shape, dtype=v.dtype)
            ws = np.zeros(ws_bytes + 64, dtype=np.uint8)
            rc = self.L.msda_backward_ws(VAR[bwd] | flags, dt, p(v), p(sh), p(st), p(sh), p(loc), p(aw), p(go), *dims,
                                         p(g_value), p(g_loc), p(g_aw), p(ws), ws_bytes, None)
        else:
            g_value = np.zeros(v.shape, dtype=aux)                                   # (float32 staging for bfloat16 inputs)
            rc = self.L.msda_backward_ex(VAR[bwd], dt, p(v), p(sh), p(st), p(loc), p(aw), p(go), *dims, p(g_value),
                                         p(g_loc), p(g_aw), None)
        assert rc == 0, f"backward {bwd}: status {rc}"
        return dec(out), dec(g_value), g_loc.astype(np.float64), g_aw.astype(np.float64)


FULL = os.environ.get("RLIPV2_TEST_EMU_FULL", "0") == "1"     # the whole matrix takes ~10 minutes (the generic kernels' wave
#                                                               reductions are thousands of rendezvous per workgroup)


@pytest.fixture(scope="module")
def lib(emu_library):
    return EmuLib(emu_library())


def close32(got, ref, rtol=1e-4, atol_rel=1e-5):
    np.testing.assert_allclose(got, ref, rtol=rtol, atol=atol_rel * max(1.0, float(np.abs(ref).max())))


def variants(g, dt):
    """(forward, backward) kernel pairs; the generic kernels only on the tiny test.py cases unless FULL"""
    D, L, P = g["value"].shape[-1], g["loc"].shape[3], g["loc"].shape[4]
    small = g["value"].shape[1] * g["loc"].shape[1] < 5000
    v = [("generic", "generic")] if (small or FULL) else []
    if D == 32 and L == 4 and P == 4 and dt != F64:
        v += [("quad", "quad"), ("quad", "dest")]
    if small or FULL:
        v.append(("auto", "auto"))
    return v


@pytest.mark.parametrize("case", ["testpy_d2", "testpy_d32", "testpy_d71"] + (["model_dec"] if FULL else []))
def test_goldens_float64(lib, case):
    g = load_golden(case)
    out, gv, gl, ga = lib.run("generic", "generic", F64, g)
    np.testing.assert_allclose(out, g["out_f64"], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(gv, g["g_value_f64"], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(ga, g["g_aw_f64"], rtol=1e-5, atol=1e-8)
    keep = ~boundary_samples(g)
    np.testing.assert_allclose(gl[keep], g["g_loc_f64"][keep], rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("case", ["testpy_d30", "testpy_d32"] + (["model_enc", "model_dec"] if FULL else []))
def test_goldens_float32_every_variant(lib, case):
    """model_enc: direct-gather forward, K1 + the sorting pass (bin / dest / combine kernels) for grad_value"""
    g = load_golden(case)
    keep = ~kink_samples(g)
    for fwd, bwd in variants(g, F32):
        out, gv, gl, ga = lib.run(fwd, bwd, F32, g)
        close32(out, g["out_f32"])
        close32(gv, g["g_value_f32"])
        close32(ga, g["g_aw_f32"])
        close32(gl[keep], g["g_loc_f32"][keep])


@pytest.mark.parametrize("case", ["model_enc", "model_dec"])
def test_goldens_bfloat16(lib, case):
    """model_enc (Lq == S) through "dest" is the matrix-core route: cell_backward_kernel + patch_dest_kernel, and its forward
    additionally runs through the experimental "cell" variant; model_dec is the few-query pass (sparse_dest_kernel)"""
    g = load_golden(case)
    g = dict(g, value=bf16_val(bf16_bits(g["value"])), grad_out=bf16_val(bf16_bits(g["grad_out"])))
    args = (g["value"].astype(np.float64), g["shapes"], g["starts"], g["loc"].astype(np.float64), g["aw"].astype(np.float64))
    ref_out = O.forward(*args)
    ref_gv, ref_gl, ref_ga = O.backward(*args, g["grad_out"].astype(np.float64))
    keep = ~kink_samples(g)
    todo = [("quad", "dest")] + ([("quad", "quad"), ("generic", "generic"), ("auto", "auto")] if FULL else [])
    for fwd, bwd in todo:
        out, gv, gl, ga = lib.run(fwd, bwd, BF16, g)
        np.testing.assert_allclose(out, ref_out, rtol=2.0 ** -7, atol=1e-3 * float(np.abs(ref_out).max()))
        np.testing.assert_allclose(gv, ref_gv, rtol=2.0 ** -7, atol=1e-3 * float(np.abs(ref_gv).max()))
        close32(ga, ref_ga)
        close32(gl[keep], ref_gl[keep])
    if case == "model_enc":
        out = lib.run("cell", None, BF16, g)[0]
        np.testing.assert_allclose(out, ref_out, rtol=2.0 ** -7, atol=1e-3 * float(np.abs(ref_out).max()))


def _encoder_problem(pyr, N, M, seed, uniform=False):
    rng = np.random.default_rng(seed)
    pyr = np.asarray(pyr, dtype=np.int64)
    starts = np.concatenate(([0], np.cumsum(pyr[:, 0] * pyr[:, 1])[:-1])).astype(np.int64)
    S = int((pyr[:, 0] * pyr[:, 1]).sum())
    ref = []
    for H, W in pyr:
        ys, xs = np.meshgrid((np.arange(H) + 0.5) / H, (np.arange(W) + 0.5) / W, indexing="ij")
        ref.append(np.stack([xs.ravel(), ys.ravel()], -1))
    ref = np.concatenate(ref, 0)                                                     # [S, 2]
    value = bf16_val(bf16_bits(rng.standard_normal((N, S, M, 32)) * 0.5)).astype(np.float32)
    grad_out = bf16_val(bf16_bits(rng.standard_normal((N, S, M * 32)))).astype(np.float32)
    return pyr, starts, S, ref, value, grad_out, rng


def test_fused_geometry_route_of_the_train_step(lib):
    """msda_fused_forward / msda_fused_backward_ws on a bfloat16 encoder call (the route FusedMSDeformAttnFunction takes in
    the train step: geometry as prologue of the gather kernel and as epilogue of cell_backward_kernel, grad_value from
    patch_dest_kernel) against the oracle fed with float64 locations / weights from the same projection rows."""
    M, L, P = 2, 4, 4
    pyr, starts, S, ref2, value, grad_out, rng = _encoder_problem([(20, 27), (10, 14), (5, 7), (3, 4)], 1, M, seed=21)
    N, Lq = 1, S
    qproj = rng.standard_normal((N, Lq, M * L * P * 3))
    qproj[..., :M * L * P * 2] *= 2.0                                                # offsets of a few pixels
    qproj = bf16_val(bf16_bits(qproj)).astype(np.float32)
    ref = np.ascontiguousarray(np.broadcast_to(ref2[None, :, None, :], (N, Lq, L, 2)), dtype=np.float32)
    vb, qb, gob = bf16_bits(value), bf16_bits(qproj), bf16_bits(grad_out)
    dims = (N, S, M, 32, L, Lq, P)
    p = lambda a: a.ctypes.data                                                       # noqa: E731
    Lb = lib.L
    assert Lb.msda_fused_supported(BF16, p(pyr), 2, *dims) == 2
    out = np.zeros((N, Lq, M * 32), dtype=np.uint16)
    loc = np.full((N, Lq, M, L, P, 2), np.nan, dtype=np.float32)
    aw = np.full((N, Lq, M, L, P), np.nan, dtype=np.float32)
    assert Lb.msda_fused_forward(BF16, p(vb), p(pyr), p(starts), p(qb), p(ref), 2, *dims, p(out), p(loc), p(aw), None) == 0
    # the experimental one-kernel form of the same call (cell_forward_kernel<2>: geometry + saves + LDS windows + MFMA):
    # the saved locations / weights must be the product kernel's bit for bit (same geometry instructions)
    out_c = np.zeros_like(out)
    loc_c, aw_c = np.full_like(loc, np.nan), np.full_like(aw, np.nan)
    assert Lb.msda_fused_forward_hs(VAR["cell"], BF16, p(vb), p(pyr), p(starts), p(pyr), p(qb), p(ref), 2, *dims, p(out_c),
                                    p(loc_c), p(aw_c), None) == 0
    assert np.array_equal(loc_c.view(np.uint32), loc.view(np.uint32)) and np.array_equal(aw_c.view(np.uint32), aw.view(np.uint32))
    ws_bytes = Lb.msda_backward_workspace_bytes(BF16, p(pyr), *dims)
    assert ws_bytes > 0
    ws = np.zeros(ws_bytes + 64, dtype=np.uint8)
    gv, gq = np.zeros(vb.shape, dtype=np.uint16), np.zeros(qb.shape, dtype=np.uint16)
    assert Lb.msda_fused_backward_ws(FLAG_BF16_GV, BF16, p(vb), p(pyr), p(starts), p(pyr), p(loc), p(aw), p(ref), 2, p(gob),
                                     *dims, p(gv), p(gq), p(ws), ws_bytes, None) == 0
    # float64 restatement of ms_deform_attn.py:101-109
    qd = qproj.astype(np.float64)
    off = qd[..., :M * L * P * 2].reshape(N, Lq, M, L, P, 2)
    lg = qd[..., M * L * P * 2:].reshape(N, Lq, M, L * P)
    e = np.exp(lg - lg.max(-1, keepdims=True))
    awd = (e / e.sum(-1, keepdims=True)).reshape(N, Lq, M, L, P)
    norm = np.stack([pyr[:, 1], pyr[:, 0]], -1).astype(np.float64)
    locd = ref.astype(np.float64)[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
    np.testing.assert_allclose(loc, locd, rtol=1e-5, atol=1e-6)                      # what the forward saved for the backward
    np.testing.assert_allclose(aw, awd, rtol=1e-4, atol=1e-6)
    a = (value.astype(np.float64), pyr, starts, locd, awd)
    ref_out = O.forward(*a)
    ref_gv, ref_gl, ref_ga = O.backward(*a, grad_out.astype(np.float64))
    tol = 2.0 ** -7
    assert np.abs(bf16_val(out) - ref_out).max() <= tol * np.abs(ref_out).max()
    assert np.abs(bf16_val(out_c) - ref_out).max() <= tol * np.abs(ref_out).max()
    assert np.abs(bf16_val(gv) - ref_gv).max() <= tol * np.abs(ref_gv).max()
    g_off = ref_gl / norm[None, None, None, :, None, :]
    g_logit = awd * (ref_ga - (awd * ref_ga).sum((-1, -2), keepdims=True))
    ref_gq = np.concatenate([g_off.reshape(N, Lq, -1), g_logit.reshape(N, Lq, -1)], -1)
    keep = np.broadcast_to(~kink_samples({"loc": locd, "shapes": pyr}, 1e-3)[..., None], ref_gl.shape).reshape(N, Lq, -1)
    keep = np.concatenate([keep, np.ones((N, Lq, M * L * P), dtype=bool)], -1)
    assert np.abs(bf16_val(gq) - ref_gq)[keep].max() <= 2.0 ** -6 * np.abs(ref_gq).max()


def test_samples_out_of_reach_take_the_sorting_pass_within_the_same_call(lib):
    """bfloat16 encoder call whose samples leave their cell's neighbourhood (uniform random locations): the device-side
    "far" flag must hand grad_value to the sorting pass (bin / dest / combine kernels, gated launches) with no host decision
    -- and the result must be the oracle's either way."""
    M = 1
    pyr, starts, S, ref2, value, grad_out, rng = _encoder_problem([(40, 54), (20, 27), (10, 14), (5, 7)], 1, M, seed=31)
    loc = rng.random((1, S, M, 4, 4, 2)).astype(np.float32)                         # anywhere in the image
    aw = rng.random((1, S, M, 4, 4))
    aw = (aw / aw.sum((-1, -2), keepdims=True)).astype(np.float32)
    g = dict(value=value, loc=loc, aw=aw, grad_out=grad_out, shapes=pyr, starts=starts)
    a = (value.astype(np.float64), pyr, starts, loc.astype(np.float64), aw.astype(np.float64))
    ref_gv, ref_gl, ref_ga = O.backward(*a, grad_out.astype(np.float64))
    out, gv, gl, ga = lib.run("quad", "dest", BF16, g)
    np.testing.assert_allclose(gv, ref_gv, rtol=2.0 ** -7, atol=1e-3 * float(np.abs(ref_gv).max()))
    close32(ga, ref_ga)
    keep = ~kink_samples(g)
    close32(gl[keep], ref_gl[keep])


@pytest.fixture(scope="module")
def ablation_lib(emu_library):
    return EmuLib(emu_library("-DMSDA_ABLATION"))


def test_far_return_arm_hands_every_gradient_to_the_gated_k1(ablation_lib, monkeypatch):
    """Arm RLIPV2_CELL_FAR_RETURN (round 6; VERDICT r4 item 3e / r5 item 3c, ablation build): on a call with "far" samples the
    workgroups of cell_backward_kernel stop after their binning phase once the flag is up, and a K1 launch gated on the same word
    (quad_backward_shared_kernel<., ., REFDIM | 8>) writes every gradient of the locations / weights.  Bar: bit-equal to the route
    that runs K1 unconditionally (RLIPV2_MSDA_CELL=0: same K1 arithmetic, same sorting pass), whatever the interleaving of the
    workgroups; on a call WITHOUT far samples the arm changes nothing (the gated launch returns)."""
    lib = ablation_lib
    M = 1
    pyr, starts, S, ref2, value, grad_out, rng = _encoder_problem([(20, 54), (10, 27), (5, 14), (3, 7)], 1, M, seed=31)   # (4 cell columns: samples out of reach)
    loc = rng.random((1, S, M, 4, 4, 2)).astype(np.float32)                         # anywhere in the image: far samples
    aw = rng.random((1, S, M, 4, 4))
    aw = (aw / aw.sum((-1, -2), keepdims=True)).astype(np.float32)
    far = dict(value=value, loc=loc, aw=aw, grad_out=grad_out, shapes=pyr, starts=starts)
    for k in ("RLIPV2_MSDA_CELL", "RLIPV2_CELL_FAR_RETURN"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("RLIPV2_MSDA_CELL", "0")
    _, gv_k1, gl_k1, ga_k1 = lib.run("quad", "dest", BF16, far)
    monkeypatch.delenv("RLIPV2_MSDA_CELL")
    monkeypatch.setenv("RLIPV2_CELL_FAR_RETURN", "1")
    for _ in range(1):
        _, gv, gl, ga = lib.run("quad", "dest", BF16, far)
        # grad_value comes from the sorting pass in both routes.  On the lane-level model that pass is repeatable only to a
        # bfloat16 rounding: a wave's LDS atomics return their slots in the order the model's lane THREADS arrive (the hardware
        # serves the lanes of one instruction in lane order: bit-repeatable there, profiles/r03_nondeterminism.txt)
        assert np.abs(gv - gv_k1).max() <= 2.0 ** -7 * np.abs(gv_k1).max() and (gv != gv_k1).mean() < 1e-3
        assert np.array_equal(gl.astype(np.float32).view(np.uint32), gl_k1.astype(np.float32).view(np.uint32))
        assert np.array_equal(ga.astype(np.float32).view(np.uint32), ga_k1.astype(np.float32).view(np.uint32))
    ref_gv, ref_gl, ref_ga = O.backward(value.astype(np.float64), pyr, starts, loc.astype(np.float64), aw.astype(np.float64),
                                        grad_out.astype(np.float64))
    close32(ga, ref_ga)
    # arm RLIPV2_DEST_QUEUE on top: the sorting fallback's bin / combine launches as fixed grids striding over the items
    # (bin_queue_kernel, combine_queue_kernel: the item bodies are the product kernels' own text) -- the same grad_value
    monkeypatch.setenv("RLIPV2_DEST_QUEUE", "1")
    _, gv_q, gl_q, ga_q = lib.run("quad", "dest", BF16, far)
    monkeypatch.delenv("RLIPV2_DEST_QUEUE")
    assert np.abs(gv_q - gv_k1).max() <= 2.0 ** -7 * np.abs(gv_k1).max() and (gv_q != gv_k1).mean() < 1e-3
    np.testing.assert_allclose(gv_q, ref_gv, rtol=2.0 ** -7, atol=1e-3 * float(np.abs(ref_gv).max()))
    assert np.array_equal(ga_q.astype(np.float32).view(np.uint32), ga_k1.astype(np.float32).view(np.uint32))
    # no far sample: the product kernels' results, bit for bit
    pyr2, starts2, S2, value2, loc2, aw2 = make_problem([(20, 27), (10, 14), (5, 7), (3, 4)], 1, (1.5, 1.5, 1.0, 0.7), seed=11)
    go2 = bf16_val(bf16_bits(np.random.default_rng(5).standard_normal((1, S2, 32)))).astype(np.float32)
    near = dict(value=value2, loc=loc2, aw=aw2, grad_out=go2, shapes=pyr2, starts=starts2)
    monkeypatch.setenv("RLIPV2_DEST_QUEUE", "1")                                     # (both arms: three empty launches of <= 512 workgroups)
    with_arm = lib.run("quad", "dest", BF16, near)
    monkeypatch.delenv("RLIPV2_CELL_FAR_RETURN")
    monkeypatch.delenv("RLIPV2_DEST_QUEUE")
    without = lib.run("quad", "dest", BF16, near)
    for a, b in zip(with_arm[1:], without[1:]):                                     # (the patch pass: repeatable on the model too)
        assert np.array_equal(np.nan_to_num(a), np.nan_to_num(b))


def test_far_return_arm_on_the_fused_route(ablation_lib, monkeypatch):
    """The same arm through msda_fused_backward_ws (the train step's route: geometry backward as the kernels' epilogue, the gradient of
    the projection rows instead of grad_sampling_loc / grad_attn_weight).  Here the gate word travels to the gated K1 in the pointer
    argument that instantiation does not use (g_loc); wide offsets put samples out of their cells' reach.  Bar: the projection rows'
    gradient bit-equal to the route that runs K1 (with its epilogue) unconditionally."""
    Lb = ablation_lib.L
    M, L, P = 1, 4, 4
    pyr, starts, S, ref2, value, grad_out, rng = _encoder_problem([(20, 54), (10, 27), (5, 14), (3, 7)], 1, M, seed=33)
    N, Lq = 1, S
    qproj = rng.standard_normal((N, Lq, M * L * P * 3))
    qproj[..., :M * L * P * 2] *= 40.0                                               # offsets of tens of pixels: far samples
    qproj = bf16_val(bf16_bits(qproj)).astype(np.float32)
    ref = np.ascontiguousarray(np.broadcast_to(ref2[None, :, None, :], (N, Lq, L, 2)), dtype=np.float32)
    vb, qb, gob = bf16_bits(value), bf16_bits(qproj), bf16_bits(grad_out)
    dims = (N, S, M, 32, L, Lq, P)
    p = lambda a: a.ctypes.data                                                       # noqa: E731
    out = np.zeros((N, Lq, M * 32), dtype=np.uint16)
    loc = np.full((N, Lq, M, L, P, 2), np.nan, dtype=np.float32)
    aw = np.full((N, Lq, M, L, P), np.nan, dtype=np.float32)
    assert Lb.msda_fused_forward(BF16, p(vb), p(pyr), p(starts), p(qb), p(ref), 2, *dims, p(out), p(loc), p(aw), None) == 0
    ws_bytes = Lb.msda_backward_workspace_bytes(BF16, p(pyr), *dims)

    def backward():
        ws = np.zeros(ws_bytes + 64, dtype=np.uint8)
        gv, gq = np.zeros(vb.shape, dtype=np.uint16), np.full(qb.shape, 0x7fc0, dtype=np.uint16)
        assert Lb.msda_fused_backward_ws(FLAG_BF16_GV, BF16, p(vb), p(pyr), p(starts), p(pyr), p(loc), p(aw), p(ref), 2, p(gob),
                                         *dims, p(gv), p(gq), p(ws), ws_bytes, None) == 0
        return gv, gq
    for k in ("RLIPV2_MSDA_CELL", "RLIPV2_CELL_FAR_RETURN", "RLIPV2_DEST_QUEUE"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("RLIPV2_MSDA_CELL", "0")
    gv_k1, gq_k1 = backward()
    monkeypatch.delenv("RLIPV2_MSDA_CELL")
    gv_cell, gq_cell = backward()                                                    # the product route: cell kernel's values stand
    monkeypatch.setenv("RLIPV2_CELL_FAR_RETURN", "1")
    monkeypatch.setenv("RLIPV2_DEST_QUEUE", "1")
    gv_arm, gq_arm = backward()
    assert np.isfinite(bf16_val(gq_arm)).all()
    assert np.array_equal(gq_arm, gq_k1)                                             # every row rewritten by the gated K1
    assert np.array_equal(gq_cell, gq_k1) or np.abs(bf16_val(gq_cell) - bf16_val(gq_k1)).max() <= 2.0 ** -7 * np.abs(bf16_val(gq_k1)).max()
    # grad_value comes from the sorting pass either way: equal to a rounding on the model
    assert np.abs(bf16_val(gv_arm) - bf16_val(gv_k1)).max() <= 2.0 ** -7 * np.abs(bf16_val(gv_k1)).max()
    # ... and the samples WERE out of reach (so the gated K1 did run): the record-emitting forward raises the same flag on them
    vp, i, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
    Lb.msda_records_bytes.argtypes = [i, vp, *[i] * 7]
    Lb.msda_records_bytes.restype = sz
    Lb.msda_records_forward.argtypes = [i, vp, vp, vp, vp, vp, vp, i, vp, vp, *[i] * 7, vp, vp, sz, vp]
    rec_bytes = Lb.msda_records_bytes(BF16, p(pyr), *dims)
    records = np.zeros(rec_bytes, dtype=np.uint8)
    assert Lb.msda_records_forward(BF16, p(vb), p(pyr), p(starts), p(pyr), None, None, 0, p(loc), p(aw), *dims, p(out), p(records),
                                   rec_bytes, None) == 0
    assert int(records[:256].view(np.int32)[60]) != 0


def test_plain_b0_signature_without_host_shapes(lib):
    """msda_forward / msda_backward exactly as the reference's extension is bound (no host copy of the shapes,
    no workspace): "auto" then takes the direct-gather forward and K1 + the sorted scatter of round 1 (msda_window.hip: LDS
    counting sort, row_newbcast DPP operands, one float atomic per finished row) -- a small decoder-like call against the
    oracle; with FULL also the reference goldens, and the "window" pair (LDS-DMA staged tile forward) on the encoder one."""
    rng = np.random.default_rng(41)
    pyr = np.asarray([(12, 15), (6, 8), (3, 4), (2, 2)], dtype=np.int64)
    starts = np.concatenate(([0], np.cumsum(pyr[:, 0] * pyr[:, 1])[:-1])).astype(np.int64)
    S, N, M, Lq = int((pyr[:, 0] * pyr[:, 1]).sum()), 2, 2, 23
    ref = rng.uniform(-0.05, 1.05, size=(N, Lq, 2))
    off = rng.standard_normal((N, Lq, M, 4, 4, 2)) * 2.0
    loc = (ref[:, :, None, None, None, :] + off / np.stack([pyr[:, 1], pyr[:, 0]], -1)[None, None, None, :, None, :])
    aw = rng.random((N, Lq, M, 4, 4))
    g = dict(value=rng.standard_normal((N, S, M, 32)).astype(np.float32), loc=loc.astype(np.float32),
             aw=(aw / aw.sum((-1, -2), keepdims=True)).astype(np.float32),
             grad_out=rng.standard_normal((N, Lq, M * 32)).astype(np.float32), shapes=pyr, starts=starts)
    a = (g["value"].astype(np.float64), pyr, starts, g["loc"].astype(np.float64), g["aw"].astype(np.float64))
    ref_out = O.forward(*a)
    ref_gv, ref_gl, ref_ga = O.backward(*a, g["grad_out"].astype(np.float64))
    keep = ~kink_samples(g)
    out, gv, gl, ga = lib.run("auto", "auto", F32, g, host_shapes=False)
    close32(out, ref_out)
    close32(gv, ref_gv)
    close32(ga, ref_ga)
    close32(gl[keep], ref_gl[keep])
    for case in (["model_dec", "model_enc"] if FULL else []):
        g = load_golden(case)
        keep = ~kink_samples(g)
        for fwd, bwd in [("auto", "auto")] + ([("window", "window")] if case == "model_enc" else []):
            out, gv, gl, ga = lib.run(fwd, bwd, F32, g, host_shapes=False)
            close32(out, g["out_f32"])
            close32(gv, g["g_value_f32"])
            close32(ga, g["g_aw_f32"])
            close32(gl[keep], g["g_loc_f32"][keep])


@pytest.mark.parametrize("D", [256, 1])
@pytest.mark.parametrize("dt", [F32, BF16])
def test_one_head_of_many_channels_as_sample_then_project_calls_the_op(lib, D, dt):
    """deform_attn.sample_then_project (round 5) calls the op with ONE head of 256 channels (the unprojected memory) or of 1
    channel (the coverage of the value projection's bias) and the (query, head) pairs as queries: M' = 1, D' in {256, 1},
    Lq' = Lq x 8, with and without host shapes ("auto" must end on a route that takes these dimensions).  Against the oracle."""
    rng = np.random.default_rng(5 + D)
    pyr = np.asarray([(9, 12), (5, 6), (3, 3), (2, 2)], dtype=np.int64)
    starts = np.concatenate(([0], np.cumsum(pyr[:, 0] * pyr[:, 1])[:-1])).astype(np.int64)
    S, N, Lq = int((pyr[:, 0] * pyr[:, 1]).sum()), 2, 6 * 8
    ref = rng.uniform(-0.05, 1.05, size=(N, Lq, 2))
    off = rng.standard_normal((N, Lq, 1, 4, 4, 2)) * 2.0
    loc = ref[:, :, None, None, None, :] + off / np.stack([pyr[:, 1], pyr[:, 0]], -1)[None, None, None, :, None, :]
    aw = rng.random((N, Lq, 1, 4, 4))
    value = rng.standard_normal((N, S, 1, D)).astype(np.float32)
    go = rng.standard_normal((N, Lq, D)).astype(np.float32)
    if dt == BF16:
        value, go = bf16_val(bf16_bits(value)).astype(np.float32), bf16_val(bf16_bits(go)).astype(np.float32)
    g = dict(value=value, loc=loc.astype(np.float32), aw=(aw / aw.sum((-1, -2), keepdims=True)).astype(np.float32), grad_out=go,
             shapes=pyr, starts=starts)
    a = (g["value"].astype(np.float64), pyr, starts, g["loc"].astype(np.float64), g["aw"].astype(np.float64))
    ref_out = O.forward(*a)
    ref_gv, ref_gl, ref_ga = O.backward(*a, g["grad_out"].astype(np.float64))
    keep = ~kink_samples(g)
    for host_shapes in (True, False):
        out, gv, gl, ga = lib.run("auto", "auto", dt, g, host_shapes=host_shapes)
        tol = dict(rtol=2.0 ** -7, atol_rel=2.0 ** -7) if dt == BF16 else {}
        close32(out, ref_out, **tol)
        close32(gv, ref_gv, **tol)
        close32(ga, ref_ga)
        close32(gl[keep], ref_gl[keep])


@pytest.mark.parametrize("dt,pyr,Q", ([(F32, [(9, 12), (5, 6), (3, 3), (2, 2)], 24)] if FULL else []) + [(BF16, [(40, 52), (16, 33), (5, 7), (3, 3)], 8)])
def test_rows_backward_against_the_oracle(lib, dt, pyr, Q):
    """csrc/msda_rows.hip (round 5, never run on hardware): backward of the op with ONE head of 256 channels -- the memory's
    gradient by the ownership scatter (pixel ranges of 16 ... 128 pixels per workgroup: the second pyramid has levels with the 64- and the 32-pixel
    range, several ranges per level and partial last ranges), location / weight gradients by the dots kernel --
    against the oracle on the same operands; bit-repeatable; every row of grad_src written (no zero-fill by the caller)."""
    L = lib.L
    rng = np.random.default_rng(len(pyr) + Q + dt)
    pyr = np.asarray(pyr, dtype=np.int64)
    starts = np.concatenate(([0], np.cumsum(pyr[:, 0] * pyr[:, 1])[:-1])).astype(np.int64)
    S, N, C, P = int((pyr[:, 0] * pyr[:, 1]).sum()), 2, 256, 4
    ref = rng.uniform(-0.05, 1.05, size=(N, Q, 2))
    ref[:, : Q // 4] = ref[:, :1]                                        # a cluster of queries on one spot (long per-pixel sums)
    off = rng.standard_normal((N, Q, 1, 4, P, 2)) * 2.0
    loc = (ref[:, :, None, None, None, :] + off / np.stack([pyr[:, 1], pyr[:, 0]], -1)[None, None, None, :, None, :]).astype(np.float32)
    aw = rng.random((N, Q, 1, 4, P))
    aw = (aw / aw.sum((-1, -2), keepdims=True)).astype(np.float32)
    src = rng.standard_normal((N, S, 1, C)).astype(np.float32)
    dz = rng.standard_normal((N, Q, C)).astype(np.float32)
    if dt == BF16:
        src, dz = bf16_val(bf16_bits(src)).astype(np.float32), bf16_val(bf16_bits(dz)).astype(np.float32)
    a = (src.astype(np.float64), pyr, starts, loc.astype(np.float64), aw.astype(np.float64))
    ref_gv, ref_gl, ref_ga = O.backward(*a, dz.astype(np.float64))
    vp, i = ctypes.c_void_p, ctypes.c_int
    L.msda_rows_backward_supported.argtypes = [i, vp, i, i, i, i, i, i]
    L.msda_rows_backward.argtypes = [i, vp, vp, vp, vp, vp, vp, i, i, i, i, i, i, vp, vp, vp, vp]
    p = lambda x: x.ctypes.data                                                  # noqa: E731
    assert L.msda_rows_backward_supported(dt, p(pyr), N, S, C, 4, Q, P) == 1
    assert L.msda_rows_backward_supported(dt, p(pyr), N, S, 128, 4, Q, P) == 0 and L.msda_rows_backward_supported(dt, p(pyr), N, S + 1, C, 4, Q, P) == 0
    enc = (lambda x: np.ascontiguousarray(bf16_bits(x))) if dt == BF16 else (lambda x: np.ascontiguousarray(x, dtype=np.float32))
    src_d, dz_d = enc(src), enc(dz)
    runs = []
    for _ in range(2):
        g_src = np.full(src_d.shape, 0x7fc0 if dt == BF16 else np.nan, dtype=src_d.dtype)       # NaN: an unwritten row would show
        g_loc, g_aw = np.full(loc.shape, np.nan, dtype=np.float32), np.full(aw.shape, np.nan, dtype=np.float32)
        assert L.msda_rows_backward(dt, p(src_d), p(starts), p(pyr), p(loc), p(aw), p(dz_d), N, S, C, 4, Q, P, p(g_src), p(g_loc),
                                    p(g_aw), None) == 0
        runs.append((g_src.copy(), g_loc.copy(), g_aw.copy()))
    for x, y in zip(*runs):
        np.testing.assert_array_equal(x, y)                              # bit-repeatable
    g_src = bf16_val(runs[0][0]).astype(np.float64) if dt == BF16 else runs[0][0].astype(np.float64)
    assert np.isfinite(g_src).all()
    tol = dict(rtol=2.0 ** -7, atol_rel=2.0 ** -7) if dt == BF16 else {}
    close32(g_src, ref_gv, **tol)
    g = dict(loc=loc, shapes=pyr)
    keep = ~kink_samples(g)
    close32(runs[0][2].astype(np.float64), ref_ga)
    close32(runs[0][1].astype(np.float64)[keep], ref_gl[keep])
