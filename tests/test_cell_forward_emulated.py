"""The DEVICE SOURCE of cell_forward_kernel (rlipv2_amd/csrc/msda_cell_forward.inc -- the very file hipcc compiles into the
library) executed on the CPU against the lane-level model of a gfx950 workgroup (tools/emu/: a host thread per lane, LDS as
a byte array, DPP / readfirstlane / transposing LDS read / 4x4x4 MFMA as rendezvous with the semantics measured on the
hardware) through the library's own C ABI (msda_forward_hs, variant "cell"), checked against the oracle.  This pins the kernel's logic -- indexing, window staging with its
zero border, records, operand placement, the plain-load route of a level that does not fit, the store pattern -- without a
GPU; what it cannot pin is the compiler's code generation and the hardware itself (tests/test_msda_cell_forward_gpu.py)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import msda_oracle as O  # noqa: E402  (tests may use the oracle)

CLANG = "/opt/rocm/lib/llvm/bin/clang++"
pytestmark = pytest.mark.skipif(not os.path.exists(CLANG), reason="needs the ROCm clang++ (ext_vector_type) as host compiler")


@pytest.fixture(scope="module")
def emulator(tmp_path_factory):
    """the MSDA library built for the workgroup model (tools/emu/build_lib.sh); the kernel is reached through its C ABI"""
    import ctypes
    so = str(tmp_path_factory.mktemp("emu") / "libmsda_emu.so")
    subprocess.run([os.path.join(ROOT, "tools", "emu", "build_lib.sh"), so], check=True, capture_output=True, timeout=900)
    L = ctypes.CDLL(so)
    vp, i = ctypes.c_void_p, ctypes.c_int
    L.msda_forward_hs.argtypes = [i, i, vp, vp, vp, vp, vp, vp, *([i] * 7), vp, vp]
    return L


def bf16_bits(x):
    u = np.asarray(x, dtype=np.float32).view(np.uint32)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def bf16_val(bits):
    return (np.asarray(bits, dtype=np.uint32) << 16).view(np.float32)


def make_problem(pyr, M, spread, seed):
    rng = np.random.default_rng(seed)
    pyr = np.asarray(pyr, dtype=np.int64)
    starts = np.concatenate(([0], np.cumsum(pyr[:, 0] * pyr[:, 1])[:-1])).astype(np.int64)
    S = int((pyr[:, 0] * pyr[:, 1]).sum())
    ref = []
    for H, W in pyr:
        ys, xs = np.meshgrid((np.arange(H) + 0.5) / H, (np.arange(W) + 0.5) / W, indexing="ij")
        ref.append(np.stack([xs.ravel(), ys.ravel()], -1))
    ref = np.concatenate(ref, 0)
    off = rng.normal(0.0, 1.0, (1, S, M, 4, 4, 2)) * np.asarray(spread, dtype=np.float64).reshape(1, 1, 1, 4, 1, 1)
    loc = ref[None, :, None, None, None, :] + off / np.stack([pyr[:, 1], pyr[:, 0]], -1)[None, None, None, :, None, :]
    loc[0, 0, 0, :, 0] = (-0.7, 0.5)                  # out of range of every level
    loc[0, 1, 0, :, 1] = (0.0, 0.0)                   # the level's corner: three corners outside
    loc[0, 2, 0, :, 2] = (1.0, 1.0)
    loc[0, 3, 0, :, 3] = (np.nan, 0.3)                # NaN location: skipped (.cuh:285)
    aw = rng.random((1, S, M, 4, 4))
    aw /= aw.sum((-1, -2), keepdims=True)
    value = bf16_val(bf16_bits(rng.standard_normal((1, S, M, 32))))
    return pyr, starts, S, value.astype(np.float32), loc.astype(np.float32), aw.astype(np.float32)


def run_emulator(L, tmp_path, pyr, starts, S, M, value, loc, aw):
    vb = np.ascontiguousarray(bf16_bits(value))
    sh, st = np.ascontiguousarray(pyr, dtype=np.int64), np.ascontiguousarray(starts, dtype=np.int64)
    loc, aw = np.ascontiguousarray(loc, dtype=np.float32), np.ascontiguousarray(aw, dtype=np.float32)
    out = np.full((1, S, M * 32), 0x7FC0, dtype=np.uint16)                 # NaN: an unwritten query shows
    p = lambda a: a.ctypes.data                                            # noqa: E731
    rc = L.msda_forward_hs(6, 2, p(vb), p(sh), p(st), p(sh), p(loc), p(aw), 1, S, M, 32, 4, S, 4, p(out), None)   # "cell", bf16
    assert rc == 0, rc
    return bf16_val(out).reshape(1, S, M * 32)


@pytest.mark.parametrize("name,pyr,M,spread", [
    ("windows of every level in LDS, 2 x 2 cells", [(20, 27), (10, 14), (5, 7), (3, 4)], 2, (1.5, 1.5, 1.0, 0.7)),
    ("level 0 too wide for the window budget: plain-load route", [(30, 40), (15, 20), (8, 10), (4, 5)], 1, (25.0, 2.0, 1.0, 0.7)),
    ("ragged pyramid (sizes not halving exactly), wide offsets on the coarse levels", [(25, 34), (13, 17), (7, 9), (4, 5)], 1,
     (2.0, 3.0, 3.0, 3.0)),
])
def test_device_source_on_the_lane_level_model(emulator, tmp_path, name, pyr, M, spread):
    pyr, starts, S, value, loc, aw = make_problem(pyr, M, spread, seed=11)
    got = run_emulator(emulator, tmp_path, pyr, starts, S, M, value, loc, aw)
    ref = O.forward(value.astype(np.float64), pyr, starts, loc.astype(np.float64), aw.astype(np.float64))
    assert np.isfinite(got).all(), "a query was not written, or garbage LDS reached a result"
    # bfloat16 output of a float32 sum with weights to 2^-16: one bfloat16 rounding
    np.testing.assert_allclose(got, ref, rtol=2.0 ** -7, atol=1e-3 * float(np.abs(ref).max()))
    # and tight against the float64 result BEFORE the final rounding is not available from the kernel; the mean error shows
    # that nothing systematic (a dropped lo half, a missing sample) hides under the rounding
    assert float(np.abs(got - ref).mean()) < 2e-3 * float(np.abs(ref).mean() + 1e-30)
