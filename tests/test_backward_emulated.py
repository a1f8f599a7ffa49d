"""rlipv2_amd/csrc/msda_patch.hip ITSELF -- cell_backward_kernel, patch_dest_kernel and their launchers -- compiled for the
CPU against the lane-level workgroup model of tools/emu/ and run on a small encoder problem:

* the PRODUCT kernels (validated on an MI355X) must reproduce the oracle here too -- that calibrates the model (4x4x4 and
  16x16x32 MFMA operand layouts, transposing LDS reads, DPP scans and broadcasts, ballot / readlane, buffer loads);
* every EXPERIMENT arm of round 3 (written while the GPU pool was closed: geometry once per quad, swapped MFMA operands,
  level starts without the dependent vector load, window copies issued up front, the un-branched mask-word prefetch, several
  patches per wave; round 6: grad_out rows read from a cell-major copy) must reproduce the product kernels' three gradients BIT FOR BIT, B0 signature and fused geometry.

What this cannot see: the compiler's code generation, timing, anything the hardware does differently from the measured
semantics the model encodes.  The GPU runs of tools/r03_experiments.py remain the last word."""
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

PYR = [(20, 27), (10, 14), (5, 7), (3, 4)]
M = 1


@pytest.fixture(scope="module")
def emulator(tmp_path_factory):
    d = tmp_path_factory.mktemp("emu_bwd")
    exe = str(d / "backward_emu")
    emu = os.path.join(ROOT, "tools", "emu")
    # EMU_TSAN=1: ThreadSanitizer build (a lane pair that no rendezvous / barrier orders = a data race; tools/emu/README.md)
    san = ["-g", "-fsanitize=thread"] if os.environ.get("EMU_TSAN") == "1" else []
    subprocess.run([CLANG, "-x", "c++", "-std=c++20", "-O1", "-pthread", *san, "-I" + os.path.join(emu, "stub"),
                    "-I" + os.path.join(ROOT, "include"), "-Wno-unknown-pragmas", os.path.join(emu, "backward_emu.cpp"),
                    "-o", exe], check=True, capture_output=True, timeout=600)
    pyr, starts, S, value, loc, aw = make_problem(PYR, M, (1.5, 1.5, 1.0, 0.7), seed=11)
    go = bf16_val(bf16_bits(np.random.default_rng(5).standard_normal((1, S, M * 32)))).astype(np.float32)
    prob = str(d / "problem.bin")
    with open(prob, "wb") as f:
        f.write(np.asarray([1, S, M, S] + [int(v) for hw in pyr for v in hw], dtype=np.int32).tobytes())
        f.write(bf16_bits(value).tobytes())
        f.write(starts.astype(np.int64).tobytes())
        f.write(loc.astype(np.float32).tobytes())
        f.write(aw.astype(np.float32).tobytes())
        f.write(bf16_bits(go).tobytes())
    return dict(exe=exe, prob=prob, dir=d, pyr=pyr, starts=starts, S=S, value=value, loc=loc, aw=aw, go=go, cache={})


def run(emu, env, fused=False):
    key = (tuple(sorted(env.items())), fused)
    if key in emu["cache"]:
        return emu["cache"][key]
    e = {k: v for k, v in os.environ.items() if not k.startswith("RLIPV2_")}
    e.update(env)
    e["EMU_FUSED"] = "1" if fused else "0"
    out = str(emu["dir"] / "out.bin")
    subprocess.run([emu["exe"], emu["prob"], out], check=True, env=e, timeout=900)
    raw = np.fromfile(out, dtype=np.uint8)
    S = emu["S"]
    n_gv = S * M * 32
    res = {"g_value": raw[:n_gv * 2].view(np.uint16).copy()}
    o = n_gv * 2
    if fused:
        res["g_qproj"] = raw[o:o + S * M * 48 * 2].view(np.uint16).copy()
        o += S * M * 48 * 2
    else:
        res["g_loc"] = raw[o:o + S * M * 32 * 4].view(np.uint32).copy()
        o += S * M * 32 * 4
        res["g_aw"] = raw[o:o + S * M * 16 * 4].view(np.uint32).copy()
        o += S * M * 16 * 4
    res["far"] = int(raw[o:o + 4].view(np.int32)[0])
    emu["cache"][key] = res
    return res


def test_product_kernels_on_the_model_reproduce_the_oracle(emulator):
    e = emulator
    S = e["S"]
    ref_gv, ref_gl, ref_ga = O.backward(e["value"].astype(np.float64), e["pyr"], e["starts"], e["loc"].astype(np.float64),
                                        e["aw"].astype(np.float64), e["go"].astype(np.float64))
    got = run(e, {})
    assert got["far"] == 0
    gv = bf16_val(got["g_value"]).reshape(ref_gv.shape)
    gl = got["g_loc"].view(np.float32).reshape(ref_gl.shape)
    ga = got["g_aw"].view(np.float32).reshape(ref_ga.shape)
    assert np.isfinite(gv).all() and np.isfinite(ga).all()
    np.testing.assert_allclose(gv, ref_gv, rtol=2.0 ** -7, atol=2e-3 * float(np.abs(ref_gv).max()))       # bfloat16 result
    np.testing.assert_allclose(ga, ref_ga, rtol=1e-4, atol=1e-5 * float(np.abs(ref_ga).max()))
    keep = np.isfinite(ref_gl) & np.isfinite(gl)                     # (the NaN location of the problem: skipped sample)
    np.testing.assert_allclose(gl[keep], ref_gl[keep], rtol=1e-4, atol=1e-5 * float(np.abs(ref_gl[keep]).max()))
    # fused geometry (the route of the train step): softmax backward + offset scaling of the reference module
    fz = run(e, {}, fused=True)
    assert np.array_equal(fz["g_value"], got["g_value"])
    gq = bf16_val(fz["g_qproj"]).reshape(S, M * 48)
    aw = e["aw"][0].astype(np.float64)                               # [S, M, 4, 4]
    gaw = np.nan_to_num(ref_ga[0])
    dot = (aw * gaw).sum((-1, -2), keepdims=True)
    g_logit = (aw * (gaw - dot)).reshape(S, M * 16)
    scale = 1.0 / np.stack([e["pyr"][:, 1], e["pyr"][:, 0]], -1).astype(np.float64)          # (1 / W, 1 / H) per level
    g_off = (np.nan_to_num(ref_gl[0]) * scale[None, None, :, None, :]).reshape(S, M * 32)
    np.testing.assert_allclose(gq[:, :M * 32], g_off, rtol=2.0 ** -7, atol=2e-3 * float(np.abs(g_off).max()))
    np.testing.assert_allclose(gq[:, M * 32:], g_logit, rtol=2.0 ** -7, atol=2e-3 * float(np.abs(g_logit).max()))


ARMS = [{"RLIPV2_CELL_SHARED": "3"}, {"RLIPV2_CELL_SHARED": "4"}, {"RLIPV2_PATCH_MULTI": "1"},
        {"RLIPV2_CELL_SHARED": "2", "RLIPV2_PATCH_REPS": "3"},
        {"RLIPV2_PATCH_CELLG": "1"}, {"RLIPV2_PATCH_CELLG": "1", "RLIPV2_PATCH_REPS": "3", "RLIPV2_CELL_SHARED": "3"}]     # round 6
if os.environ.get("RLIPV2_TEST_EMU_FULL", "0") == "1":       # (the default set covers every code path of the arms once)
    ARMS += [{"RLIPV2_CELL_SHARED": "1"}, {"RLIPV2_CELL_SHARED": "2"}, {"RLIPV2_PATCH_REPS": "2"}, {"RLIPV2_PATCH_REPS": "4"},
             {"RLIPV2_PATCH_REPS": "8"}, {"RLIPV2_CELL_SHARED": "3", "RLIPV2_PATCH_REPS": "4"}]


@pytest.mark.parametrize("arm", ARMS, ids=lambda a: ",".join(f"{k[7:]}={v}" for k, v in a.items()))
@pytest.mark.parametrize("fused", [False, True], ids=["b0", "fused"])
def test_experiment_arms_reproduce_the_product_kernels_bit_for_bit(emulator, arm, fused):
    base = run(emulator, {}, fused)
    got = run(emulator, arm, fused)
    assert got["far"] == 0
    for k in base:
        if k != "far":
            assert np.array_equal(got[k], base[k]), f"{k} differs from the product kernels' under {arm}"
