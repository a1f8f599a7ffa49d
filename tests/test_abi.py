"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol the
header declares; host-side argument checks reproduce the reference's error behaviour.
No compute is launched here (there is no GPU in the build container)."""
import ctypes
import os
import re

import pytest
import torch

from rlipv2_amd import _lib, msda

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions(cpu=False):
    """functions declared by include/*.h: the HIP library's headers, or (cpu=True) the CPU twins' header"""
    names = []
    for header in sorted(os.listdir(os.path.join(ROOT, "include"))):
        if (header == "rlipv2_msda_cpu.h") != cpu:
            continue
        text = open(os.path.join(ROOT, "include", header)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names += re.findall(r"\b((?:msda|linear|add_layernorm|layernorm_wide|adamw|alif_attention|window_attention|add_relu|affine_relu|groupnorm_tokens|dab|hoi_assign)_[a-z0-9_]+|add_relu_bf16)\s*\(", text)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    declared = _declared_functions()
    assert set(declared) == set(_lib.EXPORTS), (declared, _lib.EXPORTS)
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/*.h but not exported"


def test_cpu_library_exports_every_declared_symbol():
    L = _lib.cpu_lib()
    declared = _declared_functions(cpu=True)
    assert set(declared) == set(_lib.CPU_EXPORTS), (declared, _lib.CPU_EXPORTS)
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/rlipv2_msda_cpu.h but not exported"
    assert L.msda_cpu_abi_version() == 1 and L.msda_cpu_strerror(0) == b"ok"
    # host-side argument checks: bad dtype / dimensions / null operands are refused before anything is touched
    assert L.msda_forward_cpu(7, None, None, None, None, None, 1, 1, 1, 1, 1, 1, 1, None) != 0
    shapes = (ctypes.c_int64 * 2)(2, 2)
    starts = (ctypes.c_int64 * 1)(0)
    assert L.msda_forward_cpu(0, None, shapes, starts, None, None, 1, 4, 1, 1, 1, 1, 1, None) == -2
    assert L.msda_forward_cpu(0, None, shapes, starts, None, None, 1, 4, 0, 1, 1, 1, 1, None) == -3
    assert L.msda_forward_cpu(0, None, shapes, starts, None, None, 0, 4, 1, 1, 1, 1, 1, None) == 0          # empty batch
    assert L.msda_forward_cpu(0, None, shapes, starts, None, None, 0, 3, 1, 1, 1, 1, 1, None) == -4         # level outside value


def test_abi_version_and_strerror():
    L = _lib.lib()
    assert L.msda_abi_version() == 1
    assert _lib.strerror(0) == "ok"
    for st in range(-7, 0):
        assert _lib.strerror(st) not in ("ok", "unknown status")
    assert _lib.strerror(-99) == "unknown status"
    assert L.msda_variant_name(2) == b"quad"


def test_im2col_step_rule_matches_reference():
    # reference: ms_deform_attn_cuda.cu:50-52 -- batch % min(batch, im2col_step) == 0
    L = _lib.lib()
    for batch in range(1, 70):
        for step in (1, 2, 3, 4, 8, 64):
            want_ok = batch % min(batch, step) == 0
            assert (L.msda_check_im2col_step(batch, step) == 0) == want_ok, (batch, step)


def test_algorithmic_bytes_match_baseline_table():
    # 800x1333 pyramid, M8 D32 L4 P4, per image
    dims = (1, 22223, 8, 32, 4, 22223, 4)
    assert _lib.algorithmic_bytes(_lib.MSDA_F32, False, *dims) == 79647232
    assert _lib.algorithmic_bytes(_lib.MSDA_BF16, False, *dims) == 56890880
    assert _lib.algorithmic_bytes(_lib.MSDA_F32, True, *dims) == 136538112
    dec = (1, 22223, 8, 32, 4, 300, 4)
    assert round(_lib.algorithmic_bytes(_lib.MSDA_F32, False, *dec) / 1e6, 2) == 23.52


def test_variant_choice_is_static():
    L = _lib.lib()
    enc = (4, 22223, 8, 32, 4, 22223, 4)
    assert L.msda_pick_variant(0, _lib.MSDA_F64, *enc) == _lib.VARIANT_GENERIC
    assert L.msda_pick_variant(0, _lib.MSDA_F32, 1, 30, 2, 2, 2, 2, 2) == _lib.VARIANT_GENERIC
    assert L.msda_pick_variant(0, _lib.MSDA_F32, *enc) in (_lib.VARIANT_QUAD, _lib.VARIANT_WINDOW)


def test_bad_arguments_are_reported_not_launched():
    L = _lib.lib()
    # bad dtype / negative dims are rejected before anything touches the device
    assert L.msda_forward(7, None, None, None, None, None, 1, 1, 1, 1, 1, 1, 1, None, None) == -1
    assert L.msda_forward(0, None, None, None, None, None, -1, 1, 1, 1, 1, 1, 1, None, None) == -2
    # empty output is a no-op success (empty batch / no queries)
    assert L.msda_forward(0, None, None, None, None, None, 0, 10, 8, 32, 4, 5, 4, None, None) == 0
    assert L.msda_forward(0, None, None, None, None, None, 2, 10, 8, 32, 4, 0, 4, None, None) == 0


def _cpu_inputs():
    shapes = torch.tensor([[6, 4], [3, 2]], dtype=torch.long)
    starts = torch.tensor([0, 24], dtype=torch.long)
    value = torch.rand(1, 30, 2, 2)
    loc = torch.rand(1, 2, 2, 2, 2, 2)
    aw = torch.rand(1, 2, 2, 2, 2)
    return value, shapes, starts, loc, aw


def test_cpu_tensors_take_the_cpu_twins():
    """Reference: CPU tensors raise "Not implemented on the CPU" (models/ops/src/ms_deform_attn.h:54) and its models use
    ms_deform_attn_core_pytorch instead; SURVEY.md 8b asks the drop-in to accept CPU tensors.  Here they are served by the CPU
    twins (include/rlipv2_msda_cpu.h); values against the goldens: tests/test_msda_cpu.py.  Mixed devices still raise."""
    value, shapes, starts, loc, aw = _cpu_inputs()
    out = msda.ms_deform_attn_forward(value, shapes, starts, loc, aw, 64)
    assert out.shape == (1, 2, 4) and not out.is_cuda
    assert msda.MSDeformAttnFunction.apply(value, shapes, starts, loc, aw, 64).shape == (1, 2, 4)
    gv, gl, ga = msda.ms_deform_attn_backward(value, shapes, starts, loc, aw, torch.rand(1, 2, 4), 64)
    assert gv.shape == value.shape and gl.shape == loc.shape and ga.shape == aw.shape


def test_cuda_tensors_never_reach_the_cpu_twins(monkeypatch):
    """the CPU library serves CPU tensors only: with the HIP library missing a CUDA call raises (no fallback), it does not
    quietly compute on the host"""
    src = open(os.path.join(ROOT, "rlipv2_amd", "msda.py")).read()
    assert src.count("_cpu_call(") == 3                      # the definition + one dispatch per entry point ...
    assert src.count("if not value.is_cuda:\n") >= 2          # ... each behind the device test
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/librlipv2_msda.so")
    with pytest.raises(RuntimeError, match="no CPU / PyTorch fallback"):
        _lib.lib()


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/librlipv2_msda.so")
    with pytest.raises(RuntimeError, match="no CPU / PyTorch fallback"):
        _lib.lib()


def test_product_package_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "rlipv2_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.replace("no oracle", ""), f"{f} mentions the oracle"


def test_linear_wgrad_plan_and_cpu_behaviour():
    from rlipv2_amd import linear
    L = _lib.lib()
    # supported: out/in features multiples of 128; the workspace holds one float32 partial per token chunk
    assert L.linear_wgrad_supported(88892, 256, 256) == 1
    assert L.linear_wgrad_supported(88892, 1024, 256) == 1
    assert L.linear_wgrad_supported(88892, 384, 256) == 1
    assert L.linear_wgrad_supported(88892, 100, 256) == 0
    assert L.linear_wgrad_supported(0, 256, 256) == 0
    assert L.linear_wgrad_workspace_bytes(88892, 100, 256) == 0
    nbytes = L.linear_wgrad_workspace_bytes(88892, 256, 256)
    # per chunk: one 256x256 partial of dW + the bias-gradient partials of the 2 workgroups of a tile row (K / 128 = 2)
    assert nbytes % ((256 * 256 + 2 * 256) * 4) == 0 and 0 < nbytes <= 64 << 20
    assert L.linear_wgrad_workspace_bytes(1, 128, 128) == (128 * 128 + 128) * 4
    # CPU tensors never reach the kernel: token_linear is the library call there, linear_wgrad refuses
    x = torch.randn(3, 5, 128, requires_grad=True)
    w = torch.randn(128, 128, requires_grad=True)
    assert not linear.supported(x, w)
    torch.testing.assert_close(linear.token_linear(x, w, None), torch.nn.functional.linear(x, w))
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        linear.linear_wgrad(torch.zeros(4, 128), torch.zeros(4, 128))


def test_compat_shim_registers_reference_import_paths():
    """The names unmodified reference code imports (ms_deform_attn_func.py:22, ParSetransformer.py:27,
    dab_deformable/deformable_transformer.py:28) resolve to this package after compat.install()."""
    import importlib
    import sys
    import rlipv2_amd.compat as compat
    from rlipv2_amd import deform_attn
    saved = {k: v for k, v in sys.modules.items() if k == "MultiScaleDeformableAttention" or k.startswith("models")}
    try:
        for k in saved:
            del sys.modules[k]
        compat.install()
        MSDA = importlib.import_module("MultiScaleDeformableAttention")
        assert MSDA.ms_deform_attn_forward is msda.ms_deform_attn_forward
        assert MSDA.ms_deform_attn_backward is msda.ms_deform_attn_backward
        for root in ("models.ops", "models.dab_deformable.ops"):
            f = importlib.import_module(root + ".functions")
            assert f.MSDeformAttnFunction is msda.MSDeformAttnFunction
            m = importlib.import_module(root + ".modules")
            assert m.MSDeformAttn is deform_attn.MSDeformAttn
        assert MSDA.ms_deform_attn_forward(*_cpu_inputs(), 64).shape == (1, 2, 4)     # CPU tensors: the CPU twins
    finally:
        for k in [k for k in sys.modules if k == "MultiScaleDeformableAttention" or k.startswith("models")]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_round2_entry_points_validate_their_arguments_on_the_host():
    """groupnorm / decoder-glue / matcher entry points: shape and pointer checks run before anything touches a device."""
    import ctypes
    from rlipv2_amd import _lib
    L = _lib.lib()
    assert L.groupnorm_tokens_supported(256, 32, 4) == 1 and L.groupnorm_tokens_supported(256, 32, 1) == 1
    assert L.groupnorm_tokens_supported(128, 32, 4) == 0 and L.groupnorm_tokens_supported(256, 16, 4) == 0
    assert L.groupnorm_tokens_supported(256, 32, 5) == 0 and L.groupnorm_tokens_supported(256, 32, 0) == 0
    hw = (ctypes.c_int * 4)(16700, 4200, 1050, 273)
    chunks = sum((n + 255) // 256 for n in hw)                       # 256 tokens per workgroup
    want = (4 * chunks * 256 * 2 + 4 * 4 * 32 * 2 + 4 * 4 * 256 * 2) * 4
    assert L.groupnorm_tokens_workspace_bytes(4, hw, 4) == want
    assert L.groupnorm_tokens_workspace_bytes(0, hw, 4) == 0 and L.groupnorm_tokens_workspace_bytes(4, hw, 5) == 0
    bad = (ctypes.c_int * 2)(10, 0)
    assert L.groupnorm_tokens_workspace_bytes(2, bad, 2) == 0
    assert L.groupnorm_tokens_forward_bf16(None, hw, 4, 4, None, None, 1e-5, None, None, None, None, 0, None) != 0
    # decoder glue: empty problems are a no-op, negative sizes and missing pointers are refused
    assert L.dab_refine_boxes(None, 1, None, None, 0, 1e-5, None) == 0
    assert L.dab_refine_boxes(None, 1, None, None, -1, 1e-5, None) != 0
    assert L.dab_refine_boxes(None, 1, None, None, 8, 1e-5, None) != 0
    assert L.dab_reference_embed(None, None, None, None, 0, 5, 4, 1, None, None, 1, None) == 0
    assert L.dab_reference_embed(None, None, None, None, 2, 5, 9, 1, None, None, 1, None) != 0
    assert L.dab_reference_embed(None, None, None, None, 2, 5, 4, 1, None, None, 1, None) != 0
    # matcher: argument errors are -2
    sizes = (ctypes.c_int * 1)(3)
    assert L.hoi_assign_batch(None, 1, 1, 4, sizes, None, None, 3) == -2


def test_records_entry_points_validate_their_arguments_on_the_host():
    """msda_records_bytes / msda_records_forward / msda_records_backward (round 5): which calls the route takes, and that every
    argument error is reported before anything touches a device (real library, no GPU here)"""
    import ctypes
    import numpy as np
    from rlipv2_amd import _lib
    L = _lib.lib()
    pyr = [(100, 167), (50, 84), (25, 42), (13, 21)]
    S = sum(h * w for h, w in pyr)
    hs = (ctypes.c_int64 * 8)(*[v for hw in pyr for v in hw])
    dims = (4, S, 8, 32, 4, S, 4)
    need = L.msda_records_bytes(_lib.MSDA_BF16, hs, *dims)
    cells, items = 7 * 11, 4 * 8 * 7 * 11
    assert need > 256 + items * 128 + items * 340 * 256 and need % 16 == 0            # control block + window tables + sample records + masks + group records
    assert cells * 340 >= S                                                             # (every query has a slot in its cell)
    assert L.msda_records_bytes(_lib.MSDA_F32, hs, *dims) == 0                          # bfloat16 only
    assert L.msda_records_bytes(_lib.MSDA_BF16, hs, 4, S, 8, 32, 4, 300, 4) == 0        # encoder calls only (Lq == S)
    assert L.msda_records_bytes(_lib.MSDA_BF16, None, *dims) == 0                       # needs the host copy of the shapes
    ws = L.msda_backward_workspace_bytes(_lib.MSDA_BF16, hs, *dims)
    assert ws >= 4 * S * 8 * 16 * 12                                                    # room for rebuilt locations / weights (far fallback)
    a = np.zeros(64, dtype=np.uint8).ctypes.data                                        # any aligned non-null address: nothing is dereferenced
    ok_ptrs = dict(value=a, shapes=a, starts=a, out=a, records=a)
    fwd = lambda **kw: L.msda_records_forward(kw.get("dtype", _lib.MSDA_BF16), kw.get("value", a), a, a, hs, kw.get("qproj"), kw.get("ref"),   # noqa: E731
                                              kw.get("refdim", 0), kw.get("loc", a), kw.get("aw", a), *dims, kw.get("out", a),
                                              kw.get("records", a), kw.get("nbytes", need), None)
    assert fwd(refdim=3) != 0 and fwd(dtype=_lib.MSDA_F32) != 0                          # unsupported reference dimension / dtype
    assert fwd(out=None) != 0 and fwd(records=None) != 0 and fwd(loc=None) != 0          # op signature needs its locations
    assert fwd(refdim=2, qproj=None, ref=a) != 0 and fwd(refdim=2, qproj=a, ref=a, loc=a, aw=None) != 0   # both or neither
    assert fwd(nbytes=need - 1) != 0                                                      # records buffer too small
    assert fwd(value=a + 4) != 0                                                          # misaligned
    bwd = lambda **kw: L.msda_records_backward(kw.get("flags", _lib.FLAG_GRAD_VALUE_BF16), _lib.MSDA_BF16, a, a, a, hs, kw.get("loc", a),   # noqa: E731
                                               kw.get("aw", a), kw.get("ref"), kw.get("refdim", 0), a, *dims, kw.get("gv", a), kw.get("gl", a),
                                               kw.get("ga", a), kw.get("gq"), kw.get("records", a), kw.get("nbytes", need), kw.get("ws", a),
                                               kw.get("wsb", ws), None)
    assert bwd(gv=None) != 0 and bwd(records=None) != 0 and bwd(ws=None) != 0 and bwd(gl=None) != 0
    assert bwd(refdim=2, ref=a, gq=None) != 0 and bwd(refdim=2, ref=None, gq=a) != 0     # module operands need ref and grad_qproj
    assert bwd(loc=None) != 0                                                              # op signature needs its locations
    assert bwd(nbytes=need - 1) != 0 and bwd(wsb=ws - 1) != 0 and bwd(refdim=1) != 0
    del ok_ptrs
