"""The two Swin routes of round 5 -- fused window attention (csrc/window_attention.hip) and residual add + LayerNorm at the Swin
widths (csrc/layernorm_wide.hip) -- AS THE MODULES CALL THEM, on the lane-level model of tools/emu/: the product's Python code
(table preparation, packed qkv views, compact shift masks, autograd functions, the restructured block loop) runs unchanged, only
the library behind `_lib.lib()` is the host-model build of the same kernel sources (host pointers).  Against the same modules on
their PyTorch op sequence, in bfloat16.  Neither kernel has run on hardware; tests/test_zz_round5_gpu.py repeats this on the GPU."""
import ctypes
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_cell_forward_emulated import CLANG  # noqa: E402

from rlipv2_amd import _lib, norm, swin  # noqa: E402

pytestmark = pytest.mark.skipif(not os.path.exists(CLANG), reason="needs the ROCm clang++ as host compiler")


@pytest.fixture(scope="module")
def emu_lib(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("emu_swin") / "libdense_emu.so")
    subprocess.run([os.path.join(ROOT, "tools", "emu", "build_dense_lib.sh"), so], check=True, capture_output=True, timeout=900)
    L = ctypes.CDLL(so)
    vp, i, lg, f32 = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float
    L.window_attention_supported.argtypes = [i, i, i, i]
    L.window_attention_forward_bf16.argtypes = [vp, vp, vp, vp, i, i, i, i, f32, vp, vp]
    L.window_attention_backward_bf16.argtypes = [vp, vp, vp, vp, vp, i, i, i, i, f32, vp, vp]
    L.window_attention_rows_forward_bf16.argtypes = [vp, vp, vp, i, i, vp, vp, vp, i, i, i, i, f32, vp, vp]
    L.window_attention_rows_backward_bf16.argtypes = [vp, vp, vp, i, i, vp, vp, vp, vp, i, i, i, i, f32, vp, vp, vp]
    L.layernorm_wide_supported.argtypes = [lg, i]
    L.layernorm_wide_forward_bf16.argtypes = [vp, vp, vp, vp, lg, i, f32, vp, vp, vp, vp, vp]
    L.layernorm_wide_backward_bf16.argtypes = [vp, vp, vp, vp, vp, vp, lg, i, vp, vp]
    return L


@pytest.fixture
def on_model(emu_lib, monkeypatch):
    """CPU tensors are "on the device", the library is the host-model build"""
    monkeypatch.setattr(_lib, "lib", lambda: emu_lib)
    monkeypatch.setattr(norm, "_on_device", lambda t: True)
    yield
    norm.fused_wide_layer_norm = False
    swin.fused_window_attention = False


def _grads(module, run):
    for p in module.parameters():
        p.grad = None
    out, x = run()
    return [out.detach().float(), x.grad.float()] + [p.grad.float() for p in module.parameters() if p.grad is not None]


@pytest.mark.parametrize("ws,heads,shift", [(7, 3, True), (7, 6, False), (4, 3, True)])
def test_window_attention_module(on_model, ws, heads, shift):
    torch.manual_seed(ws * heads)
    C, B, Hp, Wp = heads * 32, 2, 2 * ws, 3 * ws
    attn = swin.WindowAttention(C, ws, heads).to(torch.bfloat16)
    attn.relative_position_bias_table.requires_grad_(False)        # (frozen, as in the Swin backbones)
    with torch.no_grad():
        attn.relative_position_bias_table.add_((0.5 * torch.randn(attn.relative_position_bias_table.shape)).to(torch.bfloat16))
    mask = None
    if shift:
        mask = swin.shift_mask(Hp, Wp, ws, ws // 2, "cpu")
        mask.compact = swin.compact_masks(mask)
        assert mask.compact[0] is not None and int((mask.compact[1] >= 0).sum()) >= 2
    nW = (Hp // ws) * (Wp // ws)
    x0 = torch.randn(B, nW, ws * ws, C).to(torch.bfloat16)
    gy = torch.randn(B, nW, ws * ws, C).to(torch.bfloat16)

    def run():
        x = x0.clone().requires_grad_(True)
        y = attn(x, mask)
        y.backward(gy)
        return y, x
    res = {}
    for fused in (True, False):
        swin.fused_window_attention = fused
        res[fused] = _grads(attn, run)
    swin.fused_window_attention = False
    scale = float(res[False][0].abs().max())
    assert float((res[True][0] - res[False][0]).abs().max()) <= 2.0 ** -6 * scale
    for a, b in zip(res[True][1:], res[False][1:]):
        assert float((a - b).norm()) <= 3e-2 * float(b.norm()), (float((a - b).norm()), float(b.norm()))


def test_swin_stage_with_both_routes(on_model):
    """a BasicLayer (two blocks, the second shifted, padded windows, output norm, patch merging) with fused attention + fused
    residual norms against the same layer on PyTorch ops: outputs and the gradients of the input and of every trainable parameter"""
    torch.manual_seed(5)
    dim, heads = 96, 3
    layer = swin.BasicLayer(dim, 2, heads, window_size=7, downsample=True).to(torch.bfloat16)
    out_norm = torch.nn.LayerNorm(dim).to(torch.bfloat16)
    for n, p in list(layer.named_parameters()) + list(out_norm.named_parameters()):
        if "norm" in n or "relative_position_bias_table" in n or p in set(out_norm.parameters()):
            p.requires_grad_(False)                                 # the reference's freezing rule (models/swin/backbone.py:66-69)
    with torch.no_grad():
        for m in layer.modules():
            if isinstance(m, swin.WindowAttention):
                m.relative_position_bias_table.add_((0.3 * torch.randn(m.relative_position_bias_table.shape)).to(torch.bfloat16))
    x0 = torch.randn(1, 10, 17, dim).to(torch.bfloat16)          # 10 x 17 -> padded to 14 x 21 windows of 7
    g1, g2 = torch.randn(1, 10, 17, dim).to(torch.bfloat16), torch.randn(1, 5, 9, 2 * dim).to(torch.bfloat16)
    params = [p for p in layer.parameters() if p.requires_grad]

    def run(on):
        norm.fused_wide_layer_norm = swin.fused_window_attention = on
        for p in params:
            p.grad = None
        x = x0.clone().requires_grad_(True)
        _, down, normed = layer(x, out_norm)
        kinds, seen = set(), {}                                   # (the wrappers are kept alive: ids of dead ones are reused)
        stack = [normed.grad_fn, down.grad_fn]
        while stack:
            n = stack.pop()
            if n is not None and id(n) not in seen:
                seen[id(n)] = n
                kinds.add(type(n).__name__)
                stack.extend(f for f, _ in n.next_functions)
        ((normed.float() * g1.float()).sum() + (down.float() * g2.float()).sum()).backward()
        return [normed.detach().float(), down.detach().float(), x.grad.float()] + [p.grad.float() for p in params], kinds
    fused, kinds = run(True)
    plain, kinds_plain = run(False)
    norm.fused_wide_layer_norm = swin.fused_window_attention = False
    # (the blocks run the image-order form of the attention kernel: no pad / roll / partition copies around it)
    assert {"WindowAttentionRowsFunctionBackward", "WideAddLayerNormFunctionBackward", "WideLayerNormFunctionBackward"} <= kinds, kinds
    assert "RollBackward0" not in kinds                              # (a pad remains: patch merging pads odd maps)
    assert not any("Wide" in k or "WindowAttention" in k for k in kinds_plain) and "RollBackward0" in kinds_plain
    for a, b in zip(fused[:2], plain[:2]):
        assert float((a - b).abs().max()) <= 2.0 ** -5 * float(b.abs().max())
    for a, b in zip(fused[2:], plain[2:]):
        assert float((a - b).norm()) <= 4e-2 * float(b.norm()) + 1e-3, (float((a - b).norm()), float(b.norm()))


def test_whole_backbone_with_both_routes(on_model):
    """SwinTransformer at the Swin-T widths (96 / 192 / 384 / 768: every width the LayerNorm kernel is built for that a backbone
    preset uses below Swin-L's 1536) on an image whose maps need patch, window and merge padding at every stage, drop path off:
    the three output maps and the gradients of the image and of every trainable parameter, both routes on against both off"""
    torch.manual_seed(11)
    m = swin.SwinTransformer(embed_dim=96, depths=(2, 2, 2, 2), num_heads=(3, 6, 12, 24), window_size=7, drop_path_rate=0.0,
                             out_indices=(1, 2, 3)).to(torch.bfloat16)
    for n, p in m.named_parameters():
        if "norm" in n or "relative_position_bias_table" in n:
            p.requires_grad_(False)
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, swin.WindowAttention):
                mod.relative_position_bias_table.add_((0.3 * torch.randn(mod.relative_position_bias_table.shape)).to(torch.bfloat16))
    img = torch.randn(1, 3, 70, 90).to(torch.bfloat16)
    params = [p for p in m.parameters() if p.requires_grad]
    probes = None

    def run(on):
        nonlocal probes
        norm.fused_wide_layer_norm = swin.fused_window_attention = on
        for p in params:
            p.grad = None
        x = img.clone().requires_grad_(True)
        outs = m(x)
        if probes is None:
            probes = {k: torch.randn(v.shape).to(torch.bfloat16) for k, v in outs.items()}
        loss = sum((v.float() * probes[k].float()).sum() for k, v in outs.items())
        seen, stack = {}, [loss.grad_fn]
        while stack:
            n = stack.pop()
            if n is not None and id(n) not in seen:
                seen[id(n)] = n
                stack.extend(f for f, _ in n.next_functions)
        names = [type(n).__name__ for n in seen.values()]
        loss.backward()
        return ([v.detach().float() for _, v in sorted(outs.items())], [x.grad.float()] + [p.grad.float() for p in params],
                {k: names.count(k) for k in set(names)})
    outs_on, grads_on, kinds = run(True)
    outs_off, grads_off, kinds_off = run(False)
    norm.fused_wide_layer_norm = swin.fused_window_attention = False
    assert len(outs_on) == 3
    # every one of the 8 blocks took the image-order attention kernel and both of its fused residual norms
    assert kinds.get("WindowAttentionRowsFunctionBackward") == 8 and "RollBackward0" not in kinds, kinds
    assert kinds.get("WideAddLayerNormFunctionBackward", 0) + kinds.get("WideLayerNormFunctionBackward", 0) >= 16 + 3, kinds
    assert not any("Wide" in k or "WindowAttention" in k for k in kinds_off)
    for a, b in zip(outs_on, outs_off):
        assert float((a - b).norm()) <= 2e-2 * float(b.norm())
    whole = sum(float((a - b).norm()) ** 2 for a, b in zip(grads_on, grads_off)) ** 0.5
    assert whole <= 4e-2 * sum(float(b.norm()) ** 2 for b in grads_off) ** 0.5
    for a, b in zip(grads_on, grads_off):
        assert float((a - b).norm()) <= 8e-2 * float(b.norm()) + 1e-3, (float((a - b).norm()), float(b.norm()))
