"""Dense helper kernels of the train step -- fused clip + AdamW (csrc/fused_adamw.hip), residual add + LayerNorm forward and
backward (csrc/add_layernorm.hip), the backbone's add + ReLU / affine + ReLU tails (csrc/elementwise.hip) -- built for the
CPU against the lane-level workgroup model (tools/emu/build_dense_lib.sh: the same sources hipcc compiles) and checked against
float32 PyTorch, the checker of their GPU tests (tests/test_optim_gpu.py, test_norm_gpu.py, test_linear_gpu.py).  Kernel logic
without a GPU; test infrastructure only -- the product never loads these libraries (its own CPU path is the op's CPU twins, csrc/msda_cpu.cpp)."""
import ctypes
import math
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_cell_forward_emulated import CLANG  # noqa: E402

pytestmark = pytest.mark.skipif(not os.path.exists(CLANG), reason="needs the ROCm clang++ as host compiler")
# Kernels whose device code is instruction-identical to a build that passed the GPU suite (tools/isa_audit.py,
# profiles/r04_isa_audit_vs_round2_gputest.txt) gain nothing from the host model: their cases run only when asked for.  The
# fused optimiser (its `grad_scale` argument has not run on hardware) and the weight-gradient kernel (under work) always run.
unchanged_since_gpu_run = pytest.mark.skipif(os.environ.get("RLIPV2_TEST_EMU_FULL", "0") != "1",
                                             reason="device code unchanged since a green GPU run; RLIPV2_TEST_EMU_FULL=1 runs it")
vp, ci = ctypes.c_void_p, ctypes.c_int


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("emu_dense") / "libdense_emu.so")
    subprocess.run([os.path.join(ROOT, "tools", "emu", "build_dense_lib.sh"), so], check=True, capture_output=True, timeout=900)
    return ctypes.CDLL(so)


def ptr(t):
    return t.data_ptr()


class _Group(ctypes.Structure):
    _fields_ = [(n, ctypes.c_float) for n in ("beta1", "beta2", "one_minus_beta1", "one_minus_beta2", "eps", "decay",
                                              "step_size", "bias_correction2_sqrt")]


@pytest.mark.parametrize("max_norm,grad_scale", [(0.1, 1.0), (0.0, 1.0), (0.1, 0.25)])
def test_fused_adamw_against_torch(lib, max_norm, grad_scale):
    """two parameter groups, tensors of 1 / 37*5 / 2.4 chunks, four steps with gradients of very different size; grad_scale:
    the data-parallel SUM route (the kernels multiply, the reference gets pre-scaled gradients)"""
    sizes = ctypes.c_int * 4
    s = sizes()
    lib.adamw_abi_sizes(ctypes.byref(s, 0), ctypes.byref(s, 4), ctypes.byref(s, 8), ctypes.byref(s, 12))
    assert (s[0], s[1], s[2]) == (56, 8, ctypes.sizeof(_Group))
    chunk = s[3]
    torch.manual_seed(0)
    shapes, group_of, lrs = [(1,), (37, 5), (300, 129)], [0, 1, 0], [1e-2, 5e-3]
    wd, b1, b2, eps = 1e-2, 0.9, 0.999, 1e-8
    master = [torch.randn(sh) for sh in shapes]
    params = [m.to(torch.bfloat16) for m in master]
    master = [p.float().clone() for p in params]
    m1 = [torch.zeros_like(m) for m in master]
    m2 = [torch.zeros_like(m) for m in master]
    ref_params = [m.clone().requires_grad_(True) for m in master]
    ref = torch.optim.AdamW([{"params": [ref_params[0], ref_params[2]]}, {"params": [ref_params[1]], "lr": lrs[1]}],
                            lr=lrs[0], weight_decay=wd, betas=(b1, b2), eps=eps)
    grads = [torch.zeros(sh, dtype=torch.bfloat16) for sh in shapes]
    rows = np.array([[ptr(g), ptr(m), ptr(a), ptr(b), ptr(p), p.numel(), gi]
                     for g, m, a, b, p, gi in zip(grads, master, m1, m2, params, group_of)], dtype=np.int64)
    chunks = np.array([(r, c) for r, p in enumerate(params) for c in range((p.numel() + chunk - 1) // chunk)], dtype=np.int32)
    sq = torch.zeros(1)
    lib.adamw_grad_sqnorm_bf16.argtypes = [vp, vp, ci, vp, vp]
    lib.adamw_step_scaled_bf16.argtypes = [vp, vp, ci, vp, ctypes.c_float, ctypes.c_float, vp, ci, vp]
    gen = torch.Generator().manual_seed(1)
    for step in range(1, 5):
        for g, r in zip(grads, ref_params):
            g.copy_((torch.randn(g.shape, generator=gen) * (10.0 if step % 2 else 0.01)).to(torch.bfloat16))
            r.grad = g.float() * grad_scale
        if max_norm > 0:
            total = torch.nn.utils.clip_grad_norm_(ref_params, max_norm)
        ref.step()
        groups = (_Group * 2)()
        for gi, lr in enumerate(lrs):
            groups[gi] = _Group(b1, b2, 1.0 - b1, 1.0 - b2, eps, 1.0 - lr * wd, lr / (1.0 - b1 ** step),
                                math.sqrt(1.0 - b2 ** step))
        if max_norm > 0:
            assert lib.adamw_grad_sqnorm_bf16(rows.ctypes.data, chunks.ctypes.data, len(chunks), ptr(sq), None) == 0
            torch.testing.assert_close(sq.sqrt()[0] * grad_scale, total, rtol=1e-5, atol=0)
        assert lib.adamw_step_scaled_bf16(rows.ctypes.data, chunks.ctypes.data, len(chunks), ptr(sq), max_norm, grad_scale,
                                          groups, 2, None) == 0
        for m, p, r in zip(master, params, ref_params):
            torch.testing.assert_close(m, r.detach(), rtol=5e-6, atol=1e-6 * float(r.detach().abs().max()))
            assert torch.equal(p, m.to(torch.bfloat16))
    for i, r in enumerate(ref_params):
        st = ref.state[r]
        torch.testing.assert_close(m1[i], st["exp_avg"], rtol=5e-6, atol=1e-6 * float(st["exp_avg"].abs().max()))
        # (the host build has no fused multiply-add: one more rounding per step in b2 * v + (1 - b2) * g * g than the device)
        torch.testing.assert_close(m2[i], st["exp_avg_sq"], rtol=2e-5, atol=1e-6 * float(st["exp_avg_sq"].abs().max()))


@pytest.mark.parametrize("C,rows,form", [(192, 333, "pre_norm"), (384, 70, "pre_norm"), (768, 41, "plain"), (1536, 19, "pre_norm"),
                                         (96, 130, "post_norm"), (128, 67, "pre_norm"), (512, 9, "plain"), (1024, 13, "pre_norm")])
def test_wide_layernorm_forward_and_backward_against_torch(lib, C, rows, form):
    """csrc/layernorm_wide.hip (round 5, never run on hardware) at every supported width.  pre_norm: s = a + b rounded to bf16
    and written out, y = LN(s), backward dx = LN'(dy) + ds (the Swin block, models/swin/swin_transformer.py:386-401);
    plain: y = LN(a); post_norm: y = LN(a + b) with the sum kept in float32.  Against float32 PyTorch on the same operands."""
    eps = 1e-5
    lg = ctypes.c_long
    lib.layernorm_wide_supported.argtypes = [lg, ci]
    assert lib.layernorm_wide_supported(rows, C) == 1 and lib.layernorm_wide_supported(rows, 256) == 0
    torch.manual_seed(C + rows)
    a = torch.randn(rows, C).to(torch.bfloat16)
    b = None if form == "plain" else (0.5 * torch.randn(rows, C)).to(torch.bfloat16)
    gamma = (1.0 + 0.1 * torch.randn(C)).to(torch.bfloat16)
    beta = (0.1 * torch.randn(C)).to(torch.bfloat16)
    dy = torch.randn(rows, C).to(torch.bfloat16)
    ds = torch.randn(rows, C).to(torch.bfloat16) if form == "pre_norm" else None
    y, s_out = torch.empty_like(a), (torch.empty_like(a) if form == "pre_norm" else None)
    mean, rstd = torch.empty(rows), torch.empty(rows)
    lib.layernorm_wide_forward_bf16.argtypes = [vp, vp, vp, vp, lg, ci, ctypes.c_float, vp, vp, vp, vp, vp]
    assert lib.layernorm_wide_forward_bf16(ptr(a), ptr(b) if b is not None else None, ptr(gamma), ptr(beta), rows, C, eps, ptr(y),
                                           ptr(s_out) if s_out is not None else None, ptr(mean), ptr(rstd), None) == 0
    if form == "pre_norm":
        assert torch.equal(s_out, (a.float() + b.float()).to(torch.bfloat16))        # the sum the next block reads, bit for bit
        x = s_out.float().requires_grad_(True)                                       # statistics are taken from the rounded sum
    else:
        x = (a.float() + (b.float() if b is not None else 0)).requires_grad_(True)
    ref = torch.nn.functional.layer_norm(x, (C,), gamma.float(), beta.float(), eps)
    torch.testing.assert_close(y.float(), ref.detach(), rtol=2.0 ** -7, atol=2.0 ** -7)        # one bfloat16 rounding
    torch.testing.assert_close(mean, x.detach().mean(1), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(rstd, (x.detach().var(1, unbiased=False) + eps).rsqrt(), rtol=1e-5, atol=1e-6)
    ref.backward(dy.float())
    want = x.grad + (ds.float() if ds is not None else 0)
    dx = torch.empty_like(a)
    x_saved = s_out if form == "pre_norm" else (a if form == "plain" else (a.float() + b.float()).to(torch.bfloat16))
    lib.layernorm_wide_backward_bf16.argtypes = [vp, vp, vp, vp, vp, vp, lg, ci, vp, vp]
    assert lib.layernorm_wide_backward_bf16(ptr(dy), ptr(ds) if ds is not None else None, ptr(x_saved), ptr(gamma), ptr(mean),
                                            ptr(rstd), rows, C, ptr(dx), None) == 0
    tol = 2.0 ** -6 if form != "post_norm" else 2.0 ** -5       # (post_norm: the backward sees the ROUNDED sum as its input)
    torch.testing.assert_close(dx.float(), want, rtol=tol, atol=tol * float(want.abs().max()))
    # argument checks: unsupported width, missing operands, a sum output without a second addend, misaligned pointers
    assert lib.layernorm_wide_forward_bf16(ptr(a), None, ptr(gamma), ptr(beta), rows, 200, eps, ptr(y), None, ptr(mean), ptr(rstd), None) != 0
    assert lib.layernorm_wide_forward_bf16(ptr(a), None, ptr(gamma), ptr(beta), rows, C, eps, ptr(y), ptr(y), ptr(mean), ptr(rstd), None) != 0
    assert lib.layernorm_wide_forward_bf16(None, None, ptr(gamma), ptr(beta), rows, C, eps, ptr(y), None, ptr(mean), ptr(rstd), None) != 0
    assert lib.layernorm_wide_backward_bf16(ptr(dy) + 2, None, ptr(x_saved), ptr(gamma), ptr(mean), ptr(rstd), rows, C, ptr(dx), None) != 0


@unchanged_since_gpu_run
@pytest.mark.parametrize("rows,with_b", [(700, True), (129, False)])
def test_add_layernorm_forward_and_backward_against_torch(lib, rows, with_b):
    C, eps = 256, 1e-5
    assert lib.add_layernorm_supported(ctypes.c_long(rows), C) == 1
    torch.manual_seed(2)
    a = torch.randn(rows, C).to(torch.bfloat16)
    b = torch.randn(rows, C).to(torch.bfloat16) if with_b else None
    gamma = (1.0 + 0.1 * torch.randn(C)).to(torch.bfloat16)
    beta = (0.1 * torch.randn(C)).to(torch.bfloat16)
    dy = torch.randn(rows, C).to(torch.bfloat16)
    y = torch.empty_like(a)
    mean, rstd = torch.empty(rows), torch.empty(rows)
    lib.add_layernorm_forward_bf16.argtypes = [vp, vp, vp, vp, ctypes.c_long, ci, ctypes.c_float, vp, vp, vp, vp]
    assert lib.add_layernorm_forward_bf16(ptr(a), ptr(b) if with_b else None, ptr(gamma), ptr(beta), rows, C, eps, ptr(y),
                                          ptr(mean), ptr(rstd), None) == 0
    x = (a.float() + (b.float() if with_b else 0)).requires_grad_(True)
    g32, b32 = gamma.float().requires_grad_(True), beta.float().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(x, (C,), g32, b32, eps)
    torch.testing.assert_close(y.float(), ref.detach(), rtol=2.0 ** -7, atol=2.0 ** -7)        # one bfloat16 rounding
    torch.testing.assert_close(mean, x.detach().mean(1), rtol=1e-5, atol=1e-6)
    ref.backward(dy.float())
    lib.add_layernorm_workspace_bytes.restype = ctypes.c_size_t
    lib.add_layernorm_workspace_bytes.argtypes = [ctypes.c_long, ci]
    wsb = lib.add_layernorm_workspace_bytes(rows, C)
    ws = torch.zeros(max(wsb, 16), dtype=torch.uint8)
    dx, dgamma, dbeta = torch.empty_like(a), torch.empty_like(gamma), torch.empty_like(beta)
    lib.add_layernorm_backward_bf16.argtypes = [vp, vp, vp, vp, vp, vp, ctypes.c_long, ci, vp, vp, vp, vp, ctypes.c_size_t, vp]
    assert lib.add_layernorm_backward_bf16(ptr(dy), ptr(a), ptr(b) if with_b else None, ptr(gamma), ptr(mean), ptr(rstd), rows,
                                           C, ptr(dx), ptr(dgamma), ptr(dbeta), ptr(ws), wsb, None) == 0
    tol = dict(rtol=2.0 ** -6, atol=2.0 ** -6)
    torch.testing.assert_close(dx.float(), x.grad, rtol=2.0 ** -6, atol=2.0 ** -6 * float(x.grad.abs().max()))
    torch.testing.assert_close(dgamma.float(), g32.grad, rtol=2.0 ** -6, atol=2.0 ** -6 * float(g32.grad.abs().max()))
    torch.testing.assert_close(dbeta.float(), b32.grad, rtol=2.0 ** -6, atol=2.0 ** -6 * float(b32.grad.abs().max()))
    del tol


@unchanged_since_gpu_run
def test_elementwise_tails_against_torch(lib):
    torch.manual_seed(3)
    n, C = 8 * 1000 + 8 * 3, 64                                     # (multiples of 8 elements: the kernels' vector width)
    a, b = torch.randn(n).to(torch.bfloat16), torch.randn(n).to(torch.bfloat16)
    y = torch.empty_like(a)
    lib.add_relu_bf16.argtypes = [vp, vp, vp, ctypes.c_long, vp]
    assert lib.add_relu_bf16(ptr(a), ptr(b), ptr(y), n, None) == 0
    assert torch.equal(y, torch.relu(a.float() + b.float()).to(torch.bfloat16))
    rows = 200
    x = torch.randn(rows, C).to(torch.bfloat16)
    scale, bias = (1 + 0.2 * torch.randn(C)).to(torch.bfloat16), torch.randn(C).to(torch.bfloat16)
    y = torch.empty_like(x)
    lib.affine_relu_bf16.argtypes = [vp, vp, vp, vp, ctypes.c_long, ci, vp]
    assert lib.affine_relu_bf16(ptr(x), ptr(scale), ptr(bias), ptr(y), rows * C, C, None) == 0
    ref = torch.relu(x.float() * scale.float() + bias.float())
    torch.testing.assert_close(y.float(), ref, rtol=2.0 ** -7, atol=2.0 ** -8)
    dy = torch.randn(rows, C).to(torch.bfloat16)
    dx = torch.empty_like(x)
    lib.affine_relu_backward_bf16.argtypes = [vp, vp, vp, vp, ctypes.c_long, ci, vp]
    assert lib.affine_relu_backward_bf16(ptr(dy), ptr(y), ptr(scale), ptr(dx), rows * C, C, None) == 0
    refdx = dy.float() * scale.float() * (y.float() > 0)
    torch.testing.assert_close(dx.float(), refdx, rtol=2.0 ** -7, atol=2.0 ** -8)


@unchanged_since_gpu_run
def test_dab_box_refinement_against_the_reference_formula(lib):
    """dab_refine_boxes: sigmoid(delta + inverse_sigmoid(ref)) with the reference's clamp and eps (util/misc.py:460-464,
    dab_deformable/deformable_transformer.py:1511-1541), float32 and bfloat16 deltas, values on and beyond the clamp"""
    torch.manual_seed(4)
    rows, eps = 301, 1e-5
    ref = torch.rand(rows, 4)
    ref[0] = torch.tensor([0.0, 1.0, -0.2, 1.3])                         # clamped to [0, 1], then eps
    for bf16 in (0, 1):
        delta = torch.randn(rows, 4) * 2
        d = delta.to(torch.bfloat16) if bf16 else delta
        out = torch.empty(rows, 4)
        lib.dab_refine_boxes.argtypes = [vp, ci, vp, vp, ctypes.c_long, ctypes.c_float, vp]
        assert lib.dab_refine_boxes(ptr(d), bf16, ptr(ref), ptr(out), rows, eps, None) == 0
        x = ref.clamp(0, 1)
        inv = torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))
        want = torch.sigmoid(d.float() + inv)
        torch.testing.assert_close(out, want, rtol=2e-6, atol=1e-7)


@unchanged_since_gpu_run
def test_token_major_group_norm_against_torch(lib):
    """GroupNorm(32, 256) of every pyramid level straight into its slice of the flattened [N, S, 256] tensor, forward and
    backward (csrc/groupnorm_tokens.hip; reference input_proj's GroupNorm + the flatten, models/hoi.py:1936-1957,
    dab_deformable/deformable_transformer.py:520-547) against float32 torch.nn.functional.group_norm"""
    torch.manual_seed(5)
    N, C, G, eps = 2, 256, 32, 1e-5
    hw = [35, 12, 5, 2]
    levels = len(hw)
    assert lib.groupnorm_tokens_supported(C, G, levels) == 1
    xs = [torch.randn(N, h, C).to(torch.bfloat16) for h in hw]
    gammas = [(1 + 0.2 * torch.randn(C)).to(torch.bfloat16) for _ in hw]
    betas = [(0.2 * torch.randn(C)).to(torch.bfloat16) for _ in hw]
    S = sum(hw)
    out = torch.empty(N, S, C, dtype=torch.bfloat16)
    mean, rstd = torch.empty(levels, N, G), torch.empty(levels, N, G)
    hw_c = (ci * levels)(*hw)
    parr = lambda ts: (vp * len(ts))(*[t.data_ptr() for t in ts])                    # noqa: E731
    lib.groupnorm_tokens_workspace_bytes.restype = ctypes.c_size_t
    lib.groupnorm_tokens_workspace_bytes.argtypes = [ci, vp, ci]
    wsb = lib.groupnorm_tokens_workspace_bytes(N, hw_c, levels)
    ws = torch.zeros(max(wsb, 16), dtype=torch.uint8)
    lib.groupnorm_tokens_forward_bf16.argtypes = [vp, vp, ci, ci, vp, vp, ctypes.c_float, vp, vp, vp, vp, ctypes.c_size_t, vp]
    assert lib.groupnorm_tokens_forward_bf16(parr(xs), hw_c, levels, N, parr(gammas), parr(betas), eps, ptr(out), ptr(mean),
                                             ptr(rstd), ptr(ws), wsb, None) == 0
    dy = torch.randn(N, S, C).to(torch.bfloat16)
    dxs = [torch.empty_like(x) for x in xs]
    dgs, dbs = [torch.empty_like(g) for g in gammas], [torch.empty_like(g) for g in gammas]
    lib.groupnorm_tokens_backward_bf16.argtypes = [vp, vp, vp, ci, ci, vp, vp, vp, vp, vp, vp, vp, ctypes.c_size_t, vp]
    assert lib.groupnorm_tokens_backward_bf16(ptr(dy), parr(xs), hw_c, levels, N, parr(gammas), ptr(mean), ptr(rstd), parr(dxs),
                                              parr(dgs), parr(dbs), ptr(ws), wsb, None) == 0
    start = 0
    for l, h in enumerate(hw):
        x = xs[l].float().requires_grad_(True)
        g32, b32 = gammas[l].float().requires_grad_(True), betas[l].float().requires_grad_(True)
        ref = torch.nn.functional.group_norm(x.transpose(1, 2), G, g32, b32, eps).transpose(1, 2)      # [N, C, hw] -> tokens
        torch.testing.assert_close(out[:, start:start + h].float(), ref.detach(), rtol=2.0 ** -7, atol=2.0 ** -7)
        ref.backward(dy[:, start:start + h].float())
        for got, want in ((dxs[l], x.grad), (dgs[l], g32.grad), (dbs[l], b32.grad)):
            torch.testing.assert_close(got.float(), want, rtol=2.0 ** -6, atol=2.0 ** -6 * float(want.abs().max()))
        start += h


@unchanged_since_gpu_run
@pytest.mark.parametrize("dropout", [False, True])
def test_alif_attention_core_against_torch(lib, dropout):
    """alif_attention_forward_bf16 / alif_attention_softmax_backward_bf16 (csrc/alif_attention.hip: shared logits q k^T, a
    softmax over the text tokens for the vision side and over the vision tokens for the language side, dropout, both value
    products on v_mfma_f32_32x32x16_bf16; reference models/fuse_helper.py:365-466) against the float32 formula"""
    torch.manual_seed(6)
    B, H, Tv, Tl, HD = 1, 2, 45, 11, 256
    E = H * HD
    assert lib.alif_attention_supported(B, H, Tv, Tl, HD) == 1
    Tvp = lib.alif_attention_padded_tv(Tv)
    q = (torch.randn(B, Tv, E) * 0.08).to(torch.bfloat16)
    k = torch.randn(B, Tl, E).to(torch.bfloat16)
    val_l = torch.randn(B, Tl, E).to(torch.bfloat16)
    val_v = torch.randn(B, Tv, E).to(torch.bfloat16)
    vlt = torch.zeros(B, E, 64, dtype=torch.bfloat16)
    vlt[:, :, :Tl] = val_l.transpose(1, 2)
    vvt = torch.zeros(B, E, Tvp, dtype=torch.bfloat16)
    vvt[:, :, :Tv] = val_v.transpose(1, 2)
    p_drop = 0.1
    keep_v = (torch.rand(B, H, Tv, Tl) >= p_drop).to(torch.uint8) if dropout else None
    keep_l = (torch.rand(B, H, Tl, Tv) >= p_drop).to(torch.uint8) if dropout else None
    scale = 1.0 / (1.0 - p_drop) if dropout else 1.0
    out_v, out_l = torch.zeros(B, Tv, E, dtype=torch.bfloat16), torch.zeros(B, Tl, E, dtype=torch.bfloat16)
    probs_v = torch.zeros(B, H, Tv, Tl, dtype=torch.bfloat16)
    probs_l = torch.zeros(B, H, Tl, Tv, dtype=torch.bfloat16)
    lib.alif_attention_forward_bf16.argtypes = [vp, vp, vp, vp, vp, vp, ctypes.c_float, ci, ci, ci, ci, vp, vp, vp, vp, vp]
    assert lib.alif_attention_forward_bf16(ptr(q), ptr(k), ptr(vlt), ptr(vvt), ptr(keep_v) if dropout else None,
                                           ptr(keep_l) if dropout else None, scale, B, H, Tv, Tl, ptr(out_v), ptr(out_l),
                                           ptr(probs_v), ptr(probs_l), None) == 0
    heads = lambda t, T: t.float().view(B, T, H, HD).transpose(1, 2)                 # noqa: E731  [B, H, T, HD]
    S = heads(q, Tv) @ heads(k, Tl).transpose(-1, -2)                                # [B, H, Tv, Tl]
    Pv, Pl = torch.softmax(S, -1), torch.softmax(S.transpose(-1, -2), -1)
    torch.testing.assert_close(probs_v.float(), Pv, rtol=2.0 ** -7, atol=2.0 ** -9)
    torch.testing.assert_close(probs_l.float(), Pl, rtol=2.0 ** -7, atol=2.0 ** -9)
    # the value products consume the bfloat16 probabilities the kernel keeps in LDS
    Dv = probs_v.float() * (keep_v.float() * scale if dropout else 1.0)
    Dl = probs_l.float() * (keep_l.float() * scale if dropout else 1.0)
    want_v = (Dv.to(torch.bfloat16).float() @ heads(val_l, Tl)).transpose(1, 2).reshape(B, Tv, E)
    want_l = (Dl.to(torch.bfloat16).float() @ heads(val_v, Tv)).transpose(1, 2).reshape(B, Tl, E)
    torch.testing.assert_close(out_v.float(), want_v, rtol=2.0 ** -6, atol=2.0 ** -6 * float(want_v.abs().max()))
    torch.testing.assert_close(out_l.float(), want_l, rtol=2.0 ** -6, atol=2.0 ** -6 * float(want_l.abs().max()))
    # backward of the two softmaxes (+ dropouts) into the shared logits
    d_pv = torch.randn(B, H, Tv, Tl).to(torch.bfloat16)
    d_pl = torch.randn(B, H, Tl, Tv).to(torch.bfloat16)
    d_logits = torch.zeros(B, H, Tv, Tl, dtype=torch.bfloat16)
    dropped_v, dropped_l = torch.zeros_like(probs_v), torch.zeros_like(probs_l)
    lib.alif_attention_softmax_backward_bf16.argtypes = [vp, vp, vp, vp, vp, vp, ctypes.c_float, ci, ci, ci, ci, vp, vp, vp, vp]
    assert lib.alif_attention_softmax_backward_bf16(ptr(probs_v), ptr(probs_l), ptr(d_pv), ptr(d_pl),
                                                    ptr(keep_v) if dropout else None, ptr(keep_l) if dropout else None, scale,
                                                    B, H, Tv, Tl, ptr(d_logits), ptr(dropped_v) if dropout else None,
                                                    ptr(dropped_l) if dropout else None, None) == 0
    pv, pl = probs_v.float(), probs_l.float()
    gv = d_pv.float() * (keep_v.float() * scale if dropout else 1.0)
    gl = d_pl.float() * (keep_l.float() * scale if dropout else 1.0)
    ds_v = pv * (gv - (pv * gv).sum(-1, keepdim=True))
    ds_l = pl * (gl - (pl * gl).sum(-1, keepdim=True))
    want = ds_v + ds_l.transpose(-1, -2)
    torch.testing.assert_close(d_logits.float(), want, rtol=2.0 ** -6, atol=2.0 ** -6 * float(want.abs().max()))
    if dropout:
        torch.testing.assert_close(dropped_v.float(), (pv * keep_v.float() * scale), rtol=2.0 ** -7, atol=2.0 ** -9)
        torch.testing.assert_close(dropped_l.float(), (pl * keep_l.float() * scale), rtol=2.0 ** -7, atol=2.0 ** -9)


@pytest.mark.parametrize("T,M,K,f32", [(100, 128, 128, False), (256, 256, 128, True), (1203, 128, 256, False)])
def test_mfma_weight_gradient_against_torch(lib, T, M, K, f32):
    """linear_wgrad_bf16 (csrc/token_gemm.hip: dW = dY^T X and db = column sums of dY on v_mfma_f32_32x32x16_bf16, LDS-DMA
    staged tiles read through transposing LDS reads, chunk partials + a reduce pass that also takes the T % 32 tail rows; one
    chunk and no tail: the direct-store instantiation) against float32 matmul -- 100 rows (3 steps + 4 tail rows), 256 rows
    (direct), 1 203 rows (several chunks + 19 tail rows)"""
    torch.manual_seed(7)
    dy = torch.randn(T, M).to(torch.bfloat16)
    x = torch.randn(T, K).to(torch.bfloat16)
    lib.linear_wgrad_supported.argtypes = [ci, ci, ci]
    assert lib.linear_wgrad_supported(T, M, K) == 1
    lib.linear_wgrad_workspace_bytes.restype = ctypes.c_size_t
    lib.linear_wgrad_workspace_bytes.argtypes = [ci, ci, ci]
    wsb = lib.linear_wgrad_workspace_bytes(T, M, K)
    ws = torch.zeros(wsb + 64, dtype=torch.uint8)
    odt = torch.float32 if f32 else torch.bfloat16
    dw, db = torch.full((M, K), float("nan"), dtype=odt), torch.full((M,), float("nan"), dtype=odt)
    lib.linear_wgrad_bf16.argtypes = [vp, vp, ci, ci, ci, vp, vp, ci, vp, ctypes.c_size_t, vp]
    assert lib.linear_wgrad_bf16(ptr(dy), ptr(x), T, M, K, ptr(dw), ptr(db), 1 if f32 else 0, ptr(ws), wsb, None) == 0
    want_w = dy.float().t() @ x.float()
    want_b = dy.float().sum(0)
    tol = 1e-5 if f32 else 2.0 ** -7
    torch.testing.assert_close(dw.float(), want_w, rtol=tol, atol=tol * float(want_w.abs().max()))
    torch.testing.assert_close(db.float(), want_b, rtol=tol, atol=tol * float(want_b.abs().max()))


@unchanged_since_gpu_run
@pytest.mark.parametrize("T,N,mask,bias,relu", [(300, 128, False, True, True), (77, 192, True, False, False),
                                                (256, 64, False, False, False), (513, 128, True, True, False)])
def test_mfma_expand_gemm_against_torch(lib, T, N, mask, bias, relu):
    """linear_expand_bf16 (csrc/expand_gemm.hip: C[T, N] = A[T, 256] B[N, 256]^T on v_mfma_f32_32x32x16_bf16 with the A slice
    resident in registers, LDS-DMA'd swizzled B tiles, a per-wave swizzled staging tile that the mask rows reach by DMA and the
    results replace; epilogue = + bias, ReLU, keep-where-mask-positive) against float32 matmul: ragged last row block, one to
    three column steps, every epilogue combination the encoder FFN uses"""
    torch.manual_seed(11)
    a = torch.randn(T, 256).to(torch.bfloat16)
    b = (torch.randn(N, 256) / 16).to(torch.bfloat16)
    bv = torch.randn(N).to(torch.bfloat16) if bias else None
    mk = torch.randn(T, N).to(torch.bfloat16) if mask else None
    if mask:
        mk[::7, ::5] = 0.0           # zeros and negative zeros are "not positive"
        mk[1::7, 1::5] = -0.0
    c = torch.full((T, N), float("nan"), dtype=torch.bfloat16)
    lib.linear_expand_supported.argtypes = [ci, ci, ci]
    assert lib.linear_expand_supported(T, N, 256) == 1 and lib.linear_expand_supported(T, N, 128) == 0
    lib.linear_expand_bf16.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, vp, vp]
    rc = lib.linear_expand_bf16(ptr(a), ptr(b), ptr(bv) if bias else None, ptr(mk) if mask else None, T, N, 256,
                                1 if relu else 0, ptr(c), None)
    assert rc == 0
    want = a.float() @ b.float().t()
    if bias:
        want = want + bv.float()
    if relu:
        want = want.relu()
    want = want.to(torch.bfloat16).float()
    if mask:
        want = torch.where(mk.float() > 0, want, torch.zeros_like(want))
    torch.testing.assert_close(c.float(), want, rtol=2.0 ** -7, atol=2.0 ** -8)
    if mask:
        assert bool((c[mk.float() <= 0] == 0).all())


def _window_attention_reference(qkv, bias, mask, mask_id, wpi, scale):
    """WindowAttention's core (reference models/swin/swin_transformer.py:272-297) in float32 on the bf16 operands"""
    W, N, _, h, d = qkv.shape
    q, k, v = (qkv[:, :, t].permute(0, 2, 1, 3).float() for t in range(3))           # [W, h, N, d]
    attn = (q * scale) @ k.transpose(-2, -1) + bias[None]
    if mask is not None:
        ids = mask_id[torch.arange(W) % wpi]
        add = torch.where(ids[:, None, None] >= 0, mask[ids.clamp_min(0)], torch.zeros(()))
        attn = attn + add[:, None]
    p = attn.softmax(-1)
    return (p @ v).transpose(1, 2).reshape(W, N, h * d), p


def _padded(t, N, pad_keys):
    """[..., N, N] (query, key) -> [..., 64, 64], `pad_keys` in the key columns >= N, 0 elsewhere (include/rlipv2_swin.h)"""
    out = torch.zeros(*t.shape[:-2], 64, 64)
    out[..., :, N:] = pad_keys
    out[..., :N, :N] = t
    return out.contiguous()


@pytest.mark.parametrize("N,heads,windows,masked", [(49, 6, 5, True), (49, 3, 2, False), (64, 4, 3, True), (16, 12, 4, False),
                                                    (1, 1, 1, False), (33, 5, 7, True)])
def test_window_attention_against_torch(lib, N, heads, windows, masked):
    """csrc/window_attention.hip (round 5, never run on hardware) on the lane-level model: forward and backward of the Swin
    window attention against float32 PyTorch on the same bf16 operands.  N = 49 (window 7) exercises the padded keys /
    queries, `masked` the shift masks through the compact table, heads not a multiple of the 4 waves of a workgroup the
    wave-to-task mapping."""
    torch.manual_seed(N + heads)
    d, wpi = 32, max(1, windows - 1)
    scale = d ** -0.5
    qkv = (0.8 * torch.randn(windows, N, 3, heads, d)).to(torch.bfloat16)
    bias = 0.5 * torch.randn(heads, N, N)
    mask = mask_id = None
    if masked:
        region = torch.randint(0, 3, (2, N))
        mask = torch.where(region[:, :, None] != region[:, None, :], torch.tensor(-100.0), torch.tensor(0.0))   # 2 distinct masks
        mask_id = torch.tensor(([-1, 0, 1, -1, 1] * wpi)[:wpi], dtype=torch.int32)
    bias_t = _padded(bias, N, -30000.0)
    mask_t = _padded(mask, N, 0.0) if masked else None
    out = torch.zeros(windows, N, heads * d, dtype=torch.bfloat16)
    lib.window_attention_supported.argtypes = [ci] * 4
    assert lib.window_attention_supported(windows, heads, N, d) == 1 and lib.window_attention_supported(windows, heads, 65, d) == 0
    lib.window_attention_forward_bf16.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, ctypes.c_float, vp, vp]
    assert lib.window_attention_forward_bf16(ptr(qkv), ptr(bias_t), ptr(mask_t) if masked else None, ptr(mask_id) if masked else None,
                                             windows, wpi, heads, N, scale, ptr(out), None) == 0
    q32 = qkv.float().requires_grad_(True)
    ref, p = _window_attention_reference(q32, bias, mask, mask_id.long() if masked else None, wpi, scale)
    # one bfloat16 rounding of the probabilities and of the result
    torch.testing.assert_close(out.float(), ref.detach(), rtol=2.0 ** -6, atol=2.0 ** -6 * float(ref.abs().max()))
    d_out = torch.randn(windows, N, heads * d).to(torch.bfloat16)
    ref.backward(d_out.float())
    d_qkv = torch.full_like(qkv, float("nan"))
    lib.window_attention_backward_bf16.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci, ci, ctypes.c_float, vp, vp]
    assert lib.window_attention_backward_bf16(ptr(qkv), ptr(d_out), ptr(bias_t), ptr(mask_t) if masked else None,
                                              ptr(mask_id) if masked else None, windows, wpi, heads, N, scale, ptr(d_qkv), None) == 0
    assert torch.isfinite(d_qkv.float()).all()                      # every element written
    want = q32.grad
    for t, name in enumerate(("dq", "dk", "dv")):
        got, w = d_qkv[:, :, t].float(), want[:, :, t]
        err = float((got - w).abs().max()) / max(float(w.abs().max()), 1e-6)      # (N = 1: dq and dk are exactly zero)
        assert err < 2.0 ** -5, (name, err)                          # bf16 probabilities / dS in the products, bf16 results
    # argument checks
    assert lib.window_attention_forward_bf16(ptr(qkv), None, None, None, windows, wpi, heads, N, scale, ptr(out), None) != 0
    assert lib.window_attention_forward_bf16(ptr(qkv), ptr(bias_t), ptr(bias_t), None, windows, wpi, heads, N, scale, ptr(out), None) != 0
    assert lib.window_attention_backward_bf16(ptr(qkv), ptr(d_out), ptr(bias_t), None, None, windows, 0, heads, N, scale, ptr(d_qkv), None) != 0


def swin_row_map(H, W, ws, shift):
    """the image-order row of every token of every window as the reference's pad -> roll -> window_partition lays them out
    (models/swin/swin_transformer.py:362-379); padding positions -> -(slot + 1).  Returns (map [nW * ws * ws] int32, pads)."""
    Hp, Wp = -(-H // ws) * ws, -(-W // ws) * ws
    idx = torch.full((Hp, Wp), 0, dtype=torch.int64)
    real = torch.zeros(Hp, Wp, dtype=torch.bool)
    real[:H, :W] = True
    idx[real] = torch.arange(H * W)
    n_pad = int((~real).sum())
    idx[~real] = -(torch.arange(n_pad) + 1)
    if shift:
        idx = torch.roll(idx, shifts=(-shift, -shift), dims=(0, 1))
    return idx.view(Hp // ws, ws, Wp // ws, ws).permute(0, 2, 1, 3).reshape(-1).to(torch.int32).contiguous(), n_pad


@pytest.mark.parametrize("H,W,ws,shift,heads", [(10, 17, 7, 3, 3), (8, 8, 4, 0, 2), (9, 6, 4, 2, 5)])
def test_window_attention_rows_mode_against_torch(lib, H, W, ws, shift, heads):
    """the window attention kernels with pad / shift / partition / reverse folded into their addressing (row map, image-order
    tensors, padding tokens = the projection's bias, their gradient rows in a side buffer) against the reference's op sequence
    (pad the token map, roll, partition, attention, reverse, roll back, crop) in float32 on the same bf16 operands."""
    torch.manual_seed(H * W + heads)
    B, d, N = 2, 32, ws * ws
    C = heads * d
    scale = d ** -0.5
    rowmap, n_pad = swin_row_map(H, W, ws, shift)
    nW = rowmap.numel() // N
    Hp, Wp = -(-H // ws) * ws, -(-W // ws) * ws
    qkv = (0.8 * torch.randn(B, H * W, 3 * C)).to(torch.bfloat16)
    pad_row = (0.5 * torch.randn(3 * C)).to(torch.bfloat16)
    bias = 0.5 * torch.randn(heads, N, N)
    mask = None
    if shift:
        img = torch.zeros(Hp, Wp)
        cnt = 0
        for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
                img[hs, wsl] = cnt
                cnt += 1
        win = img.view(Hp // ws, ws, Wp // ws, ws).permute(0, 2, 1, 3).reshape(-1, N)
        mask = torch.where(win[:, None, :] - win[:, :, None] != 0, torch.tensor(-100.0), torch.tensor(0.0))      # [nW, N, N]
    # reference: the op sequence in float32
    q32 = qkv.float().requires_grad_(True)
    p32 = pad_row.float().requires_grad_(True)
    full = p32.expand(B, Hp, Wp, 3 * C).clone()
    full[:, :H, :W] = q32.view(B, H, W, 3 * C)
    if shift:
        full = torch.roll(full, shifts=(-shift, -shift), dims=(1, 2))
    win_t = full.view(B, Hp // ws, ws, Wp // ws, ws, 3 * C).permute(0, 1, 3, 2, 4, 5).reshape(B * nW, N, 3, heads, d)
    q, k, v = (win_t[:, :, t].permute(0, 2, 1, 3) for t in range(3))
    attn = (q * scale) @ k.transpose(-2, -1) + bias[None]
    if mask is not None:
        attn = (attn.view(B, nW, heads, N, N) + mask[None, :, None]).view(-1, heads, N, N)
    o = (attn.softmax(-1) @ v).transpose(1, 2).reshape(B, Hp // ws, Wp // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, C)
    if shift:
        o = torch.roll(o, shifts=(shift, shift), dims=(1, 2))
    ref = o[:, :H, :W].reshape(B, H * W, C)
    # kernel
    bias_t = _padded(bias, N, -30000.0)
    mask_t = mask_id = None
    if mask is not None:
        distinct, inv = torch.unique(mask.reshape(nW, -1), dim=0, return_inverse=True)
        nz = distinct.abs().sum(1) != 0
        remap = torch.cumsum(nz.int(), 0) - 1
        mask_id = torch.where(nz[inv], remap[inv], torch.full_like(remap[inv], -1)).to(torch.int32).contiguous()
        mask_t = _padded(distinct[nz].view(-1, N, N), N, 0.0)
    out = torch.full((B, H * W, C), float("nan")).to(torch.bfloat16)
    lg = ctypes.c_long
    del lg
    lib.window_attention_rows_forward_bf16.argtypes = [vp, vp, vp, ci, ci, vp, vp, vp, ci, ci, ci, ci, ctypes.c_float, vp, vp]
    lib.window_attention_rows_backward_bf16.argtypes = [vp, vp, vp, ci, ci, vp, vp, vp, vp, ci, ci, ci, ci, ctypes.c_float, vp, vp, vp]
    mp = lambda t: ptr(t) if t is not None else None                                                     # noqa: E731
    assert lib.window_attention_rows_forward_bf16(ptr(qkv), ptr(pad_row), ptr(rowmap), H * W, n_pad, ptr(bias_t), mp(mask_t), mp(mask_id),
                                                  B * nW, nW, heads, N, scale, ptr(out), None) == 0
    assert torch.isfinite(out.float()).all()                        # every real token's row written
    torch.testing.assert_close(out.float(), ref.detach(), rtol=2.0 ** -6, atol=2.0 ** -6 * float(ref.abs().max()))
    d_out = torch.randn(B, H * W, C).to(torch.bfloat16)
    ref.backward(d_out.float())
    d_qkv = torch.full_like(qkv, float("nan"))
    d_pad = torch.full((B * max(n_pad, 1), 3 * C), float("nan")).to(torch.bfloat16)
    assert lib.window_attention_rows_backward_bf16(ptr(qkv), ptr(pad_row), ptr(rowmap), H * W, n_pad, ptr(d_out), ptr(bias_t), mp(mask_t),
                                                   mp(mask_id), B * nW, nW, heads, N, scale, ptr(d_qkv), ptr(d_pad), None) == 0
    assert torch.isfinite(d_qkv.float()).all()
    err = float((d_qkv.float() - q32.grad).abs().max()) / float(q32.grad.abs().max())
    assert err < 2.0 ** -5, err
    if n_pad:
        got = d_pad.float().sum(0)
        assert torch.isfinite(got).all()
        assert float((got - p32.grad).abs().max()) <= 2.0 ** -4 * float(p32.grad.abs().max()) + 1e-3       # (a sum of bf16-rounded rows)
    # argument checks: the map must cover windows x tokens, the tensors must be aligned
    assert lib.window_attention_rows_forward_bf16(ptr(qkv), ptr(pad_row), ptr(rowmap), H * W + 1, n_pad, ptr(bias_t), mp(mask_t), mp(mask_id),
                                                  B * nW, nW, heads, N, scale, ptr(out), None) != 0
    assert lib.window_attention_rows_forward_bf16(ptr(qkv), ptr(pad_row), None, H * W, n_pad, ptr(bias_t), mp(mask_t), mp(mask_id),
                                                  B * nW, nW, heads, N, scale, ptr(out), None) != 0
