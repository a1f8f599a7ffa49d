"""GPU parity of the token-major weight-gradient kernel (csrc/token_gemm.hip) against a float32
PyTorch reference of the same GEMM (floating-point kernel: torch fp32 is the checker here).

Tolerance: inputs are bf16-exact, products are exact in float32, accumulation is float32 in a different
order than the reference -> relative error ~1e-6 of the sum of |terms|; bf16 outputs add one rounding
(2^-9 relative)."""
import pytest
import torch

gpu = pytest.mark.gpu


def _case(T, M, K, seed=0):
    g = torch.Generator().manual_seed(seed)
    dy = torch.randn(T, M, generator=g).to(torch.bfloat16)
    x = torch.randn(T, K, generator=g).to(torch.bfloat16)
    # transpose-detecting: make rows / columns / token order all distinguishable
    dy *= (1 + torch.arange(M) / M)[None, :].to(torch.bfloat16)
    x *= (1 + 2 * torch.arange(K) / K)[None, :].to(torch.bfloat16)
    return dy.cuda(), x.cuda()


@gpu
@pytest.mark.parametrize("T,M,K", [(1, 128, 128), (31, 128, 128), (32, 128, 128), (33, 256, 128), (1000, 128, 256),
                                   (4099, 384, 256), (22223, 256, 256), (88892, 256, 256), (88892, 1024, 256),
                                   (88892, 256, 1024), (88892, 384, 256)])
@pytest.mark.parametrize("out_dtype", [torch.float32, torch.bfloat16])
def test_wgrad_matches_float32_reference(T, M, K, out_dtype):
    from rlipv2_amd import linear
    dy, x = _case(T, M, K, seed=T + M)
    dw, db = linear.linear_wgrad(dy, x, with_bias=True, out_dtype=out_dtype)
    ref_w = dy.float().t() @ x.float()
    ref_b = dy.float().sum(0)
    scale_w = (dy.float().abs().t() @ x.float().abs())          # sum of |terms| per output
    scale_b = dy.float().abs().sum(0)
    rel = 2e-6 * max(1, T) ** 0.5 if out_dtype == torch.float32 else 2.0 ** -8
    assert dw.dtype == out_dtype and db.dtype == out_dtype
    if out_dtype == torch.float32:
        assert ((dw - ref_w).abs() <= rel * scale_w + 1e-6).all(), float(((dw - ref_w).abs() / (scale_w + 1e-6)).max())
        assert ((db - ref_b).abs() <= rel * scale_b + 1e-6).all()
    else:
        assert ((dw.float() - ref_w).abs() <= rel * ref_w.abs() + 2e-6 * T ** 0.5 * scale_w + 1e-6).all()
        assert ((db.float() - ref_b).abs() <= rel * ref_b.abs() + 2e-6 * T ** 0.5 * scale_b + 1e-6).all()


@gpu
def test_wgrad_exact_on_integer_data():
    """Small-integer operands: every product and partial sum is exact in float32, so the result must be
    bit-identical to the integer GEMM whatever the accumulation order (catches any operand mix-up)."""
    from rlipv2_amd import linear
    g = torch.Generator().manual_seed(3)
    T, M, K = 5000, 256, 384
    dy = torch.randint(-3, 4, (T, M), generator=g)
    x = torch.randint(-3, 4, (T, K), generator=g)
    dw, db = linear.linear_wgrad(dy.to(torch.bfloat16).cuda(), x.to(torch.bfloat16).cuda(), out_dtype=torch.float32)
    assert torch.equal(dw.cpu().long(), dy.t() @ x)
    assert torch.equal(db.cpu().long(), dy.sum(0))
    dw2, no_bias = linear.linear_wgrad(dy.to(torch.bfloat16).cuda(), x.to(torch.bfloat16).cuda(), with_bias=False,
                                       out_dtype=torch.float32)
    assert no_bias is None and torch.equal(dw2, dw)


@gpu
def test_token_linear_gradients_match_library_linear():
    from rlipv2_amd import linear
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 6000, 256, generator=g).to(torch.bfloat16).cuda()
    w = (torch.randn(384, 256, generator=g) * 0.05).to(torch.bfloat16).cuda()
    b = torch.randn(384, generator=g).to(torch.bfloat16).cuda()
    dy = torch.randn(2, 6000, 384, generator=g).to(torch.bfloat16).cuda()
    outs = []
    for fn in (linear.token_linear, torch.nn.functional.linear):
        xx, ww, bb = (t.clone().requires_grad_(True) for t in (x, w, b))
        assert fn is not linear.token_linear or linear.supported(xx, ww)
        y = fn(xx, ww, bb)
        y.backward(dy)
        outs.append((y, xx.grad, ww.grad, bb.grad))
    for a, r in zip(*outs):
        torch.testing.assert_close(a.float(), r.float(), rtol=2e-2, atol=2e-2 * float(r.float().abs().max()))
    # the weight gradient is closer to the float32 result than bf16 rounding of it allows the library to be
    ref = dy.float().flatten(0, 1).t() @ x.float().flatten(0, 1)
    assert (outs[0][2].float() - ref).abs().max() <= 2.0 ** -8 * ref.abs().max() + 1e-3


@gpu
def test_small_linear_and_module_swap():
    """FastLinear (class swap of nn.Linear) keeps parameters / state_dict and reproduces nn.Linear's output and
    gradients on a decoder-sized input (bias gradient as a ones-row GEMM instead of a column reduction)."""
    from rlipv2_amd import linear
    torch.manual_seed(0)
    for dtype, tol in ((torch.float32, 1e-4), (torch.bfloat16, 2e-2)):
        ref = torch.nn.Sequential(torch.nn.Linear(256, 512), torch.nn.ReLU(), torch.nn.Linear(512, 64)).cuda().to(dtype)
        fast = torch.nn.Sequential(torch.nn.Linear(256, 512), torch.nn.ReLU(), torch.nn.Linear(512, 64)).cuda().to(dtype)
        fast.load_state_dict(ref.state_dict())
        assert linear.swap_linears(fast) == 2 and isinstance(fast[0], linear.FastLinear)
        assert list(fast.state_dict().keys()) == list(ref.state_dict().keys())
        x = torch.randn(4, 150, 256, device="cuda", dtype=dtype)
        dy = torch.randn(4, 150, 64, device="cuda", dtype=dtype)
        res = []
        for net in (fast, ref):
            xx = x.clone().requires_grad_(True)
            y = net(xx)
            y.backward(dy)
            res.append([y.detach(), xx.grad] + [p.grad for p in net.parameters()])
        for a, b in zip(*res):
            torch.testing.assert_close(a.float(), b.float(), rtol=tol, atol=tol * float(b.float().abs().max()))
