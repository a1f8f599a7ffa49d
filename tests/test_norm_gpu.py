"""GPU parity of the fused residual-add + LayerNorm kernels (csrc/add_layernorm.hip) against a float32
PyTorch reference of the same op (floating-point kernel: torch fp32 is the checker).

Tolerances: outputs are bf16 -> one rounding (2^-8 relative) of a float32-exact result; statistics are
float32 (compared at 1e-5); parameter gradients are sums over all rows in float32, rounded once to bf16."""
import pytest
import torch
import torch.nn.functional as F

gpu = pytest.mark.gpu


def _inputs(rows, seed, with_b=True):
    g = torch.Generator().manual_seed(seed)
    a = (torch.randn(rows, 256, generator=g) * 1.5 + 0.3).to(torch.bfloat16)
    b = (torch.randn(rows, 256, generator=g) * 0.7).to(torch.bfloat16) if with_b else None
    w = (1 + 0.2 * torch.randn(256, generator=g)).to(torch.bfloat16)
    bias = (0.1 * torch.randn(256, generator=g)).to(torch.bfloat16)
    dy = torch.randn(rows, 256, generator=g).to(torch.bfloat16)
    return a, b, w, bias, dy


def _reference(a, b, w, bias, dy, eps):
    x = (a.double() + (0 if b is None else b.double())).requires_grad_(True)
    wd, bd = w.double().requires_grad_(True), bias.double().requires_grad_(True)
    y = F.layer_norm(x, (256,), wd, bd, eps)
    y.backward(dy.double())
    return y.detach(), x.grad, wd.grad, bd.grad, x.detach().mean(-1), x.detach().var(-1, unbiased=False)


@gpu
@pytest.mark.parametrize("rows", [1, 7, 8, 9, 4099, 88892])
@pytest.mark.parametrize("with_b", [True, False])
def test_add_layernorm_forward_backward(rows, with_b):
    from rlipv2_amd import norm
    eps = 1e-5
    a, b, w, bias, dy = _inputs(rows, rows + int(with_b), with_b)
    ref_y, ref_dx, ref_dw, ref_db, ref_mean, ref_var = _reference(a, b, w, bias, dy, eps)
    ac = a.cuda().requires_grad_(True)
    bc = b.cuda().requires_grad_(True) if with_b else None
    wc, biasc = w.cuda().requires_grad_(True), bias.cuda().requires_grad_(True)
    y = norm.AddLayerNormFunction.apply(ac, bc, wc, biasc, eps)
    y.backward(dy.cuda())
    tol = 2.0 ** -8
    assert y.dtype == torch.bfloat16
    assert ((y.double().cpu() - ref_y).abs() <= tol * ref_y.abs() + 1e-6).all()
    assert ((ac.grad.double().cpu() - ref_dx).abs() <= tol * ref_dx.abs() + 2e-3 * ref_dx.abs().max()).all()
    if with_b:
        assert torch.equal(bc.grad, ac.grad)
    scale_w = (dy.double().abs() * ((a.double() + (0 if b is None else b.double()) - ref_mean[:, None])
                                    / (ref_var[:, None] + eps).sqrt()).abs()).sum(0)
    assert ((wc.grad.double().cpu() - ref_dw).abs() <= tol * ref_dw.abs() + 1e-5 * scale_w + 1e-6).all()
    assert ((biasc.grad.double().cpu() - ref_db).abs() <= tol * ref_db.abs() + 1e-5 * dy.double().abs().sum(0) + 1e-6).all()


@gpu
def test_add_layer_norm_module_route_and_fallback():
    """encoder-sized bf16 input takes the fused kernel; a small or float32 input takes PyTorch's ops; both
    agree with the plain module."""
    from rlipv2_amd import norm
    ln = torch.nn.LayerNorm(256).cuda().to(torch.bfloat16)
    with torch.no_grad():
        ln.weight.uniform_(0.5, 1.5)
        ln.bias.uniform_(-0.2, 0.2)
    g = torch.Generator().manual_seed(1)
    big_a = torch.randn(2, 5000, 256, generator=g).to(torch.bfloat16).cuda()
    big_b = torch.randn(2, 5000, 256, generator=g).to(torch.bfloat16).cuda()
    assert norm.supported(big_a, big_b, ln.weight, ln.bias)
    fused = norm.add_layer_norm(big_a, big_b, ln)
    plain = ln((big_a.float() + big_b.float()).to(torch.bfloat16))
    torch.testing.assert_close(fused.float(), plain.float(), rtol=2e-2, atol=3e-2)
    small = big_a[:, :10]
    assert not norm.supported(small, None, ln.weight, ln.bias)
    torch.testing.assert_close(norm.add_layer_norm(small, None, ln), ln(small))
    ln32 = torch.nn.LayerNorm(256).cuda()
    assert not norm.supported(big_a.float(), None, ln32.weight, ln32.bias)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        norm.AddLayerNormFunction.apply(torch.zeros(8, 256), None, torch.ones(256), torch.zeros(256), 1e-5)


@gpu
def test_add_relu_kernel_is_bit_identical_to_add_then_relu():
    """csrc/elementwise.hip: relu(a + b) in one pass (the ResNet bottleneck tail) -- same bits as the two PyTorch ops in
    bfloat16 (the sum is rounded before the clamp), for contiguous and channels-last tensors, and the same gradients."""
    import torch
    from rlipv2_amd import backbone
    g = torch.Generator(device="cuda:0").manual_seed(0)
    for fmt in (torch.contiguous_format, torch.channels_last):
        a = torch.randn(2, 64, 25, 42, device="cuda:0", generator=g).bfloat16().contiguous(memory_format=fmt).requires_grad_(True)
        b = torch.randn(2, 64, 25, 42, device="cuda:0", generator=g).bfloat16().contiguous(memory_format=fmt).requires_grad_(True)
        dy = torch.randn(2, 64, 25, 42, device="cuda:0", generator=g).bfloat16().contiguous(memory_format=fmt)
        y = backbone.add_relu(a, b)
        assert isinstance(y.grad_fn, backbone.AddReLUFunction._backward_cls)
        y.backward(dy)
        ga, gb = a.grad.clone(), b.grad.clone()
        a.grad = b.grad = None
        ref = torch.relu(a + b)
        ref.backward(dy)
        torch.cuda.synchronize()
        assert torch.equal(y, ref) and y.stride() == ref.stride()
        assert torch.equal(ga, a.grad) and torch.equal(gb, b.grad)
    # shapes the kernel does not take fall back to the two ops
    c = torch.randn(3, 5, device="cuda:0").bfloat16()
    assert torch.equal(backbone.add_relu(c, c), torch.relu(c + c))


@gpu
def test_affine_relu_kernels_match_frozen_batchnorm_then_relu():
    """csrc/elementwise.hip: relu(bn(x)) for a FrozenBatchNorm2d as one pass (and its backward as one pass): the same
    bits as addcmul + relu forward, the same gradient as autograd's threshold_backward + mul to bf16 rounding."""
    import torch
    from rlipv2_amd import backbone
    g = torch.Generator(device="cuda:0").manual_seed(1)
    bn = backbone.FrozenBatchNorm2d(64).to("cuda:0")
    bn.weight.copy_(torch.rand(64, device="cuda:0", generator=g) + 0.5)
    bn.bias.copy_(torch.randn(64, device="cuda:0", generator=g))
    bn.running_mean.copy_(torch.randn(64, device="cuda:0", generator=g))
    bn.running_var.copy_(torch.rand(64, device="cuda:0", generator=g) + 0.3)
    bn = bn.to(torch.bfloat16)
    x = torch.randn(2, 64, 25, 42, device="cuda:0", generator=g).bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    dy = torch.randn(2, 64, 25, 42, device="cuda:0", generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
    y = backbone.bn_relu(x, bn)
    assert isinstance(y.grad_fn, backbone.AffineReLUFunction._backward_cls)
    y.backward(dy)
    gx = x.grad.clone()
    x.grad = None
    ref = torch.relu(bn(x))
    ref.backward(dy)
    torch.cuda.synchronize()
    assert torch.equal(y, ref)
    assert torch.equal(gx, x.grad)


# ---- GroupNorm(32, 256) of the token-major pyramid (csrc/groupnorm_tokens.hip) against torch.nn.functional.group_norm in
# float64 on the same bf16 inputs.  Tolerances: outputs / dx are bf16 (one rounding, 2^-8 relative, of a float32 result);
# dgamma / dbeta are sums over N * H * W terms in float32, rounded once to bf16.
def _gn_case(N, hw, seed):
    g = torch.Generator().manual_seed(seed)
    xs = [(torch.randn(N, t, 256, generator=g) * (1 + 0.5 * l) + 0.2 * l).to(torch.bfloat16) for l, t in enumerate(hw)]
    gam = [(1 + 0.2 * torch.randn(256, generator=g)).to(torch.bfloat16) for _ in hw]
    bet = [(0.1 * torch.randn(256, generator=g)).to(torch.bfloat16) for _ in hw]
    dy = torch.randn(N, sum(hw), 256, generator=g).to(torch.bfloat16)
    return xs, gam, bet, dy


@gpu
@pytest.mark.parametrize("N,hw", [(1, [1]), (2, [300, 77, 20, 6]), (3, [256, 257]), (4, [16700, 4200, 1050, 273])])
def test_level_group_norm_matches_float64_group_norm(N, hw):
    from rlipv2_amd import norm
    xs, gam, bet, dy = _gn_case(N, hw, sum(hw) + N)
    eps = 1e-5
    mods = []
    for g_, b_ in zip(gam, bet):
        m = torch.nn.GroupNorm(32, 256, eps=eps).cuda().to(torch.bfloat16)
        m.weight.data.copy_(g_); m.bias.data.copy_(b_)
        mods.append(m)
    xc = [x.cuda().requires_grad_(True) for x in xs]
    assert norm.level_group_norm_supported(xc, mods)
    out = norm.level_group_norm(xc, mods)
    out.backward(dy.cuda())
    out2 = norm.level_group_norm([x.detach() for x in xc], mods)
    assert torch.equal(out, out2)                                               # repeatable bit for bit
    start = 0
    for l, t in enumerate(hw):
        xd = xs[l].double().requires_grad_(True)
        gd, bd = gam[l].double().requires_grad_(True), bet[l].double().requires_grad_(True)
        # [N, t, 256] token-major -> [N, 256, t]: group statistics over t x 8 channels
        ref = F.group_norm(xd.transpose(1, 2), 32, gd, bd, eps).transpose(1, 2)
        ref.backward(dy[:, start:start + t].double())
        got = out[:, start:start + t].float().cpu()
        assert ((got - ref.detach()).abs() <= 2.0 ** -7 * ref.detach().abs() + 1e-3).all(), l
        dx = xc[l].grad.float().cpu()
        scale = float(xd.grad.abs().max())
        assert ((dx - xd.grad).abs() <= 2.0 ** -7 * xd.grad.abs() + 2e-3 * scale).all(), l
        for got_p, ref_p in ((mods[l].weight.grad, gd.grad), (mods[l].bias.grad, bd.grad)):
            tol = 2.0 ** -7 * ref_p.abs() + 1e-5 * (N * t) ** 0.5 * float(dy.abs().max()) * 4 + 1e-3
            assert ((got_p.float().cpu() - ref_p).abs() <= tol).all(), l
        start += t
