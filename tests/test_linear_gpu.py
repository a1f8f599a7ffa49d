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
@pytest.mark.parametrize("T,M,K", [(1, 128, 128), (31, 128, 128), (32, 128, 128), (320, 768, 768), (256, 768, 3072),
                                   (480, 384, 128), (33, 256, 128), (1000, 128, 256),
                                   (600, 256, 256), (2048, 384, 256), (2049, 256, 384), (2079, 128, 128),
                                   (4099, 384, 256), (22223, 256, 256), (88892, 256, 256), (88892, 1024, 256),
                                   (88892, 256, 1024), (88892, 384, 256),
                                   # bias gradient shared by 8 / 4 / 4 workgroups of a row of tiles (K / 128 >= 2)
                                   (88892, 2048, 256), (88892, 256, 2048), (32769, 1024, 512), (40000, 768, 768)])
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
    # many chunks, 6 tiles per row: the bias gradient's column sums are split over 4 workgroups per row
    T, M, K = 33001, 768, 768
    dy = torch.randint(-2, 3, (T, M), generator=g)
    x = torch.randint(-2, 3, (T, K), generator=g)
    dw, db = linear.linear_wgrad(dy.to(torch.bfloat16).cuda(), x.to(torch.bfloat16).cuda(), out_dtype=torch.float32)
    assert torch.equal(dw.cpu().long(), dy.t() @ x)
    assert torch.equal(db.cpu().long(), dy.sum(0))


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


@gpu
@pytest.mark.parametrize("rows", [600, 9000])
def test_relu_epilogue_matches_linear_then_relu(rows):
    """token_linear(..., relu=True): bias + ReLU in the GEMM epilogue, ReLU mask applied in backward; both the
    library route (few rows) and the MFMA weight-gradient route (many rows)."""
    from rlipv2_amd import linear
    g = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, 256, generator=g).to(torch.bfloat16).cuda()
    w = (torch.randn(384, 256, generator=g) * 0.06).to(torch.bfloat16).cuda()
    b = (torch.randn(384, generator=g) * 0.1).to(torch.bfloat16).cuda()
    dy = torch.randn(rows, 384, generator=g).to(torch.bfloat16).cuda()
    res = []
    for fused in (True, False):
        xx, ww, bb = (t.clone().requires_grad_(True) for t in (x, w, b))
        y = linear.token_linear(xx, ww, bb, relu=True) if fused else torch.relu(torch.nn.functional.linear(xx, ww, bb))
        y.backward(dy)
        res.append((y.detach(), xx.grad, ww.grad, bb.grad))
    assert (res[0][0] >= 0).all()
    for a, r in zip(*res):
        torch.testing.assert_close(a.float(), r.float(), rtol=2e-2, atol=2e-2 * float(r.float().abs().max()))


@gpu
def test_backbone_pointwise_gemm_route_matches_convolution_route():
    """Bottleneck with the 1x1 convolutions + frozen BN (+ReLU) as token-major GEMMs against the same block on
    MIOpen convolutions + addcmul: outputs and every gradient, bf16 channels-last."""
    from rlipv2_amd import backbone
    torch.manual_seed(0)
    down = torch.nn.Sequential(torch.nn.Conv2d(256, 512, 1, stride=2, bias=False), backbone.FrozenBatchNorm2d(512))
    blocks = torch.nn.Sequential(backbone.Bottleneck(256, 128, 2, down), backbone.Bottleneck(512, 128)).cuda()
    with torch.no_grad():
        for m in blocks.modules():
            if isinstance(m, backbone.FrozenBatchNorm2d):
                m.weight.uniform_(0.5, 1.5); m.bias.uniform_(-0.2, 0.2)
                m.running_mean.uniform_(-0.1, 0.1); m.running_var.uniform_(0.5, 1.5)
            if isinstance(m, torch.nn.Conv2d):
                m.weight.mul_(1.5)
    blocks.to(torch.bfloat16).to(memory_format=torch.channels_last)
    x = torch.randn(2, 256, 96, 120, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    res = []
    for flag in (True, False):
        backbone.pointwise_as_gemm = flag
        try:
            for p in blocks.parameters():
                p.grad = None
            xx = x.clone().requires_grad_(True)
            y = blocks(xx)
            y.float().square().mean().backward()
            res.append([y.detach(), xx.grad] + [p.grad.clone() for p in blocks.parameters()])
        finally:
            backbone.pointwise_as_gemm = True
    # the two routes round differently in bf16 (BN scale folded into the weights vs applied to the rounded
    # convolution output) and a few ReLUs flip: compare in the L2 sense, with a loose element-wise bound
    for a, r in zip(*res):
        assert a.shape == r.shape
        a, r = a.float(), r.float()
        assert float((a - r).norm() / r.norm()) < 5e-2
        assert float((a - r).abs().max()) < 0.25 * float(r.abs().max())


# ---- expand GEMM (csrc/expand_gemm.hip): C = epilogue(A[T,256] B[N,256]^T) ---------------------------------------------
def _expand_ref(a, b, bias, mask, relu):
    c = a.float() @ b.float().t()
    if bias is not None:
        c = c + bias.float()
    if relu:
        c = c.relu()
    c = c.to(torch.bfloat16)
    if mask is not None:
        c = torch.where(mask > 0, c, torch.zeros_like(c))
    return c


@gpu
@pytest.mark.parametrize("T", [1, 63, 128, 129, 4097, 88892])
@pytest.mark.parametrize("N", [64, 256, 2048])
@pytest.mark.parametrize("mode", ["plain", "bias_relu", "mask", "bias_mask"])
def test_expand_gemm_matches_float32_reference(T, N, mode):
    from rlipv2_amd import linear
    if T == 88892 and N != 2048:
        pytest.skip("full-size case runs at the FFN width only")
    g = torch.Generator(device="cuda").manual_seed(T * 7 + N)
    a = torch.randn(T, 256, device="cuda", generator=g).to(torch.bfloat16)
    b = (torch.randn(N, 256, device="cuda", generator=g) / 16).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g).to(torch.bfloat16) if "bias" in mode else None
    mask = torch.randn(T, N, device="cuda", generator=g).relu().to(torch.bfloat16) if "mask" in mode else None
    relu = mode == "bias_relu"
    c = linear.expand_gemm(a, b, bias=bias, mask=mask, relu=relu)
    ref = _expand_ref(a, b, bias, mask, relu)
    assert c.shape == (T, N) and c.dtype == torch.bfloat16
    # float32 accumulation in a different order, one bf16 rounding: at most one bf16 ulp apart
    err = (c.float() - ref.float()).abs()
    tol = 2.0 ** -7 * ref.float().abs() + 1e-3
    assert bool((err <= tol).all()), float((err - tol).max())
    if mask is not None:
        assert bool((c[mask <= 0] == 0).all())
    if relu:
        assert bool((c >= 0).all())


@gpu
def test_expand_gemm_rejects_what_it_cannot_do():
    from rlipv2_amd import linear
    a = torch.randn(8, 128, device="cuda").to(torch.bfloat16)
    b = torch.randn(64, 128, device="cuda").to(torch.bfloat16)
    with pytest.raises(RuntimeError):
        linear.expand_gemm(a, b)                                   # K != 256
    a = torch.randn(8, 256, device="cuda").to(torch.bfloat16)
    with pytest.raises(RuntimeError):
        linear.expand_gemm(a, torch.randn(65, 256, device="cuda").to(torch.bfloat16))   # N % 64
    with pytest.raises(RuntimeError):
        linear.expand_gemm(a.cpu(), b.cpu())


@gpu
def test_fused_ffn_gradients_match_the_unfused_path():
    from rlipv2_amd import linear
    torch.manual_seed(3)
    T = 4 * 2222
    lin1 = torch.nn.Linear(256, 1024).cuda().to(torch.bfloat16)
    lin2 = torch.nn.Linear(1024, 256).cuda().to(torch.bfloat16)
    x = torch.randn(4, T // 4, 256, device="cuda").to(torch.bfloat16).requires_grad_()
    dy = torch.randn(4, T // 4, 256, device="cuda").to(torch.bfloat16)

    def run(fn):
        for p in (*lin1.parameters(), *lin2.parameters(), x):
            p.grad = None
        y = fn()
        y.backward(dy)
        return [y.detach().float()] + [p.grad.float() for p in (x, *lin1.parameters(), *lin2.parameters())]

    fused = run(lambda: linear.fused_ffn(x, lin1, lin2))
    plain = run(lambda: torch.nn.functional.linear(torch.relu(torch.nn.functional.linear(x, lin1.weight, lin1.bias)),
                                                   lin2.weight, lin2.bias))
    for f, p in zip(fused, plain):
        rel = (f - p).norm() / p.norm().clamp_min(1e-6)
        assert float(rel) < 2e-2, float(rel)


@gpu
def test_add_row_vector_gradient_matches_broadcast_add():
    from rlipv2_amd import linear
    torch.manual_seed(5)
    full = torch.randn(4, 3000, 256, device="cuda").to(torch.bfloat16)
    x = full[:, 500:2500]                                  # a strided slice, as the per-level pieces of the pyramid are
    row = torch.randn(256, device="cuda").to(torch.bfloat16).requires_grad_()
    g = torch.randn(4, 3000, 256, device="cuda").to(torch.bfloat16)[:, 500:2500]
    y = linear.add_row_vector(x, row)
    assert torch.equal(y, x + row.view(1, 1, -1))
    y.backward(g)
    ref = g.float().sum((0, 1))
    assert float((row.grad.float() - ref).abs().max()) <= 2e-2 * float(ref.abs().max())


@gpu
@pytest.mark.first_contact(timeout=300)          # step_scaled_kernel has never run on hardware: isolated, XPASS / XFAIL (tests/conftest.py)
def test_fused_adamw_grad_scale_equals_scaling_the_gradients_first():
    """adamw_step_scaled_bf16 (include/rlipv2_optim.h): the data-parallel step leaves the all-reduced gradient SUM in the
    flat buffer and hands 1 / world to the optimiser's kernels.  Same update as scaling the gradients in float32 first
    (to the bf16 rounding of the parameters); clipping sees the scaled norm."""
    from rlipv2_amd import train
    torch.manual_seed(3)
    world = 8

    def make():
        torch.manual_seed(3)
        m = torch.nn.Sequential(torch.nn.Linear(256, 384), torch.nn.Linear(384, 128)).cuda().to(torch.bfloat16)
        return m, train.FusedMasterAdamW(m)

    grads = [torch.randn(384, 256), torch.randn(384), torch.randn(128, 384), torch.randn(128)]
    for max_norm in (0.1, 0.0):
        m1, o1 = make()
        m2, o2 = make()
        for _ in range(3):
            for p, g in zip(m1.parameters(), grads):
                p.grad = (g * world).cuda().to(torch.bfloat16)                       # the SUM over 8 equal ranks
            for p, g in zip(m2.parameters(), grads):
                p.grad = ((g * world).cuda().to(torch.bfloat16).float() / world).to(torch.bfloat16)
            o1.step(max_norm, grad_scale=1.0 / world)
            o2.step(max_norm)
        for p1, p2 in zip(m1.parameters(), m2.parameters()):
            torch.testing.assert_close(p1.float(), p2.float(), rtol=0, atol=2e-2 * float(p2.float().abs().max()))
        for a, b in zip(o1.master, o2.master):
            torch.testing.assert_close(a, b, rtol=2e-2, atol=1e-5)
