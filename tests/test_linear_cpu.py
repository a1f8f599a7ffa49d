"""Host logic of rlipv2_amd/linear.py that must hold without a GPU: the fused / custom-backward entry points step
aside for CPU tensors (plain PyTorch ops, same values and gradients) and the raw kernel wrappers refuse CPU tensors
like the reference's extension does ("Not implemented on the CPU", ms_deform_attn.h:31)."""
import pytest
import torch

from rlipv2_amd import linear


def test_fused_ffn_on_cpu_is_the_plain_composition():
    torch.manual_seed(0)
    l1, l2 = torch.nn.Linear(256, 512), torch.nn.Linear(512, 256)
    x = torch.randn(2, 40, 256, requires_grad=True)
    y = linear.fused_ffn(x, l1, l2)
    ref = l2(torch.relu(l1(x)))
    torch.testing.assert_close(y, ref)
    g, = torch.autograd.grad(y.sum(), x)
    gr, = torch.autograd.grad(ref.sum(), x)
    torch.testing.assert_close(g, gr)


def test_add_row_vector_on_cpu_is_a_broadcast_add():
    x = torch.randn(2, 7, 16)
    row = torch.randn(16, requires_grad=True)
    y = linear.add_row_vector(x, row)
    torch.testing.assert_close(y, x + row.view(1, 1, -1))
    y.sum().backward()
    torch.testing.assert_close(row.grad, torch.full((16,), 14.0))


def test_add_row_vector_function_gradient():
    x = torch.randn(3, 5, 8)
    row = torch.randn(8, requires_grad=True)
    g = torch.randn(3, 5, 8)
    linear.AddRowVectorFunction.apply(x, row).backward(g)
    torch.testing.assert_close(row.grad, g.sum((0, 1)))


def test_kernel_wrappers_refuse_cpu_tensors():
    a = torch.randn(8, 256).to(torch.bfloat16)
    b = torch.randn(64, 256).to(torch.bfloat16)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        linear.expand_gemm(a, b)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        linear.linear_wgrad(a, a)


def test_tuned_table_is_wellformed():
    """every row of the committed hipBLASLt table: validator lines first, then op, signature, solution, time"""
    rows = [l.strip().split(",") for l in open(linear.TUNED_TABLE) if l.strip()]
    assert any(r[0] == "Validator" and r[1] == "GCN_ARCH_NAME" and r[2].startswith("gfx950") for r in rows)
    ops = [r for r in rows if r[0] != "Validator"]
    assert ops and all(len(r) == 4 and r[0].startswith("Gemm") and r[2].startswith("Gemm_") and float(r[3]) > 0
                       for r in ops)


def test_residual_gradient_accumulated_by_the_ffn_gemm_logic():
    """linear.ffn_residual_norm links the fused FFN node and the fused add + LayerNorm node: the LayerNorm's backward returns one
    tensor for both addends, the FFN's input-gradient GEMM accumulates into it in place and returns None.  The kernels behind
    the two nodes are GPU-only; here their calls are replaced by torch arithmetic (test-side stand-ins) so that the LOGIC --
    who accumulates into whom, what autograd finally hands to the layer input -- is checked against plain autograd."""
    import torch.nn.functional as F

    from rlipv2_amd import norm

    class TorchAddLayerNorm(torch.autograd.Function):          # the protocol of norm.AddLayerNormFunction in torch arithmetic
        @staticmethod
        def forward(ctx, a, b, weight, bias, eps, link=None):
            ctx.link = link
            x = a + b
            mean = x.mean(-1, keepdim=True)
            rstd = (x.var(-1, unbiased=False, keepdim=True) + eps).rsqrt()
            ctx.save_for_backward(x, weight, mean, rstd)
            return (x - mean) * rstd * weight + bias

        @staticmethod
        def backward(ctx, dy):
            x, weight, mean, rstd = ctx.saved_tensors
            xh = (x - mean) * rstd
            gd = dy * weight
            dx = rstd * (gd - gd.mean(-1, keepdim=True) - xh * (gd * xh).mean(-1, keepdim=True))
            if ctx.link is not None:
                ctx.link.dx = dx
            red = tuple(range(dy.dim() - 1))
            return dx, dx, (dy * xh).sum(red), dy.sum(red), None, None

    saved = (linear.linear_wgrad, linear.expand_gemm, linear._ffn_block_ok, norm.AddLayerNormFunction)
    linear.linear_wgrad = lambda dy, x, with_bias=True, out_dtype=None: (
        dy.reshape(-1, dy.shape[-1]).t() @ x.reshape(-1, x.shape[-1]), dy.reshape(-1, dy.shape[-1]).sum(0))
    linear.expand_gemm = lambda a, b, bias=None, mask=None, relu=False: (a @ b.t()) * (mask > 0)
    linear._ffn_block_ok = lambda *a: True
    norm.AddLayerNormFunction = TorchAddLayerNorm
    linear.residual_gradient_in_gemm = True                    # (off in the package until routes.validate has passed on a GPU)
    try:
        torch.manual_seed(0)
        l1, l2, ln = torch.nn.Linear(16, 64), torch.nn.Linear(64, 16), torch.nn.LayerNorm(16)
        x0 = torch.randn(3, 7, 16)
        w = torch.randn(3, 7, 16)
        res = []
        for fused in (True, False):
            for p in list(l1.parameters()) + list(l2.parameters()) + list(ln.parameters()):
                p.grad = None
            x = x0.clone().requires_grad_(True)
            src = x * 1.5                                       # a non-leaf input with one more consumer below
            y = linear.ffn_residual_norm(src, l1, l2, ln) if fused else ln(src + l2(F.relu(l1(src))))
            ((y * w).sum() + (src * 0.25).sum()).backward()
            res.append([x.grad.clone()] + [p.grad.clone() for p in list(l1.parameters()) + list(l2.parameters()) + list(ln.parameters())])
        for a, b in zip(*res):
            torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6)
        # the accumulation really happened in place: the FFN node returned no gradient of its own
        x = x0.clone().requires_grad_(True)
        y = linear.ffn_residual_norm(x * 1.0, l1, l2, ln)
        ffn_node = [n for n, _ in y.grad_fn.next_functions if n is not None and "FusedFFN" in type(n).__name__]
        assert len(ffn_node) == 1
        seen = {}
        ffn_node[0].register_hook(lambda gin, gout: seen.setdefault("gin", gin))
        (y * w).sum().backward()
        assert seen["gin"][0] is None
    finally:
        linear.linear_wgrad, linear.expand_gemm, linear._ffn_block_ok, norm.AddLayerNormFunction = saved
        linear.residual_gradient_in_gemm = False


def test_linked_attention_block_accumulates_into_the_layernorm_gradient():
    """The encoder layer's linked attention block (encoder.py): the value projection's input-gradient GEMM and the query's add
    accumulate into the tensor the LayerNorm's backward returned for the residual.  Kernels replaced by torch stand-ins (they
    are GPU-only); the composition mirrors DeformableTransformerEncoderLayer.forward with an elementwise mix standing for the
    sampling.  Gradients of the layer input, of `pos` and of every parameter against plain autograd."""
    import torch.nn.functional as F

    from rlipv2_amd import norm

    class TorchAddLayerNorm(torch.autograd.Function):
        @staticmethod
        def forward(ctx, a, b, weight, bias, eps, link=None):
            ctx.link = link
            x = a + b
            mean = x.mean(-1, keepdim=True)
            rstd = (x.var(-1, unbiased=False, keepdim=True) + eps).rsqrt()
            ctx.save_for_backward(x, weight, mean, rstd)
            return (x - mean) * rstd * weight + bias

        @staticmethod
        def backward(ctx, dy):
            x, weight, mean, rstd = ctx.saved_tensors
            xh = (x - mean) * rstd
            gd = dy * weight
            dx = rstd * (gd - gd.mean(-1, keepdim=True) - xh * (gd * xh).mean(-1, keepdim=True))
            if ctx.link is not None:
                ctx.link.dx = dx
            red = tuple(range(dy.dim() - 1))
            return dx, dx, (dy * xh).sum(red), dy.sum(red), None, None

    saved = linear.linear_wgrad
    linear.linear_wgrad = lambda dy, x, with_bias=True, out_dtype=None: (
        dy.reshape(-1, dy.shape[-1]).t() @ x.reshape(-1, x.shape[-1]), dy.reshape(-1, dy.shape[-1]).sum(0))
    try:
        torch.manual_seed(1)
        C = 16
        vp, qp, op, ln = torch.nn.Linear(C, C), torch.nn.Linear(C, 24), torch.nn.Linear(C, C), torch.nn.LayerNorm(C)
        mods = (vp, qp, op, ln)
        x0, pos0, w = torch.randn(2, 9, C), torch.randn(2, 9, C), torch.randn(2, 9, C)
        res = []
        for linked in (True, False):
            for m in mods:
                for p in m.parameters():
                    p.grad = None
            x = x0.clone().requires_grad_(True)
            pos = pos0.clone().requires_grad_(True)
            src = x * 0.5
            if linked:
                link = norm.GradLink()
                xa = linear._Alias.apply(src, link)
                q = linear._AddInto.apply(xa, pos, link)
                value = linear.TokenLinearFunction.apply(xa, vp.weight, vp.bias, False, link)
                qproj = linear.TokenLinearFunction.apply(q, qp.weight, qp.bias, False, None)
                attn = linear.TokenLinearFunction.apply(value * torch.sigmoid(qproj[..., :C]), op.weight, op.bias, False, None)
                y = TorchAddLayerNorm.apply(xa, attn, ln.weight, ln.bias, ln.eps, link)
            else:
                value = vp(src)
                qproj = qp(src + pos)
                y = ln(src + op(value * torch.sigmoid(qproj[..., :C])))
            ((y * w).sum() + (src * 0.1).sum()).backward()
            res.append([x.grad.clone(), pos.grad.clone()] + [p.grad.clone() for m in mods for p in m.parameters()])
            if linked:
                assert link.dx is None                           # dropped by the alias node, the last of the block
        for a, b in zip(*res):
            torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6)
    finally:
        linear.linear_wgrad = saved


def test_shared_input_of_several_linears_is_summed_by_their_gemms():
    """linear.shared_input's protocol (the image memory under the six decoder layers' value projections): the first Linear's
    input gradient becomes the accumulator, the others add into it, the alias node hands the sum on.  torch stand-in for the
    weight-gradient kernel; against plain autograd."""
    from rlipv2_amd import norm
    saved = linear.linear_wgrad
    linear.linear_wgrad = lambda dy, x, with_bias=True, out_dtype=None: (
        dy.reshape(-1, dy.shape[-1]).t() @ x.reshape(-1, x.shape[-1]), dy.reshape(-1, dy.shape[-1]).sum(0))
    try:
        torch.manual_seed(2)
        lins = [torch.nn.Linear(12, 12) for _ in range(4)]
        x0 = torch.randn(2, 5, 12)
        ws = [torch.randn(2, 5, 12) for _ in lins]
        res = []
        for linked in (True, False):
            for m in lins:
                m.weight.grad = m.bias.grad = None
            x = x0.clone().requires_grad_(True)
            src = torch.tanh(x)
            if linked:
                link = norm.GradLink()
                link.first_creates = True
                xa = linear._Alias.apply(src, link)
                outs = [linear.TokenLinearFunction.apply(xa, m.weight, m.bias, False, link) for m in lins]
            else:
                outs = [m(src) for m in lins]
            sum((o * w).sum() for o, w in zip(outs, ws)).backward()
            res.append([x.grad.clone()] + [p.grad.clone() for m in lins for p in m.parameters()])
        for a, b in zip(*res):
            torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6)
    finally:
        linear.linear_wgrad = saved


@pytest.mark.parametrize("case", ["noncontiguous_dy", "unusable_accumulator"])
def test_shared_input_never_mixes_in_place_and_returned_gradients(case):
    """Advisor finding of round 4: once a link holds an accumulator, a consumer that returned its input gradient the normal way
    made autograd sum out of place, and everything later nodes added into the (now stale) accumulator was lost without an
    error.  `noncontiguous_dy`: one of four consumers is fed a transposed gradient -- it must still accumulate in place.
    `unusable_accumulator`: a consumer whose accumulator cannot be used (dtype differs) breaks the link, and every later node
    returns its gradient normally.  Either way the sum equals plain autograd's."""
    from rlipv2_amd import norm
    saved = linear.linear_wgrad
    linear.linear_wgrad = lambda dy, x, with_bias=True, out_dtype=None: (
        dy.reshape(-1, dy.shape[-1]).t() @ x.reshape(-1, x.shape[-1]), dy.reshape(-1, dy.shape[-1]).sum(0))
    try:
        torch.manual_seed(3)
        lins = [torch.nn.Linear(12, 12) for _ in range(4)]
        x0 = torch.randn(5, 5, 12)
        ws = [torch.randn(5, 5, 12) for _ in lins]
        res = []
        for linked in (True, False):
            for m in lins:
                m.weight.grad = m.bias.grad = None
            x = x0.clone().requires_grad_(True)
            src = torch.tanh(x)
            link = None
            if linked:
                link = norm.GradLink()
                link.first_creates = True
                xa = linear._Alias.apply(src, link)
                outs = [linear.TokenLinearFunction.apply(xa, m.weight, m.bias, False, link) for m in lins]
            else:
                outs = [m(src) for m in lins]
            terms = [(o * w).sum() for o, w in zip(outs, ws)]
            if case == "noncontiguous_dy":
                # consumer 1's incoming gradient is a transposed view (runs second-to-last in the backward)
                terms[1] = (outs[1].transpose(0, 1) * ws[1].transpose(0, 1).contiguous()).sum()
            if case == "unusable_accumulator" and linked:
                # swap the accumulator's dtype just before consumer 1 runs (hook on consumer 2's output gradient)
                def spoil(g):
                    link.dx = link.dx.double()
                    return g
                outs[2].register_hook(spoil)
            sum(terms).backward()
            if linked and case == "unusable_accumulator":
                assert link.broken
            res.append([x.grad.clone()] + [p.grad.clone() for m in lins for p in m.parameters()])
        for a, b in zip(*res):
            torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6)
    finally:
        linear.linear_wgrad = saved


def test_weight_gradient_of_64_multiples_by_padding():
    """linear.pad_wgrad_to_128 (Swin stage 0: 192 / 576 channels): the narrow operand is zero-padded to the kernel's 128-multiples
    and the result is the top-left block.  Logic on the CPU with a torch stand-in for the kernel call."""
    calls = []

    def stand_in(dy, x, with_bias, out_dtype):
        calls.append((tuple(dy.shape), tuple(x.shape)))
        assert dy.shape[1] % 128 == 0 and x.shape[1] % 128 == 0 and dy.is_contiguous() and x.is_contiguous()
        return (dy.float().t() @ x.float()).to(out_dtype), (dy.float().sum(0).to(out_dtype) if with_bias else None)
    saved = linear._wgrad_call
    linear._wgrad_call = stand_in
    try:
        torch.manual_seed(0)
        for M, K in ((192, 192), (576, 192), (768, 192), (192, 768), (256, 128)):
            dy, x = torch.randn(300, M).to(torch.bfloat16), torch.randn(300, K).to(torch.bfloat16)
            want_w, want_b = dy.float().t() @ x.float(), dy.float().sum(0)
            for flag in (True, False):
                linear.pad_wgrad_to_128 = flag
                if not flag and (M % 128 or K % 128):
                    with pytest.raises(AssertionError):
                        linear._wgrad_maybe_padded(dy, x, True, torch.float32)
                    continue
                dw, db = linear._wgrad_maybe_padded(dy, x, True, torch.float32)
                assert dw.shape == (M, K) and db.shape == (M,) and dw.is_contiguous()
                torch.testing.assert_close(dw, want_w, rtol=1e-5, atol=1e-4)
                torch.testing.assert_close(db, want_b, rtol=1e-5, atol=1e-4)
        assert ((300, 256), (300, 256)) in calls and ((300, 640), (300, 256)) in calls
    finally:
        linear._wgrad_call = saved
        linear.pad_wgrad_to_128 = False
