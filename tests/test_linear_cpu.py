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
