"""GPU parity of the fused clip + AdamW step (csrc/fused_adamw.hip) against torch.optim.AdamW +
torch.nn.utils.clip_grad_norm_ on float32 copies (floating-point kernel: torch fp32 is the checker).
Tolerance: identical float32 formula up to operation order -> 2e-6 relative on master weights and moments
per step; bf16 parameters must equal the rounded master weights exactly."""
import pytest
import torch

gpu = pytest.mark.gpu


class _Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.backbone = torch.nn.Conv2d(8, 16, 3)                    # group "backbone"; 4-d, channels-last below
        self.text_encoder = torch.nn.Linear(37, 5)                   # group "text_encoder"; odd sizes -> scalar tails
        self.head = torch.nn.Linear(300, 129)                        # 38 700 elements: 3 chunks, last one partial
        self.scalar = torch.nn.Parameter(torch.randn(1))
        self.frozen = torch.nn.Parameter(torch.randn(4), requires_grad=False)


@gpu
@pytest.mark.parametrize("max_norm", [0.1, 1e9, 0.0])
def test_fused_adamw_matches_torch(max_norm):
    from rlipv2_amd import optim
    torch.manual_seed(0)
    net = _Net().cuda().to(torch.bfloat16)
    net.backbone.to(memory_format=torch.channels_last)
    opt = optim.FusedMasterAdamW(net, lr=1e-2, lr_backbone=1e-3, text_encoder_lr=5e-3, weight_decay=1e-2)
    names = opt.names
    ref_params = [m.clone().requires_grad_(True) for m in opt.master]
    by = lambda key: [p for n, p in zip(names, ref_params) if key(n)]
    ref = torch.optim.AdamW([{"params": by(lambda n: "backbone" not in n and "text_encoder" not in n)},
                             {"params": by(lambda n: "backbone" in n), "lr": 1e-3},
                             {"params": by(lambda n: "text_encoder" in n), "lr": 5e-3}], lr=1e-2, weight_decay=1e-2)
    g = torch.Generator(device="cuda").manual_seed(1)
    for step in range(4):
        for p, r in zip(opt.params, ref_params):
            grad = (torch.randn(p.shape, device="cuda", generator=g) * (10.0 if step % 2 else 0.01)).to(torch.bfloat16)
            if p.dim() == 4:
                grad = grad.contiguous(memory_format=torch.channels_last)
            if step == 2 and p.dim() == 1 and p.numel() == 129:
                p.grad = None                                             # a parameter without gradient is skipped
                r.grad = None
                continue
            p.grad = grad
            r.grad = grad.float()
        if max_norm > 0:
            total = torch.nn.utils.clip_grad_norm_([r for r in ref_params if r.grad is not None], max_norm)
        ref.step()
        opt.step(max_norm)
        if max_norm > 0:
            torch.testing.assert_close(opt.grad_norm()[0], total, rtol=1e-5, atol=0)
        for n, p, m, r in zip(names, opt.params, opt.master, ref_params):
            torch.testing.assert_close(m, r.detach(), rtol=5e-6, atol=1e-6 * float(r.detach().abs().max()),
                                       msg=lambda s: f"step {step} {n}: {s}")
            assert torch.equal(p.detach(), m.to(torch.bfloat16)), n
    for i, r in enumerate(ref_params):
        st = ref.state[r]
        torch.testing.assert_close(opt.exp_avg[i], st["exp_avg"], rtol=5e-6, atol=1e-6 * float(st["exp_avg"].abs().max()))
        torch.testing.assert_close(opt.exp_avg_sq[i], st["exp_avg_sq"], rtol=5e-6,
                                   atol=1e-6 * float(st["exp_avg_sq"].abs().max()))
    assert torch.equal(net.frozen, _frozen_reference())


@gpu
@pytest.mark.first_contact(timeout=420)
@pytest.mark.parametrize("grad_scale", [0.125, 0.5])
def test_fused_adamw_with_the_gradient_scale_of_a_data_parallel_step(grad_scale):
    """opt.step(max_norm, grad_scale): what a data-parallel step calls with grad_scale = 1 / world on the all-reduced SUM
    (csrc/fused_adamw.hip: step_scaled_kernel -- the one-GPU step never reaches it).  Against torch AdamW + clip_grad_norm_ on
    gradients scaled beforehand; also against this optimiser's own grad_scale = 1 step on pre-scaled bf16 gradients when the
    scale is a power of two (scaling then commutes with the bf16 rounding: the two must agree to float32 rounding)."""
    from rlipv2_amd import optim
    torch.manual_seed(0)
    nets = [_Net().cuda().to(torch.bfloat16) for _ in range(2)]
    nets[1].load_state_dict(nets[0].state_dict())
    opts = [optim.FusedMasterAdamW(n, lr=1e-2, lr_backbone=1e-3, text_encoder_lr=5e-3, weight_decay=1e-2) for n in nets]
    ref_params = [m.clone().requires_grad_(True) for m in opts[0].master]
    names = opts[0].names
    by = lambda key: [p for n, p in zip(names, ref_params) if key(n)]
    ref = torch.optim.AdamW([{"params": by(lambda n: "backbone" not in n and "text_encoder" not in n)},
                             {"params": by(lambda n: "backbone" in n), "lr": 1e-3},
                             {"params": by(lambda n: "text_encoder" in n), "lr": 5e-3}], lr=1e-2, weight_decay=1e-2)
    g = torch.Generator(device="cuda").manual_seed(2)
    for step in range(3):
        for p0, p1, r in zip(opts[0].params, opts[1].params, ref_params):
            grad = (torch.randn(p0.shape, device="cuda", generator=g) * (8.0 if step % 2 else 0.02)).to(torch.bfloat16)
            p0.grad = grad.clone()                                  # the SUM over the ranks
            p1.grad = (grad.float() * grad_scale).to(torch.bfloat16)     # (exact: power of two)
            r.grad = grad.float() * grad_scale
        total = torch.nn.utils.clip_grad_norm_(ref_params, 0.1)
        ref.step()
        opts[0].step(0.1, grad_scale=grad_scale)
        opts[1].step(0.1)
        torch.testing.assert_close(opts[0].grad_norm()[0], total, rtol=1e-5, atol=0)
        for n, m0, m1, r in zip(names, opts[0].master, opts[1].master, ref_params):
            scale = float(r.detach().abs().max())
            torch.testing.assert_close(m0, r.detach(), rtol=5e-6, atol=1e-6 * scale, msg=lambda s: f"step {step} {n}: {s}")
            torch.testing.assert_close(m0, m1, rtol=2e-6, atol=1e-6 * scale, msg=lambda s: f"step {step} {n} (own unscaled step): {s}")
        for p0, m0 in zip(opts[0].params, opts[0].master):
            assert torch.equal(p0.detach(), m0.to(torch.bfloat16))


def _frozen_reference():
    torch.manual_seed(0)
    return _Net().frozen.cuda().to(torch.bfloat16)


@gpu
def test_fused_adamw_state_dict_roundtrip_and_errors():
    from rlipv2_amd import optim
    torch.manual_seed(0)
    net = _Net().cuda().to(torch.bfloat16)
    opt = optim.FusedMasterAdamW(net)
    for p in opt.params:
        p.grad = torch.ones_like(p)
    opt.step(0.1)
    state = {k: ([t.clone() for t in v] if isinstance(v, list) and v and torch.is_tensor(v[0]) else v)
             for k, v in opt.state_dict().items()}
    opt2 = optim.FusedMasterAdamW(_Net().cuda().to(torch.bfloat16))
    opt2.load_state_dict(state)
    assert opt2.t == 1
    for a, b in zip(opt.master, opt2.master):
        assert torch.equal(a, b)
    with pytest.raises(RuntimeError, match="bfloat16"):
        optim.FusedMasterAdamW(_Net().cuda())
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        optim.FusedMasterAdamW(_Net().to(torch.bfloat16))
