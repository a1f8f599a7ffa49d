"""Timing of the fused add+LayerNorm kernels against PyTorch's add + layer_norm (profiling aid)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlipv2_amd import norm
T = 4 * 22223
a = torch.randn(T, 256, device="cuda", dtype=torch.bfloat16, requires_grad=True)
b = torch.randn(T, 256, device="cuda", dtype=torch.bfloat16, requires_grad=True)
ln = torch.nn.LayerNorm(256).cuda().to(torch.bfloat16)
dy = torch.randn(T, 256, device="cuda", dtype=torch.bfloat16)


def t_us(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for name, fn in (("fused", lambda: norm.add_layer_norm(a, b, ln)), ("torch", lambda: ln(a + b))):
    fwd = t_us(lambda: fn())
    y = fn()
    both = t_us(lambda: torch.autograd.grad(fn(), (a, b, ln.weight, ln.bias), dy))
    print(f"{name}: forward {fwd:6.1f} us   forward+backward {both:6.1f} us   "
          f"(HBM-bound: fwd {3 * T * 512 / 8e12 * 1e6:.1f} us, bwd {4 * T * 512 / 8e12 * 1e6:.1f} us)")
