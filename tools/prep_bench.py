"""Timing + float64 check of the fused sampling-geometry kernels (csrc/msda_prep.hip) at the encoder shape."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from rlipv2_amd import msda


def t_us(fn, n=30):
    """eager launches between two events (no graph: the backward allocates)"""
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

N, S, M, L, P = 4, 22223, 8, 4, 4
shapes = torch.tensor([(100, 167), (50, 84), (25, 42), (13, 21)], dtype=torch.long, device="cuda")
for dt in (torch.bfloat16, torch.float32):
    for refdim in (2, 4):
        q = (torch.randn(N, S, M * L * P * 3, device="cuda") * 0.5).to(dt).requires_grad_()
        ref = torch.rand(N, S, L, refdim, device="cuda")
        loc, aw = msda.SamplingGeometryFunction.apply(q, ref, shapes, M, L, P)
        g_loc, g_aw = torch.randn_like(loc), torch.randn_like(aw)
        fwd = t_us(lambda: msda.SamplingGeometryFunction.apply(q.detach(), ref, shapes, M, L, P))

        def both():
            a, b = msda.SamplingGeometryFunction.apply(q, ref, shapes, M, L, P)
            torch.autograd.grad((a, b), q, (g_loc, g_aw))
        fb = t_us(both)
        # reference in float64
        qd = q.detach().double().requires_grad_()
        off = qd[..., :M * L * P * 2].view(N, S, M, L, P, 2)
        lg = qd[..., M * L * P * 2:].view(N, S, M, L * P).softmax(-1).view(N, S, M, L, P)
        rd = ref.double()
        if refdim == 2:
            norm = torch.stack((shapes[:, 1], shapes[:, 0]), -1).double()
            lref = rd[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
        else:
            lref = rd[:, :, None, :, None, :2] + off / P * rd[:, :, None, :, None, 2:] * 0.5
        gq, = torch.autograd.grad((lref, lg), qd, (g_loc.double(), g_aw.double()))
        gk, = torch.autograd.grad((loc, aw), q, (g_loc, g_aw))
        e1 = float((loc.double() - lref).abs().max()); e2 = float((aw.double() - lg).abs().max())
        e3 = float((gk.double() - gq).abs().max() / gq.abs().max())
        print(f"{str(dt):15s} refdim {refdim}: forward {fwd:6.1f} us, forward+backward {fb:6.1f} us;  max err loc {e1:.2e} aw {e2:.2e} grad(rel) {e3:.2e}")
