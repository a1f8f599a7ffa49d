"""Library-GEMM layout variants for the token-major Linears (which operand layout hipBLASLt handles best)."""
import os, sys, torch
T = 4 * 22223
dev = "cuda:0"


def t_us(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for k, n in [(256, 2048), (2048, 256), (256, 256), (256, 384)]:
    x = torch.randn(T, k, device=dev, dtype=torch.bfloat16)
    w = torch.randn(n, k, device=dev, dtype=torch.bfloat16)          # nn.Linear layout [out, in]
    wt = w.t().contiguous()                                          # [in, out]
    b = torch.randn(n, device=dev, dtype=torch.bfloat16)
    dy = torch.randn(T, n, device=dev, dtype=torch.bfloat16)
    out = torch.empty(T, n, device=dev, dtype=torch.bfloat16)
    res = {
        "fwd linear(x,w,b)": t_us(lambda: torch.nn.functional.linear(x, w, b)),
        "fwd addmm(b,x,w.t())": t_us(lambda: torch.addmm(b, x, w.t())),
        "fwd addmm(b,x,wt)": t_us(lambda: torch.addmm(b, x, wt)),
        "fwd mm(x,wt) no bias": t_us(lambda: torch.mm(x, wt)),
        "fwd _addmm_activation": t_us(lambda: torch._addmm_activation(b, x, w.t())),
        "fwd _addmm_activation wt": t_us(lambda: torch._addmm_activation(b, x, wt)),
        "dgrad dy@w": t_us(lambda: dy @ w),
        "dgrad dy@wt.t()": t_us(lambda: dy @ wt.t()),
    }
    io = (T * k + T * n) * 2 / 8e12 * 1e6
    print(f"K={k} N={n}  (HBM-bound {io:.0f} us fwd)")
    for name, v in res.items():
        print(f"    {name:28s} {v:7.1f} us")
