"""Library-GEMM timings for the encoder's token-major shapes (T = 4 x 22223 tokens, bf16): forward,
dgrad and wgrad of each Linear, against their HBM-bound time.  Profiling aid."""
import os, sys, torch, time
T = 4 * 22223
dev = "cuda:0"
shapes = [("value/out_proj", 256, 256), ("qproj", 256, 384), ("linear1", 256, 1024), ("linear2", 1024, 256)]


def t_us(fn, n=20):
    """GPU time per call, replayed from a HIP graph so that host-side launch cost does not count"""
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



for name, k, n in shapes:
    x = torch.randn(T, k, device=dev, dtype=torch.bfloat16)
    w = torch.randn(n, k, device=dev, dtype=torch.bfloat16)
    b = torch.randn(n, device=dev, dtype=torch.bfloat16)
    dy = torch.randn(T, n, device=dev, dtype=torch.bfloat16)
    fwd = t_us(lambda: torch.nn.functional.linear(x, w, b))
    dgrad = t_us(lambda: dy @ w)
    wgrad = t_us(lambda: dy.t() @ x)
    bgrad = t_us(lambda: dy.sum(0))
    io_f = (T * k + T * n + n * k) * 2 / 8e12 * 1e6
    io_w = (T * k + T * n) * 2 / 8e12 * 1e6
    fl = 2 * T * k * n
    print(f"{name:16s} K={k:5d} N={n:5d}  fwd {fwd:7.1f} us ({fl / fwd / 1e6:6.0f} TF/s, hbm {io_f:5.1f} us)  "
          f"dgrad {dgrad:7.1f}  wgrad {wgrad:7.1f} (hbm {io_w:5.1f})  bias-grad {bgrad:6.1f}")
