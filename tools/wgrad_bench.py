"""Timing of the token-major weight-gradient kernel against the library GEMM + bias reduction it
replaces (profiling aid; prints HBM-bound time at 8 TB/s beside each)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlipv2_amd import linear
T = int(os.environ.get("T", 4 * 22223))


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



if __name__ == "__main__":
  for M, K in [(256, 256), (384, 256), (1024, 256), (256, 1024), (2048, 256), (256, 2048)]:
      dy = torch.randn(T, M, device="cuda", dtype=torch.bfloat16)
      x = torch.randn(T, K, device="cuda", dtype=torch.bfloat16)
      mine = t_us(lambda: linear.linear_wgrad(dy, x, True, torch.bfloat16))
      lib = t_us(lambda: (dy.t() @ x, dy.sum(0)))
      hbm = T * (M + K) * 2 / 8e12 * 1e6
      fl = 2.0 * T * M * K
      print(f"dW[{M:4d},{K:4d}]  mfma kernel {mine:7.1f} us ({fl / mine / 1e6:6.0f} TF/s, {T * (M + K) * 2 / mine / 1e3:6.0f} GB/s)"
            f"   library {lib:7.1f} us   hbm-bound {hbm:5.1f} us")
