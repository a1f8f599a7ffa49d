"""Ablation arms of the MFMA weight-gradient kernel (csrc/token_gemm.hip; ablation build, RLIPV2_WGRAD_DBG bits): main loop, DMA
stream, partial-sum round trip -- profiles/r02_wgrad_ablation.txt."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlipv2_amd import linear
T = 4 * 22223
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
for M, K in [(256, 256), (1024, 256)]:
    dy = torch.randn(T, M, device="cuda", dtype=torch.bfloat16); x = torch.randn(T, K, device="cuda", dtype=torch.bfloat16)
    print(M, K, os.environ.get("RLIPV2_WGRAD_DBG"), os.environ.get("RLIPV2_WGRAD_BLOCKS"), f"{t_us(lambda: linear.linear_wgrad(dy, x, True, torch.bfloat16)):.1f} us")
