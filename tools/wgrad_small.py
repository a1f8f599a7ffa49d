"""Small-token-count weight gradients: own split kernel vs library GEMM + ones-row bias GEMM (graph-replayed GPU time)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlipv2_amd import linear
from wgrad_bench import t_us  # noqa: E402  (prints the big-shape table first; harmless)
shapes = [(320, 768, 768), (320, 768, 3072), (320, 3072, 768), (256, 768, 768), (600, 256, 256), (1200, 256, 256),
          (1092, 2048, 256), (1200, 256, 2048), (1200, 2048, 256), (600, 256, 2048), (4200, 2048, 512), (4200, 512, 2048),
          (16800, 1024, 256), (16800, 256, 1024), (16800, 512, 1024), (66800, 512, 128), (66800, 128, 512),
          (2048, 256, 256), (4096, 256, 256), (8192, 256, 256), (8192, 768, 768)]
for T, M, K in shapes:
    dy = torch.randn(T, M, device="cuda", dtype=torch.bfloat16)
    x = torch.randn(T, K, device="cuda", dtype=torch.bfloat16)
    ones = torch.ones(1, T, device="cuda", dtype=torch.bfloat16)
    mine = t_us(lambda: linear.linear_wgrad(dy, x, True, torch.bfloat16))
    lib = t_us(lambda: (dy.t() @ x, ones @ dy))
    libw = t_us(lambda: dy.t() @ x)
    print(f"T={T:6d} dW[{M:4d},{K:4d}]  own {mine:6.1f} us   library+ones-row {lib:6.1f} us (gemm alone {libw:6.1f})")
