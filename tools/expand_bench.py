"""expand_gemm (csrc/expand_gemm.hip) against the library GEMM + elementwise pair it replaces (graph-replayed GPU time)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from rlipv2_amd import linear
from wgrad_bench import t_us
T = int(os.environ.get("T", 4 * 22223))
if os.environ.get("TUNED", "1") == "1":
    linear.use_tuned_library_gemms()
for N in ([int(os.environ["ONLY"])] if os.environ.get("ONLY") else [2048, 1024, 256, 384]):
    x = torch.randn(T, 256, device="cuda", dtype=torch.bfloat16)
    w1 = (torch.randn(N, 256, device="cuda") / 16).to(torch.bfloat16)
    b1 = torch.randn(N, device="cuda", dtype=torch.bfloat16)
    h = torch._addmm_activation(b1, x, w1.t())
    w2 = (torch.randn(256, N, device="cuda") / 16).to(torch.bfloat16)
    dy = torch.randn(T, 256, device="cuda", dtype=torch.bfloat16)
    w2t = w2.t().contiguous()
    fwd_own = t_us(lambda: linear.expand_gemm(x, w1, bias=b1, relu=True))
    fwd_lib = t_us(lambda: torch._addmm_activation(b1, x, w1.t()))
    fwd_own_nb = t_us(lambda: linear.expand_gemm(x, w1, bias=b1))
    fwd_lib_nb = t_us(lambda: torch.addmm(b1, x, w1.t()))
    bwd_own = t_us(lambda: linear.expand_gemm(dy, w2t, mask=h))
    bwd_own_t = t_us(lambda: linear.expand_gemm(dy, w2.t(), mask=h))
    bwd_lib = t_us(lambda: torch.ops.aten.threshold_backward(dy @ w2, h, 0))
    bwd_lib_gemm = t_us(lambda: dy @ w2)
    out_mb = T * N * 2 / 1e6
    print(f"N={N:5d}  fwd bias+relu own {fwd_own:6.1f} us  lib {fwd_lib:6.1f} | fwd bias own {fwd_own_nb:6.1f} lib {fwd_lib_nb:6.1f} | "
          f"dgrad+mask own {bwd_own:6.1f} (incl. W^T copy {bwd_own_t:6.1f})  lib {bwd_lib:6.1f} (gemm alone {bwd_lib_gemm:6.1f})"
          f"   [output {out_mb:.0f} MB = {out_mb / 8e3 * 1e3:.0f} us at 8 TB/s]")
