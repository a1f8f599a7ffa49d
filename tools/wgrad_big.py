"""Ablation of the token-major weight-gradient kernel on the encoder's big shapes (ablation build):
RLIPV2_WGRAD_DBG 0 = full, 1 = no DMA in the steady state (compute only), 3 = 1 + MFMAs on constant fragments (no LDS reads),
5 = 1 + LDS reads only (no MFMAs), 8 = DMA only; RLIPV2_WGRAD_BLOCKS = workgroups.  (RLIPV2_WGRAD_WIDE belonged to the
256x256-tile experiment recorded in profiles/r02_wgrad_ablation.txt; the shipped library has no such variant.)"""
import os, subprocess, sys
if len(sys.argv) == 1:
    env = dict(os.environ)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env["RLIPV2_LIB_PATH"] = os.path.join(root, "tools", "_build", "librlipv2_msda_ablation.so")
    combos = [(0, 512, d) for d in (0, 1, 3, 5, 8)] if "split" in os.environ.get("WGRAD_BIG", "") \
        else [(0, b, d) for b in (256, 512, 768) for d in (0, 1, 8)]
    for wide, blocks, dbg in combos:
        if True:
            env["RLIPV2_WGRAD_DBG"], env["RLIPV2_WGRAD_BLOCKS"], env["RLIPV2_WGRAD_WIDE"] = str(dbg), str(blocks), str(wide)
            subprocess.run([sys.executable, __file__, "child"], env=env)
    sys.exit(0)
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlipv2_amd import linear
T = 4 * 22223


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


out = []
shapes = [tuple(int(v) for v in t.split("x")) for t in os.environ["WGRAD_SHAPES"].split(",")] if os.environ.get("WGRAD_SHAPES") \
    else [(2048, 256), (256, 2048), (256, 256), (1024, 256), (512, 512)]
for M, K in shapes:
    dy = torch.randn(T, M, device="cuda", dtype=torch.bfloat16); x = torch.randn(T, K, device="cuda", dtype=torch.bfloat16)
    out.append(f"{M}x{K}: {t_us(lambda: linear.linear_wgrad(dy, x, True, torch.bfloat16)):6.1f} us")
print(f"wide {os.environ['RLIPV2_WGRAD_WIDE']} blocks {os.environ['RLIPV2_WGRAD_BLOCKS']:>4s} dbg {os.environ['RLIPV2_WGRAD_DBG']}   " + "   ".join(out), flush=True)
