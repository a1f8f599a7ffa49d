"""ALIF attention core at the train step's shape: fused HIP route against the PyTorch route (forward, forward + backward)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlipv2_amd import alif, parseda
DEV = "cuda:0"
torch.manual_seed(0)
m = alif.RLIPv2_VLFuse(parseda.default_args()).to(DEV).to(torch.bfloat16).train()
B, Tv, Tl = 4, 273, 64
v = torch.randn(B, Tv, 256, device=DEV).bfloat16().requires_grad_(True)
l = torch.randn(B, Tl, 768, device=DEV).bfloat16().requires_grad_(True)
pos = torch.randn(B, Tv, 256, device=DEV).bfloat16()
vm = torch.zeros(B, Tv, dtype=torch.bool, device=DEV); lm = torch.ones(B, Tl, dtype=torch.bool, device=DEV)
def run(bwd):
    out = m({"visual": {"src": v, "padding_mask": vm, "pos": pos}, "lang": {"hidden": l, "masks": lm}})
    if bwd:
        (out["visual"]["src"].float().sum() + out["lang"]["hidden"].float().sum()).backward()
for fused in (True, False):
    alif.fused_attention = fused
    for bwd in (False, True):
        g = torch.cuda.CUDAGraph()
        for _ in range(3): run(bwd)
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            run(bwd)
        for _ in range(5): g.replay()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50): g.replay()
        b.record(); torch.cuda.synchronize()
        print(f"fused={fused} {'fwd+bwd' if bwd else 'fwd    '} {a.elapsed_time(b) / 50 * 1e3:8.1f} us per VLFuse layer (graph replay)")
