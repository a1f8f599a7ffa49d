"""Measures how far the bf16 model (the headline dtype) is from the float32 reference golden at the small golden config:
per-output errors, cosine of the feature / sentinel-parameter gradients (numbers quoted by tests/test_modules_gpu.py)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import test_modules_cpu as C  # noqa: E402

DEV = "cuda:0"
g = C.load("parseda")
model, bb = C.build_small_parseda()
model = model.to(DEV).to(torch.bfloat16)
gb = {k: (v.to(torch.bfloat16) if v.dtype == torch.float32 else v) for k, v in g.items()}
gb["img_mask"] = g["img_mask"]
mc, out, feats, _ = C.run_small_parseda(model, bb, gb, device=DEV)
loss = 0
for k in C.KEYS:
    ref = g[k]
    got = out[k].float().cpu()
    err = (got - ref).abs().max().item()
    print(f"{k:18s} max|err| {err:.3e}   max|ref| {ref.abs().max().item():.3e}   rel-to-max {err / ref.abs().max().item():.3e}")
    a = out["aux_outputs"][0][k].float().cpu()
    print(f"  aux0 {k:13s} max|err| {(a - g['aux0_' + k]).abs().max().item():.3e}")
    loss = loss + (out[k].float() * g["g_" + k].to(DEV)).sum() + (out["aux_outputs"][0][k].float() * g["g_" + k].to(DEV)).sum() * 0.5
loss.backward()


def cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-300))


for i, (t, _) in enumerate(feats):
    r = g[f"g_feat{i}"]
    print(f"g_feat{i}: cosine {cos(t.grad.float().cpu(), r):.5f}   rel L2 {float((t.grad.float().cpu() - r).norm() / r.norm()):.3e}")
params = dict(model.named_parameters(remove_duplicate=False))
worst = 1.0
for key in g:
    if key.startswith("gparam_") and g[key].numel():
        name = key[len("gparam_"):].replace("__", ".")
        c = cos(params[name].grad.float().cpu(), g[key])
        worst = min(worst, c)
        print(f"grad {name:70s} cosine {c:.5f}  |ref| {float(g[key].norm()):.3e}")
print("worst parameter-gradient cosine:", worst)
