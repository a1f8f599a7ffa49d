"""Names the modules whose forward output or gradients are not repeatable bit for bit between two runs of the SAME eager
train step (same batch, eval mode).  Forward: module outputs compared; the first differing modules (all inputs
identical) are the sources.  Backward: for every module, grad_output identical in both runs but grad_input or a
parameter gradient differing => the module's own backward kernels are the source (not noise arriving from above).
usage: python tools/nondet_modules.py [small|full]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlipv2_amd import parseda, train  # noqa: E402

full = len(sys.argv) > 1 and sys.argv[1] == "full"
if len(sys.argv) > 2 and sys.argv[2] == "det":
    torch.backends.cudnn.deterministic = True          # MIOpen: deterministic solvers only
torch.manual_seed(0)
margs = parseda.default_args(num_queries=300) if full else parseda.default_args(num_queries=40, enc_layers=4, dec_layers=2)
model, criterion = train.build_training(margs, device="cuda:0", with_text_encoder=True)
train.to_bf16(model)
batch = train.synthetic_batch(4, 800, 1333, device="cuda:0") if full else train.synthetic_batch(2, 256, 320, device="cuda:0", triplets=3)
batch[0].tensors = batch[0].tensors.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
step = train.ParSeDATrainStep(model)
model.eval()
train.freeze_parameters_without_gradient(step, criterion, batch)
leaf = {n: m for n, m in step.named_modules() if n and (not list(m.children()) or type(m).__name__ in ("MSDeformAttn", "MultiheadAttention"))}


def tensors(x):
    if torch.is_tensor(x):
        return [x]
    if isinstance(x, (list, tuple)):
        return [t for y in x for t in tensors(y)]
    if isinstance(x, dict):
        return [t for y in x.values() for t in tensors(y)]
    return []


def run():
    rec = {"fin": {}, "fout": {}, "gout": {}, "gin": {}}
    handles = []
    for n, m in leaf.items():
        def fwd(mod, inp, out, n=n):
            rec["fin"].setdefault(n, [t.detach().clone() for t in tensors(inp)])
            rec["fout"].setdefault(n, [t.detach().clone() for t in tensors(out)])
        def bwd(mod, gin, gout, n=n):
            rec["gout"].setdefault(n, [t.detach().clone() for t in tensors(gout) if t is not None])
            rec["gin"].setdefault(n, [t.detach().clone() for t in tensors(gin) if t is not None])
        handles.append(m.register_forward_hook(fwd))
        try:
            handles.append(m.register_full_backward_hook(bwd))
        except Exception:                                    # noqa: BLE001
            pass
    for p in step.parameters():
        p.grad = None
    out = step(*batch)
    loss = criterion.weighted_sum(criterion(out, batch[2]))
    loss.backward()
    torch.cuda.synchronize()
    for h in handles:
        h.remove()
    rec["pgrad"] = {n: p.grad.detach().clone() for n, p in step.named_parameters() if p.grad is not None}
    rec["loss"] = loss.detach().clone()
    return rec


def same(a, b):
    return len(a) == len(b) and all(x.shape == y.shape and torch.equal(x, y) for x, y in zip(a, b))


run()
a, b = run(), run()
print("loss repeatable:", bool(torch.equal(a["loss"], b["loss"])), float(a["loss"]), float(b["loss"]))
src_f = [n for n in leaf if n in a["fout"] and same(a["fin"][n], b["fin"][n]) and not same(a["fout"][n], b["fout"][n])]
print(f"forward: {sum(1 for n in leaf if n in a['fout'] and not same(a['fout'][n], b['fout'][n]))} of {len(a['fout'])} module outputs differ;"
      f" SOURCES (identical inputs, differing output): {len(src_f)}")
for n in src_f[:40]:
    print("   F", n, type(leaf[n]).__name__)
src_b = []
for n in leaf:
    if n not in a["gout"] or n not in b["gout"]:
        continue
    if not same(a["gout"][n], b["gout"][n]) or (n in a["fin"] and not same(a["fin"][n], b["fin"][n])):
        continue                                              # noise arrives from above (or from the forward)
    pg = [k for k in a["pgrad"] if k.startswith(n + ".") and not torch.equal(a["pgrad"][k], b["pgrad"][k])]
    if not same(a["gin"][n], b["gin"][n]) or pg:
        src_b.append((n, type(leaf[n]).__name__, not same(a["gin"][n], b["gin"][n]), [k[len(n) + 1:] for k in pg]))
print(f"backward SOURCES (identical grad_output and forward input, differing grad_input / parameter gradient): {len(src_b)}")
for n, t, gi, pg in src_b[:60]:
    print("   B", n, t, "grad_input differs" if gi else "", "param grads:", pg)
