"""Which parameter gradients of the train step are NOT repeatable bit for bit from run to run (same batch, eval mode:
no dropout), and how far apart two runs are -- to name the kernels behind the run-to-run gradient noise (round 2
measured 3.6 % relative L2 between two captures of the same schedule and blamed "atomics" without naming anything).
usage: python tools/grad_repeat.py [small|full]"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlipv2_amd import parseda, train  # noqa: E402

full = len(sys.argv) > 1 and sys.argv[1] == "full"
torch.manual_seed(0)
margs = parseda.default_args(num_queries=300) if full else parseda.default_args(num_queries=40, enc_layers=4, dec_layers=2)
model, criterion = train.build_training(margs, device="cuda:0", with_text_encoder=True)
train.to_bf16(model)
batch = train.synthetic_batch(4, 800, 1333, device="cuda:0") if full else train.synthetic_batch(2, 256, 320, device="cuda:0", triplets=3)
batch[0].tensors = batch[0].tensors.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
step = train.ParSeDATrainStep(model)
model.eval()
train.freeze_parameters_without_gradient(step, criterion, batch)
params = [(n, p) for n, p in step.named_parameters() if p.requires_grad]


def run():
    for _, p in params:
        p.grad = None
    out = step(*batch)
    loss = criterion.weighted_sum(criterion(out, batch[2]))
    loss.backward()
    torch.cuda.synchronize()
    losses.append(float(loss))
    outs.append({k: v.detach().float().clone() for k, v in out.items() if torch.is_tensor(v)})
    return {n: p.grad.detach().clone() for n, p in params}


losses, outs = [], []
run()
a, b = run(), run()
print("losses of the three runs:", losses)
for k in outs[1]:
    print(f"   output {k:24s} repeatable: {bool(torch.equal(outs[1][k], outs[2][k]))}")
groups = collections.OrderedDict()
tot_num = tot_den = 0.0
for n, _ in params:
    ga, gb = a[n].float(), b[n].float()
    same = torch.equal(a[n], b[n])
    num, den = float((ga - gb).norm()) ** 2, float(ga.norm()) ** 2
    tot_num += num
    tot_den += den
    key = ".".join(n.split(".")[:4])
    g = groups.setdefault(key, [0, 0, 0.0, 0.0])
    g[0] += 1
    g[1] += 0 if same else 1
    g[2] += num
    g[3] += den
print(f"whole gradient: relative L2 distance between two eager runs = {(tot_num / max(tot_den, 1e-30)) ** 0.5:.3e}")
print("modules with non-repeatable parameter gradients (count differing / total, relative L2 of the module):")
for k, (n, d, num, den) in groups.items():
    if d:
        print(f"  {k:70s} {d:3d}/{n:3d}   {(num / max(den, 1e-30)) ** 0.5:.3e}")
print("bit-identical modules:", sum(1 for v in groups.values() if v[1] == 0), "of", len(groups))
