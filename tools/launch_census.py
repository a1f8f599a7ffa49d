"""Kernel launches and GPU time of the FORWARD pass by module (eager step under torch.profiler, record_function ranges
around every module down to depth DEPTH): where the launch-bound chains of tiny kernels sit."""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlipv2_amd import parseda, train
from torch.profiler import profile, ProfilerActivity, record_function
DEPTH = int(sys.argv[1]) if len(sys.argv) > 1 else 3
margs = parseda.default_args(num_queries=300)
model, criterion = train.build_training(margs, device="cuda:0", with_text_encoder=True)
batch = train.synthetic_batch(4, 800, 1333, device="cuda:0")
train.to_bf16(model)
batch[0].tensors = batch[0].tensors.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
step_module = train.ParSeDATrainStep(model)
opt = train.FusedMasterAdamW(model)
model.train()
for _ in range(3):
    train.train_step(step_module, criterion, opt, batch, autocast_dtype=None)
torch.cuda.synchronize()


def wrap(mod, name):
    orig = mod.forward

    def fwd(*a, **k):
        with record_function("MOD:" + name):
            return orig(*a, **k)
    mod.forward = fwd


for name, mod in list(model.named_modules()) + [("criterion", criterion)]:
    if name and name.count(".") < DEPTH:
        wrap(mod, name)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    train.train_step(step_module, criterion, opt, batch, autocast_dtype=None)
    torch.cuda.synchronize()


def census(e):
    n, t = len(e.kernels), sum(k.duration for k in e.kernels)
    for c in e.cpu_children:
        cn, ct = census(c)
        n += cn; t += ct
    return n, t


agg = collections.defaultdict(lambda: [0, 0, 0.0])
total = [0, 0.0]
for e in prof.events():
    if e.name.startswith("MOD:"):
        n, t = census(e)
        a = agg[e.name[4:]]; a[0] += 1; a[1] += n; a[2] += t
    if e.cpu_parent is None:
        n, t = census(e); total[0] += n; total[1] += t
bw = collections.defaultdict(lambda: [0, 0, 0.0])
other = collections.defaultdict(lambda: [0, 0, 0.0])
for e in prof.events():
    if e.name.startswith("autograd::engine::evaluate_function: "):
        n, t = census(e)
        a = bw[e.name.split(": ", 1)[1]]; a[0] += 1; a[1] += n; a[2] += t
    elif e.cpu_parent is None and not e.name.startswith("MOD:"):
        n, t = census(e)
        if n:
            a = other[e.name]; a[0] += 1; a[1] += n; a[2] += t
print(f"backward nodes: {sum(a[1] for a in bw.values())} launches, {sum(a[2] for a in bw.values()) / 1e3:.2f} ms")
for name, (c, n, t) in sorted(bw.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"  {name:66s} {c:5d} {n:8d} {t / 1e3:7.3f}")
print("top-level ops outside modules / backward nodes:")
for name, (c, n, t) in sorted(other.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"  {name:66s} {c:5d} {n:8d} {t / 1e3:7.3f}")
print(f"whole step: {total[0]} launches, {total[1] / 1e3:.2f} ms of kernels (forward + backward + optimizer)")
print(f"{'module (forward only)':70s} calls launches   ms")
for name, (c, n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
    print(f"{name:70s} {c:5d} {n:8d} {t / 1e3:7.3f}")
