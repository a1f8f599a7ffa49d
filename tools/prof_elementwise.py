"""Where the non-GEMM aten ops of the train step come from: op, input shapes, first rlipv2_amd frame, calls, GPU time."""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlipv2_amd import parseda, train
from torch.profiler import profile, ProfilerActivity
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    train.train_step(step_module, criterion, opt, batch, autocast_dtype=None)
    torch.cuda.synchronize()
skip = ("mm", "conv", "Function", "msda", "Cijk", "igemm", "void ", "Memcpy", "Memset", "wgrad", "reduce_partials")
rows = collections.defaultdict(lambda: [0, 0.0])
for e in prof.key_averages(group_by_input_shape=True, group_by_stack_n=12):
    t = e.self_device_time_total
    if t <= 0 or any(s in e.key for s in skip):
        continue
    site = next((f for f in e.stack if "rlipv2_amd" in f or "bench.py" in f), e.stack[0] if e.stack else "?")
    site = site.split("rlipv2_amd/")[-1][:60]
    shapes = str([s for s in e.input_shapes if s])[:70]
    r = rows[(e.key, shapes, site)]; r[0] += e.count; r[1] += t
tot = sum(r[1] for r in rows.values())
print(f"non-GEMM aten ops: {tot / 1e3:.2f} ms GPU per step")
for (k, shapes, site), (c, t) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:90]:
    print(f"{t / 1e3:7.3f} ms {c:4d}x {t / c:7.1f} us  {k[:32]:32s} {shapes:70s} {site}")
