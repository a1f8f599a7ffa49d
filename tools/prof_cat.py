"""Call sites of one aten op in the eager train step (default: aten::cat): op, input shapes, Python frames."""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlipv2_amd import parseda, train
from torch.profiler import profile, ProfilerActivity
op = sys.argv[1] if len(sys.argv) > 1 else "aten::cat"
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
rows = collections.defaultdict(lambda: [0, 0.0])
for e in prof.key_averages(group_by_input_shape=True, group_by_stack_n=40):
    if e.key != op:
        continue
    frames = [f for f in e.stack if ("rlipv2_amd" in f or "transformers" in f or "torch/nn/modules" in f) and "module.py" not in f]
    site = " <- ".join(f.split("/")[-1][:46] for f in frames[:3]) or (e.stack[0][:60] if e.stack else "backward / no python frame")
    r = rows[(str([s for s in e.input_shapes if s])[:60], site)]
    r[0] += e.count; r[1] += e.self_device_time_total
tot = sum(r[1] for r in rows.values())
print(f"{op}: {tot / 1e3:.3f} ms GPU per step, {sum(r[0] for r in rows.values())} calls")
for (shapes, site), (c, t) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{t / 1e3:7.3f} ms {c:4d}x  {shapes:60s} {site}")
