"""Operator-level (aten op + input shapes) GPU time of the steady-state train step (torch.profiler)."""
import os, sys, torch
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    train.train_step(step_module, criterion, opt, batch, autocast_dtype=None)
    torch.cuda.synchronize()
if os.environ.get("BY_SHAPE", "1") == "1":
    print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_device_time_total", row_limit=90,
                                                             max_name_column_width=48, max_shapes_column_width=90))
else:
    rows = [(e.key, e.count, e.self_device_time_total) for e in prof.key_averages() if e.self_device_time_total > 0]
    rows.sort(key=lambda r: -r[1])
    print(f"{'op':60s} {'calls':>7s} {'self GPU ms':>12s}")
    for k, c, t in rows[:70]:
        print(f"{k[:60]:60s} {c:7d} {t / 1e3:12.2f}")
