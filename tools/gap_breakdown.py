"""Host-side breakdown of the GPU-idle gap between the forward and the backward graph of the train step (the assignment)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlipv2_amd import parseda, train
margs = parseda.default_args(num_queries=300)
model, criterion = train.build_training(margs, device="cuda:0", with_text_encoder=True)
batch = train.synthetic_batch(4, 800, 1333, device="cuda:0")
train.to_bf16(model)
batch[0].tensors = batch[0].tensors.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
step_module = train.ParSeDATrainStep(model)
opt = train.FusedMasterAdamW(model)
model.train()
g = train.graph_step_module(step_module, model, batch, criterion=criterion)
samples, text, targets = batch
names = ["wait for the forward graph (sync)", "cost matrices -> pinned host buffer", "assign (native batched solver)", "pinned copy + H2D", "(num_interactions: static when not distributed)", "backward replay call"]
acc = [0.0] * 6; gap = 0.0
steps = 20
for it in range(steps + 3):
    opt.zero_grad(set_to_none=True)
    g._load_inputs(samples, text, targets)
    g.fwd_graph.replay()
    e0 = torch.cuda.Event(enable_timing=True); e0.record()
    t = [time.perf_counter()]
    torch.cuda.synchronize(); t.append(time.perf_counter())
    C = g._cost_on_host(); t.append(time.perf_counter())
    idx = criterion.assign(g.state, C); t.append(time.perf_counter())
    g.pinned_index.copy_(idx); g.static_index.copy_(g.pinned_index, non_blocking=True); t.append(time.perf_counter())
    t.append(time.perf_counter())
    e1 = torch.cuda.Event(enable_timing=True); e1.record()
    g.bwd_graph.replay(); t.append(time.perf_counter())
    g._deliver(); opt.step(0.1)
    torch.cuda.synchronize()
    if it >= 3:
        for k in range(6):
            acc[k] += (t[k + 1] - t[k]) * 1e6 / steps
        gap += e0.elapsed_time(e1) * 1e3 / steps
for n, a in zip(names, acc):
    print(f"{n:40s} {a:8.1f} us host")
print(f"GPU events forward-end -> before backward replay: {gap:.1f} us (includes the explicit sync this tool adds)")
