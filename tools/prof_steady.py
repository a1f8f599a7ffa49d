"""Steady-state kernel breakdown of the train step with torch.profiler (after MIOpen's find step and the
allocator have settled): per-category GPU time and launches per step, top kernels, GPU-busy vs wall."""
import os, sys, time, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlipv2_amd import parseda, train
from torch.profiler import profile, ProfilerActivity
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
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
t0 = time.perf_counter()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(steps):
        train.train_step(step_module, criterion, opt, batch, autocast_dtype=None)
    torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / steps * 1e3


def classify(n):
    l = n.lower()
    if 'msda' in l: return 'msda'
    if 'cijk' in l or 'gemm' in l and 'igemm' not in l and 'conv' not in l: return 'gemm'
    if 'conv' in l or 'igemm' in l or 'miopen' in l or 'im2col' in l: return 'conv'
    if 'memcpy' in l or 'copybuffer' in l or 'copy_kernel' in l or 'catarray' in l: return 'copy'
    if 'fillbuffer' in l or 'fillfunctor' in l or 'memset' in l: return 'fill'
    if 'layer_norm' in l or 'cucompute' in l or 'batch_norm' in l or 'layernorm' in l: return 'norm'
    if 'softmax' in l: return 'softmax'
    if 'multi_tensor' in l or 'adam' in l: return 'optimizer'
    if 'reduce' in l: return 'reduce'
    if 'elementwise' in l or 'subtensorop' in l or 'optensor' in l: return 'elementwise'
    if 'index' in l or 'gather' in l or 'scatter' in l or 'sort' in l: return 'index'
    return 'other'


cat, calls, per = collections.Counter(), collections.Counter(), collections.Counter()
pc = collections.Counter()
for e in prof.events():
    if e.device_type.name != 'CUDA' and str(e.device_type) != 'DeviceType.CUDA':
        continue
    t = e.device_time if hasattr(e, 'device_time') else e.cuda_time
    c = classify(e.name)
    cat[c] += t / steps / 1e3; calls[c] += 1 / steps
    per[e.name] += t / steps / 1e3; pc[e.name] += 1 / steps
print(f"wall {wall:.2f} ms/step; GPU busy (sum of kernels) {sum(cat.values()):.2f} ms/step; "
      f"launches {sum(calls.values()):.0f}/step")
for c, t in cat.most_common():
    print(f"  {c:12s} {t:8.2f} ms/step {calls[c]:7.0f} launches")
small = collections.Counter(); smalln = collections.Counter(); big = collections.Counter(); bign = collections.Counter()
for e in prof.events():
    if e.device_type.name != 'CUDA' and str(e.device_type) != 'DeviceType.CUDA':
        continue
    t = e.device_time if hasattr(e, 'device_time') else e.cuda_time
    c = classify(e.name)
    if t < 8:
        small[c] += t / steps / 1e3; smalln[c] += 1 / steps
    else:
        big[c] += t / steps / 1e3; bign[c] += 1 / steps
print("kernels shorter than 8 us (latency-bound) vs longer, per category:")
for c in cat:
    print(f"  {c:12s} short {small[c]:6.2f} ms ({smalln[c]:5.0f}x)   long {big[c]:6.2f} ms ({bign[c]:5.0f}x)")
print("top kernels:")
for n, t in per.most_common(int(os.environ.get("TOP", "45"))):
    print(f"  {t:7.2f} ms {pc[n]:6.0f}x  {n[:150]}")
