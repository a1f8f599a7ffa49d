"""Which weight-gradient shapes the train step runs and what each costs (eager step, HIP events around every call)."""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlipv2_amd import parseda, train, linear
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
orig = linear.linear_wgrad
log = []


def timed(dy, x, with_bias=True, out_dtype=torch.bfloat16):
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record(); r = orig(dy, x, with_bias, out_dtype); e.record()
    log.append((dy.numel() // dy.shape[-1], dy.shape[-1], x.shape[-1], s, e))
    return r


linear.linear_wgrad = timed
train.train_step(step_module, criterion, opt, batch, autocast_dtype=None)
torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for T, M, K, s, e in log:
    a = agg[(T, M, K)]; a[0] += 1; a[1] += s.elapsed_time(e) * 1e3
tot = sum(a[1] for a in agg.values())
print(f"{len(log)} calls, {tot / 1e3:.2f} ms")
for (T, M, K), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    hbm = T * (M + K) * 2 / 8e12 * 1e6
    print(f"T={T:7d} M={M:5d} K={K:5d}  {n:3d}x  {us / n:7.1f} us each  {us / 1e3:6.2f} ms   hbm {hbm:6.1f} us")
