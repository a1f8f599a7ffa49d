import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlipv2_amd import parseda, train
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29578")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
model, crit = train.build_training(parseda.default_args(num_queries=300), device="cuda:0")
train.to_bf16(model)
params = [p for p in model.parameters() if p.requires_grad]
grads = [torch.randn_like(p) for p in params]
sync = train.GradientSynchronizer(params)
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
g = torch.cuda.CUDAGraph()
sync.pack(grads); torch.cuda.synchronize()
with torch.cuda.graph(g, capture_error_mode="thread_local"):
    sync.pack(grads)
print(f"pack eager {t(lambda: sync.pack(grads)):.3f} ms   pack graph {t(lambda: g.replay()):.3f} ms   all_reduce {t(lambda: sync.all_reduce()):.3f} ms   n_params {len(params)}")
opt = train.FusedMasterAdamW(model)
for p, v in zip(params, sync.views): p.grad = v
print(f"optimizer on flat views {t(lambda: opt.step(0.1)):.3f} ms")
for p, gr in zip(params, grads): p.grad = gr
print(f"optimizer on separate grads {t(lambda: opt.step(0.1)):.3f} ms")
dist.destroy_process_group()
