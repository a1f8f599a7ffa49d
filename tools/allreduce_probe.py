"""1-rank RCCL all-reduce / copy timings for the flat gradient buffer (what the 1-rank DP overhead is made of)."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
flat = torch.zeros(212_700_000, dtype=torch.bfloat16, device="cuda")
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
print(f"all_reduce AVG 425 MB, 1 rank: {t(lambda: dist.all_reduce(flat, op=dist.ReduceOp.AVG)):.3f} ms")
print(f"all_reduce SUM 425 MB, 1 rank: {t(lambda: dist.all_reduce(flat)):.3f} ms")
src = torch.zeros_like(flat)
print(f"device copy 425 MB: {t(lambda: flat.copy_(src)):.3f} ms")
dist.destroy_process_group()
