"""1-rank RCCL check of the data-parallel schedules of the graphed step: gradients of the overlapped (in-graph, bucketed)
schedule against the flat one (distance = relative L2 over all gradients; two captures of the same schedule differ by
~0.04 from the atomics of the attention backward)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.distributed as dist
from rlipv2_amd import parseda, train
DEV = "cuda:0"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29656")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
torch.manual_seed(0)
margs = parseda.default_args(num_queries=40, enc_layers=4, dec_layers=2)
model, criterion = train.build_training(margs, device=DEV, with_text_encoder=True)
train.to_bf16(model)
batch = train.synthetic_batch(2, 256, 320, device=DEV, triplets=3)
batch[0].tensors = batch[0].tensors.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
step = train.ParSeDATrainStep(model)
model.eval()
train.freeze_parameters_without_gradient(step, criterion, batch)
params = [p for p in step.parameters() if p.requires_grad]
names = [n for n, p in step.named_parameters() if p.requires_grad]
def run(mode):
    sync = train.GradientSynchronizer(params, bucket_bytes=32 << 20)
    if mode == "sum":
        sync._avg = False
    if mode == "nocoll":
        sync.launch_bucket = lambda k, streams=(): None
    graphed = train.GraphedStep(step, model, batch, synchronizer=sync, criterion=criterion, overlap=mode != "flat")
    for _ in range(3):
        graphed.run(*batch)
    torch.cuda.synchronize()
    return [p.grad.detach().float().clone() for p in params]
def dist_(x, y):
    num = sum(float((a - b).pow(2).sum()) for a, b in zip(x, y))
    return (num / sum(float(b.pow(2).sum()) for b in y)) ** 0.5
ref = run("flat")
for mode in ("overlap", "overlap"):
    g = run(mode)
    print(mode, "distance to flat", dist_(g, ref), "absmax", max(float(t.abs().max()) for t in g), flush=True)
dist.destroy_process_group()
