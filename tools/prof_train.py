"""A few train steps for rocprofv3 --kernel-trace --stats (profiling aid)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlipv2_amd import parseda, train
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
margs = parseda.default_args(num_queries=300)
model, criterion = train.build_training(margs, device="cuda:0", with_text_encoder=True)
batch = train.synthetic_batch(4, 800, 1333, device="cuda:0")
master = not (len(sys.argv) > 2 and sys.argv[2] == "autocast")
if master:
    train.to_bf16(model)
    batch[0].tensors = batch[0].tensors.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
step_module = train.ParSeDATrainStep(model)
opt = train.FusedMasterAdamW(model) if master else train.build_optimizer(model)
model.train()
for _ in range(steps):
    train.train_step(step_module, criterion, opt, batch, autocast_dtype=None if master else torch.bfloat16)
torch.cuda.synchronize()
