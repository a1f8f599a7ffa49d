"""A few train steps for rocprofv3 --kernel-trace --stats (profiling aid)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlipv2_amd import parseda, train
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
margs = parseda.default_args(num_queries=300)
model, criterion = train.build_training(margs, device="cuda:0", with_text_encoder=True)
step_module = train.ParSeDATrainStep(model)
opt = train.build_optimizer(model)
batch = train.synthetic_batch(4, 800, 1333, device="cuda:0")
model.train()
for _ in range(steps):
    train.train_step(step_module, criterion, opt, batch)
torch.cuda.synchronize()
