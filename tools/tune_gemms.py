"""Records hipBLASLt/rocBLAS solution choices (PyTorch TunableOp) for the library GEMMs of one eager train step.
Output: gpurun_out/tunableop_step.csv (filtered into rlipv2_amd/tuned/ by hand: only the token-major shapes that win)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlipv2_amd import parseda, train
import torch.cuda.tunable as tn
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "tunableop_step.csv")
margs = parseda.default_args(num_queries=300)
model, criterion = train.build_training(margs, device="cuda:0", with_text_encoder=True)
batch = train.synthetic_batch(4, 800, 1333, device="cuda:0")
train.to_bf16(model)
batch[0].tensors = batch[0].tensors.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
step_module = train.ParSeDATrainStep(model)
opt = train.FusedMasterAdamW(model)
model.train()
train.train_step(step_module, criterion, opt, batch, autocast_dtype=None)
torch.cuda.synchronize()
tn.enable(True); tn.tuning_enable(True)
tn.set_max_tuning_duration(int(os.environ.get("TUNE_MS", 30))); tn.set_max_tuning_iterations(20)
tn.set_filename(out)
t0 = time.time()
train.train_step(step_module, criterion, opt, batch, autocast_dtype=None)
torch.cuda.synchronize()
print(f"tuning step took {time.time() - t0:.1f} s; {len(tn.get_results())} entries")
