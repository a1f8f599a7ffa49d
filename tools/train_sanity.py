"""Does the optimised train step actually train?  N steps on one fixed synthetic batch, graphed vs eager:
the weighted loss must fall along both trajectories and the two must stay close (they differ only by
atomics-order / bf16 noise and by dropout masks).  Profiling / sanity aid."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlipv2_amd import parseda, train
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
out = {}
for mode in ("graphed", "eager"):
    torch.manual_seed(0)
    margs = parseda.default_args(num_queries=300)
    model, criterion = train.build_training(margs, device="cuda:0", with_text_encoder=True)
    batch = train.synthetic_batch(4, 800, 1333, device="cuda:0")
    train.to_bf16(model)
    batch[0].tensors = batch[0].tensors.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    step = train.ParSeDATrainStep(model)
    model.train()
    train.freeze_parameters_without_gradient(step, criterion, batch)
    opt = train.FusedMasterAdamW(model, lr=1e-4, lr_backbone=1e-5, text_encoder_lr=1e-5)
    if mode == "graphed":
        step = train.graph_step_module(step, model, batch, criterion=criterion)
    losses = []
    for i in range(steps):
        losses.append(float(train.train_step(step, criterion, opt, batch, autocast_dtype=None)))
    out[mode] = losses
    print(mode, " ".join(f"{v:.3f}" for v in losses[::max(1, steps // 12)]), f"... last {losses[-1]:.3f}", flush=True)
    finite = all(v == v and abs(v) < 1e6 for v in losses)
    print(f"  finite: {finite}; first 5 mean {sum(losses[:5]) / 5:.3f} -> last 5 mean {sum(losses[-5:]) / 5:.3f}")
    del model, criterion, opt, step
    torch.cuda.empty_cache()
