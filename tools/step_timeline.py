"""GPU-side timeline of the graphed train step: HIP events at the phase boundaries (no extra syncs) and the
host clock at the same points, averaged over a few steps.  Shows where the GPU waits for the host."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlipv2_amd import parseda, train
from rlipv2_amd.train import FusedMasterAdamW as MasterWeightAdamW
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
margs = parseda.default_args(num_queries=300)
model, criterion = train.build_training(margs, device="cuda:0", with_text_encoder=True)
batch = train.synthetic_batch(4, 800, 1333, device="cuda:0")
train.to_bf16(model)
batch[0].tensors = batch[0].tensors.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
step_module = train.ParSeDATrainStep(model)
opt = MasterWeightAdamW(model)
model.train()
sync = None
if os.environ.get("STEP_DP") in ("1", "2", "3", "4", "5"):                     # 1-rank RCCL group: the data-parallel flavour of the step
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29579")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    train.freeze_parameters_without_gradient(step_module, criterion, batch)
    sync = train.GradientSynchronizer([p for p in step_module.parameters() if p.requires_grad])
    if os.environ["STEP_DP"] == "5":
        sync.comm = None                                        # ablation: bucket copies + collectives on the compute streams
    if os.environ["STEP_DP"] == "4":
        sync.launch_bucket = lambda k, streams=(): None        # ablation: hooks + copies into the flat buffer, no collective
    if os.environ["STEP_DP"] in ("2", "3"):
        sync.all_reduce = lambda: None                     # ablation: no collective
    if os.environ["STEP_DP"] == "3":
        sync.pack = lambda grads: grads                    # ablation: no packing either
graphed = train.graph_step_module(step_module, model, batch, synchronizer=sync, criterion=criterion,
                                  overlap=os.environ.get("RLIPV2_DP_OVERLAP", "1") == "1")
names = ["forward graph (model + cost matrices)", "host: D2H + assignment + H2D", "backward graph (losses + backward)", "optimizer"]
acc_gpu = [0.0] * 4; acc_host = [0.0] * 4
for it in range(steps + 3):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
    hs = []
    samples, text, targets = batch
    opt.zero_grad(set_to_none=True)
    ev[0].record(); hs.append(time.perf_counter())
    graphed._load_inputs(samples, text, targets)
    graphed.fwd_graph.replay()
    ev[1].record(); hs.append(time.perf_counter())
    graphed.pinned_index.copy_(criterion.assign(graphed.state, graphed._cost_on_host()))
    graphed.static_index.copy_(graphed.pinned_index, non_blocking=True)
    if sync is not None:
        graphed.static_num.copy_(criterion._num_interactions(graphed.sizes, graphed.static_num.device).reshape(1))
    ev[2].record(); hs.append(time.perf_counter())
    graphed.bwd_graph.replay()
    graphed._deliver()
    ev[3].record(); hs.append(time.perf_counter())
    opt.step(0.1)
    ev[4].record(); hs.append(time.perf_counter())
    torch.cuda.synchronize()
    if it >= 3:
        for k in range(4):
            acc_gpu[k] += ev[k].elapsed_time(ev[k + 1]) / steps
            acc_host[k] += (hs[k + 1] - hs[k]) * 1e3 / steps
print(f"{'phase':44s} {'GPU timeline ms':>16s} {'host issue ms':>14s}")
for k in range(4):
    print(f"{names[k]:44s} {acc_gpu[k]:16.2f} {acc_host[k]:14.2f}")
print(f"{'sum':44s} {sum(acc_gpu):16.2f} {sum(acc_host):14.2f}")
if sync is not None:
    import torch.distributed as dist
    print(f"\ndata-parallel schedule: {'bucketed all-reduce captured inside the backward graph (overlapped)' if graphed.overlap else 'one flat all-reduce after the backward replay'}"
          f"; world size {dist.get_world_size()}")
    names = [n for n, p in step_module.named_parameters() if p.requires_grad]
    for k, b in enumerate(sync.buckets):
        a, e = sync.ranges[k]
        print(f"  bucket {k}: {len(b):4d} tensors {(e - a) * sync.flat.element_size() / 1e6:7.1f} MB  first: {names[b[0]][:60]:60s} last: {names[b[-1]][:60]}")
