"""A/B of round 4's host-side restructures on the real train step (they were written while the GPU pool was closed: every one
is checked on the CPU against the goldens, none was ever timed).  For each switch: the graphed bf16 train step (bench.py's
configuration: batch 4, 800x1333, 300 queries) with the restructure ON (product) and OFF, ms per step over `steps` replays and
kernel launches per eager step.  Also checks that ON and OFF give the same loss on the same batch (dropout off).
usage (GPU box): python tools/r04_host_ab.py [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlipv2_amd import blocks, decoder, encoder, linear, parseda, train  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20


def build():
    torch.manual_seed(0)
    margs = parseda.default_args(num_queries=300)
    model, criterion = train.build_training(margs, device="cuda:0", with_text_encoder=True)
    train.to_bf16(model)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    batch = train.synthetic_batch(4, 800, 1333, n_obj=43, n_verb=21, triplets=8, device="cuda:0", seed=0)
    batch[0].tensors = batch[0].tensors.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    step_module = train.ParSeDATrainStep(model)
    model.train()
    train.freeze_parameters_without_gradient(step_module, criterion, batch)
    return model, criterion, step_module, batch


def launches(step_module, criterion, opt, batch):
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        train.train_step(step_module, criterion, opt, batch, autocast_dtype=None)
        torch.cuda.synchronize()
    return sum(1 for e in prof.events() if str(e.device_type).endswith("CUDA"))


def measure(label, setup):
    setup()
    model, criterion, step_module, batch = build()
    opt = train.FusedMasterAdamW(model)
    for _ in range(2):
        train.train_step(step_module, criterion, opt, batch, autocast_dtype=None)
    n_launch = launches(step_module, criterion, opt, batch)
    with torch.no_grad():
        loss = float(criterion.weighted_sum(criterion(step_module(*batch), batch[2])))
    graphed = train.graph_step_module(step_module, model, batch, criterion=criterion)
    for _ in range(3):
        train.train_step(graphed, criterion, opt, batch, autocast_dtype=None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        train.train_step(graphed, criterion, opt, batch, autocast_dtype=None)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print(f"{label:46s} {ms:8.3f} ms/step (graphed)   {n_launch:5d} launches (eager step)   loss after 2 steps {loss:.4f}", flush=True)
    del graphed, model, opt
    torch.cuda.empty_cache()


def product():
    encoder.inplace_tail = True
    encoder.cache_reference_points = True
    blocks.cache_padding_free = True
    parseda.batched_heads = True
    decoder.share_box_deltas = True
    decoder.one_launch_box_head = True
    linear.residual_gradient_in_gemm = True


def off(name):
    def f():
        product()
        mod, attr = name
        setattr(mod, attr, False)
    return f


if __name__ == "__main__":
    measure("product (all restructures on)", product)
    measure("fused tail through cat (encoder.inplace_tail)", off((encoder, "inplace_tail")))
    measure("reference points recomputed", off((encoder, "cache_reference_points")))
    measure("sine encodings recomputed", off((blocks, "cache_padding_free")))
    measure("heads per decoder layer", off((parseda, "batched_heads")))
    measure("box-head MLPs run twice", off((decoder, "share_box_deltas")))
    measure("op-sequence box head", off((decoder, "one_launch_box_head")))
    measure("FFN residual gradient summed by autograd", off((linear, "residual_gradient_in_gemm")))
    measure("product again (box-to-box noise)", product)
