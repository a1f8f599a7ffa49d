"""GPU tests written in round 6 -- again a round without a GPU: never run.  `first_contact` tests (tests/conftest.py): each file's
share runs in one child process with a timeout and is reported as XPASS / XFAIL (tools/gpu_triage_r06.py runs them with
--runxfail: there a failure is a verdict)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.first_contact(timeout=900)
def test_auto_gradient_schedule_on_a_one_rank_rccl_group():
    """train.choose_dp_schedule (bench.py --dp-schedule auto, the default of every N > 1 run) with REAL captured steps on a 1-rank
    RCCL group: the capture self-test, a flat GraphedStep and an overlapped one (collectives captured inside the backward graph),
    the comparison under the same random state with dropout live, the agreement all-reduces.  On RCCL the self-test passes, so the
    overlapped schedule must be chosen unless its first step differs -- either way the chosen step trains (finite loss, gradients
    are views of ITS synchronizer's flat buffer) and the other one is gone.  The 2-rank decision path is tests/test_dp_cpu.py."""
    import torch.distributed as dist
    from rlipv2_amd import parseda, train
    created = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29650 + os.getpid() % 200))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
        created = True
    try:
        torch.manual_seed(0)
        margs = parseda.default_args(num_queries=40, enc_layers=4, dec_layers=2)
        model, criterion = train.build_training(margs, device=DEV, with_text_encoder=True)
        train.to_bf16(model)
        batch = train.synthetic_batch(2, 256, 320, device=DEV, triplets=3)
        batch[0].tensors = batch[0].tensors.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        step = train.ParSeDATrainStep(model)
        model.train()                                              # dropouts live, as in bench.py
        train.broadcast_parameters(model, 0)
        train.freeze_parameters_without_gradient(step, criterion, batch)
        params = [p for p in step.parameters() if p.requires_grad]
        built, logs = [], []

        def build(overlap):
            sync = train.GradientSynchronizer(params, bucket_bytes=32 << 20)
            sync.scale_in_optimizer = True
            built.append(overlap)
            return train.graph_step_module(step, model, batch, sync, criterion=criterion, overlap=overlap)

        before = torch.cuda.get_rng_state(DEV).clone()
        assert train.captured_collective_selftest(DEV)            # RCCL: captured collectives replay
        chosen, schedule, reason = train.choose_dp_schedule(build, batch, DEV, log=logs.append)
        print(schedule, "--", reason, "|", logs)
        assert built == [False, True] and schedule in ("overlapped", "flat") and chosen.overlap == (schedule == "overlapped")
        assert schedule == "overlapped", reason                    # (a 1-rank all-reduce is the identity: the two steps are the same step)
        assert torch.equal(before, torch.cuda.get_rng_state(DEV)) and all(p.grad is None for p in params)
        opt = train.FusedMasterAdamW(model)
        for _ in range(2):
            loss = train.train_step(chosen, criterion, opt, batch, autocast_dtype=None)
        assert torch.isfinite(loss)
        flat = chosen.synchronizer.flat
        lo, hi = flat.data_ptr(), flat.data_ptr() + flat.numel() * flat.element_size()
        assert all(p.grad is not None and lo <= p.grad.data_ptr() < hi for p in params)
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize("D", [2048, 3096])
def test_reference_gradcheck_recipe_large_channels(D):
    """models/ops/test.py:67-82 with the two largest channel counts of test.py:89-90 (tests/test_msda_gpu.py runs 30 ... 1025: the
    same generic kernel, validated in round 2) -- through gradcheck's fast mode: the full Jacobian of `value` alone would be a
    185 760 x 12 384 float64 matrix, twice.  Placed in this late file so that the first run of a new size cannot hide the suite
    behind it under -x."""
    from rlipv2_amd import msda
    from test_msda_gpu import _testpy_inputs
    value, shapes, starts, loc, aw = _testpy_inputs(D, torch.float64)
    value.requires_grad_(True); loc.requires_grad_(True); aw.requires_grad_(True)
    assert torch.autograd.gradcheck(msda.MSDeformAttnFunction.apply, (value, shapes, starts, loc, aw, 2), fast_mode=True)
