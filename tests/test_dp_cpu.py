"""Data-parallel path on CPU: 2 ranks over gloo reproduce the single-process gradients.

The path shards by image: every rank holds a full replica and its own images; the
criterion's interaction count is all-reduced (it sets the loss scale, reference hoi.py:4737-4740)
and DDP averages the gradients.  With equal shard sizes the averaged 2-rank gradient equals the
gradient of the same global batch in one process -- that is what is asserted here, on a small
model with the oracle-backed op standing in for the HIP kernels.
"""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def _build():
    from rlipv2_amd import parseda, train          # (the op runs on the product's CPU arm, csrc/msda_cpu.cpp)
    args = parseda.default_args(num_queries=12, enc_layers=4, dec_layers=2, dim_feedforward=128, pseudo_verb=False)
    torch.manual_seed(1234)            # identical replicas in every process
    model, crit = train.build_training(args, device="cpu", with_text_encoder=False)
    model.eval()                       # dropout-free, deterministic
    return model, crit, train


def _batch(train, n_images, seed):
    samples, _, targets = train.synthetic_batch(n_images, 64, 96, n_obj=6, n_verb=4, triplets=2, device="cpu", seed=seed)
    for k, t in enumerate(targets):            # exactly one verb per triplet: the focal loss normalises by the
        t["verb_labels"] = torch.eye(4)[[(seed + k) % 4, (seed + k + 1) % 4]]   # LOCAL positive count (as the reference)
    g = torch.Generator().manual_seed(99)
    mem = torch.tanh(torch.randn(10, 1, 768, generator=g)).repeat(1, n_images, 1)
    text = (~(mem.sum(-1) > 0), mem, torch.tensor([[6, 4]]))
    return samples, text, targets


def _grads(model):
    return {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    model, crit, train = _build()
    samples, text, targets = _batch(train, 2, seed=100 + rank)           # this rank's image shard
    plain = train.ParSeDATrainStep(model)
    train.freeze_parameters_without_gradient(plain, crit, (samples, text, targets))
    step = torch.nn.parallel.DistributedDataParallel(plain, find_unused_parameters=False)
    outputs = step(samples, text, targets)
    loss = crit.weighted_sum(crit(outputs, targets))
    loss.backward()
    if rank == 0:
        torch.save(_grads(model), os.path.join(out_dir, "dp_grads.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradients_equal_single_process(tmp_path):
    port = 29500 + os.getpid() % 1000
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    dp = torch.load(os.path.join(str(tmp_path), "dp_grads.pt"))

    model, crit, train = _build()
    from rlipv2_amd.blocks import NestedTensor
    parts = [_batch(train, 2, seed=100 + r) for r in range(2)]
    samples = NestedTensor(torch.cat([p[0].tensors for p in parts]), torch.cat([p[0].mask for p in parts]))
    mem = parts[0][1][1][:, :1].repeat(1, 4, 1)
    text = (~(mem.sum(-1) > 0), mem, torch.tensor([[6, 4]]))
    targets = parts[0][2] + parts[1][2]
    outputs = train.ParSeDATrainStep(model)(samples, text, targets)
    crit.weighted_sum(crit(outputs, targets)).backward()
    ref = _grads(model)
    assert set(ref) == set(dp)
    worst, who = 0.0, None
    gmax = max(float(v.abs().max()) for v in ref.values())
    for n in ref:
        err = float((ref[n] - dp[n]).abs().max()) / max(1e-3 * gmax, float(ref[n].abs().max()))
        if err > worst:
            worst, who = err, n
    ratios = {n: float(dp[n].norm() / ref[n].norm().clamp(min=1e-12)) for n in list(ref)[:400:25]}
    assert worst < 2e-3, (worst, who, ratios)


def test_statically_unused_parameters_are_frozen():
    """The verb decoder's box heads only feed detached reference points: freezing them
    is the static mask that lets DDP run without find_unused_parameters."""
    model, crit, train = _build()
    samples, text, targets = _batch(train, 2, seed=5)
    # with the script's decoder depth (> 1 layer) the hand-written mask is complete: the dry run finds nothing more
    assert train.freeze_parameters_without_gradient(train.ParSeDATrainStep(model), crit, (samples, text, targets)) == []
    outputs = train.ParSeDATrainStep(model)(samples, text, targets)
    crit.weighted_sum(crit(outputs, targets)).backward()
    for n, p in model.named_parameters():
        if p.requires_grad:
            assert p.grad is not None, f"{n} is trainable but got no gradient"
    n_pred = model.transformer.ho_decoder.num_layers
    assert all(not p.requires_grad for h in list(model.sub_bbox_embed)[n_pred:] for p in h.parameters())


def _sync_worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rlipv2_amd import train
        torch.manual_seed(100 + rank)                      # different initial weights: the broadcast must fix that
        net = torch.nn.Sequential(torch.nn.Conv2d(4, 8, 3), torch.nn.Flatten(), torch.nn.Linear(8 * 6 * 6, 5))
        net[0].to(memory_format=torch.channels_last)
        train.broadcast_parameters(net, 0)
        params = list(net.parameters())
        sync = train.GradientSynchronizer(params)
        g = torch.Generator().manual_seed(7 + rank)
        grads = [torch.randn(p.shape, generator=g) for p in params]
        grads[0] = grads[0].contiguous(memory_format=torch.channels_last)
        grads[3] = None                                    # a parameter without gradient on this step
        avg = sync(grads)                                  # (gloo has no ReduceOp.AVG: exercises the SUM + scale route)
        out[rank] = ([p.detach().clone() for p in params], [a.clone() for a in avg],
                     [None if gr is None else gr.clone() for gr in grads])
    finally:
        dist.destroy_process_group()


def test_gradient_synchronizer_and_broadcast_two_ranks():
    """train.broadcast_parameters + train.GradientSynchronizer (the data-parallel path of the graphed step):
    equal parameters on both ranks afterwards; averaged gradients = mean of the ranks' gradients, in the
    parameter's own memory layout; a missing gradient counts as zero."""
    import threading
    out = {}
    port = 29500 + (os.getpid() % 400) + 37
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    shared = mgr.dict()
    procs = [mp.Process(target=_sync_worker, args=(r, 2, port, shared)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    (p0, a0, g0), (p1, a1, g1) = shared[0], shared[1]
    for x, y in zip(p0, p1):
        assert torch.equal(x, y)
    for i, (x, y) in enumerate(zip(a0, a1)):
        assert torch.equal(x, y)
        ga = torch.zeros_like(x) if g0[i] is None else g0[i]
        gb = torch.zeros_like(x) if g1[i] is None else g1[i]
        torch.testing.assert_close(x, (ga + gb) / 2)
        assert x.stride() == p0[i].stride()


def _overlap_worker(rank, world, port, out_dir, wide):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    model, crit, train = _build()
    batch = _batch(train, 2, seed=100 + rank)                            # this rank's image shard
    step = train.ParSeDATrainStep(model)
    train.freeze_parameters_without_gradient(step, crit, batch)
    params = [p for p in step.parameters() if p.requires_grad]
    # wide: the reduction runs on a wider copy of each bucket (the switchable reduction dtype of the bf16 train step)
    sync = train.GradientSynchronizer(params, reduce_dtype=torch.float64 if wide else None, bucket_bytes=1 << 20)
    launched = []
    real_launch = sync.launch_bucket
    sync.launch_bucket = lambda k, streams=(): (launched.append(k), real_launch(k, streams))[1]
    for attempt in range(2):                # first pass records the arrival order, second one runs bucketed
        outputs = step(*batch)
        total = crit.weighted_sum(crit(outputs, batch[2]))
        if attempt == 0:
            with sync.recording() as rec:
                torch.autograd.grad(total, params, allow_unused=True)
            order = [rec.order]
            dist.broadcast_object_list(order, src=0)
            sync.plan_buckets(order[0])
            continue
        with sync.hooked():
            torch.autograd.grad(total, params, allow_unused=True)
    if rank == 0:
        names = [n for n, p in step.named_parameters() if p.requires_grad]
        torch.save({"grads": {n: v.clone() for n, v in zip(names, sync.views)}, "launched": launched,
                    "buckets": [len(b) for b in sync.buckets], "first": [names[i] for i in sync.buckets[0][:3]]},
                   os.path.join(out_dir, "overlap_grads.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("wide", [False, True])
def test_bucketed_overlapped_all_reduce_two_ranks(tmp_path, wide):
    """GradientSynchronizer.hooked(): tensor hooks copy every gradient into the flat buffer as autograd produces it and
    the hook completing a bucket starts that bucket's (asynchronous) all-reduce -- the schedule the graphed
    data-parallel step captures into its backward graph.  2 gloo ranks: the averaged gradients equal the single-process
    gradients of the global batch; the buckets go out in arrival order, the decoders' / heads' gradients first."""
    port = 29500 + os.getpid() % 1000 + 5 + int(wide)
    mp.spawn(_overlap_worker, args=(2, port, str(tmp_path), wide), nprocs=2, join=True)
    res = torch.load(os.path.join(str(tmp_path), "overlap_grads.pt"))
    dp = res["grads"]
    assert len(res["buckets"]) >= 3 and res["launched"] == sorted(res["launched"]) and len(res["launched"]) == len(res["buckets"])
    assert all("backbone" not in n and "encoder" not in n for n in res["first"]), res["first"]

    model, crit, train = _build()
    from rlipv2_amd.blocks import NestedTensor
    parts = [_batch(train, 2, seed=100 + r) for r in range(2)]
    samples = NestedTensor(torch.cat([p[0].tensors for p in parts]), torch.cat([p[0].mask for p in parts]))
    mem = parts[0][1][1][:, :1].repeat(1, 4, 1)
    text = (~(mem.sum(-1) > 0), mem, torch.tensor([[6, 4]]))
    targets = parts[0][2] + parts[1][2]
    step = train.ParSeDATrainStep(model)
    train.freeze_parameters_without_gradient(step, crit, (samples, text, targets))
    outputs = step(samples, text, targets)
    crit.weighted_sum(crit(outputs, targets)).backward()
    ref = {n: p.grad.clone() for n, p in step.named_parameters() if p.grad is not None}
    assert set(ref) == set(dp)
    gmax = max(float(v.abs().max()) for v in ref.values())
    worst = max(float((ref[n] - dp[n]).abs().max()) / max(1e-3 * gmax, float(ref[n].abs().max())) for n in ref)
    assert worst < 2e-3, worst


def _cache_worker(rank, world, port, out_dir, overlap):
    """A data-parallel run in which the ranks meet DIFFERENT batch buckets at different steps: GraphedStepCache decides hit /
    miss / capture / eviction locally on every rank, and every way of running a step issues the same collectives."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    model, crit, train = _build()
    step = train.ParSeDATrainStep(model)
    shapes = {"A": (64, 96), "B": (64, 64), "C": (96, 64)}

    def batch(name, seed):
        h, w = shapes[name]
        samples, _, targets = train.synthetic_batch(2, h, w, n_obj=6, n_verb=4, triplets=2, device="cpu", seed=seed)
        for k, t in enumerate(targets):
            t["verb_labels"] = torch.eye(4)[[(seed + k) % 4, (seed + k + 1) % 4]]
        g = torch.Generator().manual_seed(99)
        mem = torch.tanh(torch.randn(10, 1, 768, generator=g)).repeat(1, 2, 1)
        return samples, (~(mem.sum(-1) > 0), mem, torch.tensor([[6, 4]])), targets

    train.freeze_parameters_without_gradient(step, crit, batch("A", 1))
    params = [p for p in step.parameters() if p.requires_grad]
    sync = train.GradientSynchronizer(params, bucket_bytes=1 << 20)

    class Captured(train.EagerSyncStep):            # stand-in for a GraphedStep (HIP graphs need a GPU): same protocol, and
        pass                                        # like a real capture its construction issues no collective

    made = []

    def factory(b):
        made.append(train.GraphedStepCache.bucket(b)[0][2:])
        return Captured(step, crit, sync, overlap=overlap)

    cache = train.GraphedStepCache(step, model, sync, criterion=crit, max_buckets=2, capture_after=2, overlap=overlap,
                                   factory=factory)
    # rank 0 cycles through three buckets with room for two captures (evictions, re-captures); rank 1 stays on two
    seq = ["A", "A", "B", "B", "C", "C", "A", "A"] if rank == 0 else ["B", "A", "B", "A", "B", "A", "B", "A"]

    class NoOptimizer:
        @staticmethod
        def zero_grad(set_to_none=True):
            for p in params:
                p.grad = None

    worst = 0.0
    for it, name in enumerate(seq):
        b = batch(name, 100 * it + rank)
        # this rank's own gradient of the batch, then the expected average over the ranks
        out = step(*b)
        local = torch.autograd.grad(crit.weighted_sum(crit(out, b[2])), params, allow_unused=True)   # (one count all-reduce)
        flat = torch.cat([(torch.zeros_like(p) if g is None else g).reshape(-1) for p, g in zip(params, local)])
        both = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(both, flat)
        expect = sum(both) / world
        loss, scale = train._forward_backward(cache, crit, NoOptimizer, b, None)
        got = torch.cat([p.grad.reshape(-1) for p in params]) * scale
        worst = max(worst, float((got - expect).abs().max()) / float(expect.abs().max()))
    torch.save({"worst": worst, "captures": cache.captures, "evictions": cache.evictions, "eager": cache.eager_steps,
                "hits": cache.hits, "made": made, "planned": sync.planned, "buckets": len(sync.buckets)},
               os.path.join(out_dir, f"cache_{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [False, True])
def test_ranks_with_different_bucket_sequences_two_ranks(tmp_path, overlap):
    """GraphedStepCache under data parallelism (reference: DistributedDataParallel takes any batch shapes, main.py:515-517):
    two gloo ranks see different bucket sequences, one of them overflows `max_buckets` and evicts; no rank ever asks the
    other what it is about to do.  Every step's synchronised gradient equals the average of the two ranks' own gradients,
    on the flat schedule and on the bucketed one (whose plan is fixed collectively after step 0)."""
    port = 29500 + os.getpid() % 1000 + 11 + int(overlap)
    mp.spawn(_cache_worker, args=(2, port, str(tmp_path), overlap), nprocs=2, join=True)
    r0, r1 = (torch.load(os.path.join(str(tmp_path), f"cache_{r}.pt")) for r in range(2))
    assert r0["worst"] < 1e-5 and r1["worst"] < 1e-5, (r0["worst"], r1["worst"])
    # rank 0: A, B captured at their second sight, C evicts A, A is captured again on its 4th sight -> 4 captures, 2 evictions
    assert r0["captures"] == 4 and r0["evictions"] == 2, r0
    assert r1["captures"] == 2 and r1["evictions"] == 0 and r1["hits"] >= 2, r1
    assert r0["captures"] + r0["hits"] + r0["eager"] == 8 and r1["captures"] + r1["hits"] + r1["eager"] == 8
    if overlap:
        assert r0["planned"] and r1["planned"] and r0["buckets"] == r1["buckets"] >= 2


def _sum_worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rlipv2_amd import train
        net = torch.nn.Linear(6, 3)
        sync = train.GradientSynchronizer(list(net.parameters()))
        # the SUM route of the fused optimiser: the buffer keeps the sum, the factor is handed over
        sync.scale_in_optimizer = True
        g = torch.Generator().manual_seed(3 + rank)
        grads = [torch.randn(p.shape, generator=g) for p in net.parameters()]
        views = sync(grads)
        res = {"sum": ([v.clone() for v in views], [x.clone() for x in grads], sync.grad_scale)}
        # ... which only FusedMasterAdamW knows how to apply: any other optimiser is refused instead of stepping on the SUM
        try:
            train._optimizer_step(torch.optim.SGD(net.parameters(), lr=0.1), 0.1, sync.grad_scale)
            res["refused"] = False
        except RuntimeError as e:
            res["refused"] = "scale_in_optimizer" in str(e)
        out[rank] = res
    finally:
        dist.destroy_process_group()


def test_sum_route_two_ranks():
    """GradientSynchronizer with scale_in_optimizer: the flat buffer holds the SUM over ranks and grad_scale = 1 / world goes
    to the optimiser; an optimiser that cannot apply it is refused."""
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 400) + 61
    shared = mp.Manager().dict()
    procs = [mp.Process(target=_sum_worker, args=(r, 2, port, shared)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    (v0, g0, s0), (v1, g1, s1) = shared[0]["sum"], shared[1]["sum"]
    assert s0 == s1 == 0.5 and shared[0]["refused"] and shared[1]["refused"]
    for a, b, x, y in zip(v0, v1, g0, g1):
        assert torch.equal(a, b)
        torch.testing.assert_close(a, x + y)


def test_one_gradient_schedule_for_captured_and_eager_steps():
    """Advisor finding of round 4: GraphedStep falls back to the flat schedule when the group's collectives cannot be captured
    (anything but RCCL), the cache's eager step must then run the flat schedule too -- a rank replaying a capture and a rank
    on the eager path would otherwise issue different collectives.  Resolved once (train.effective_overlap) and checked at
    every capture."""
    from rlipv2_amd import train

    class Sync:
        group = None
        planned = False
        params = []

    assert train.effective_overlap(True, None) is False
    assert train.effective_overlap(False, Sync()) is False
    assert train.effective_overlap(True, Sync()) is False            # no process group here: nothing can be captured
    cache = train.GraphedStepCache(torch.nn.Linear(2, 2), None, Sync(), overlap=True)
    assert cache.overlap is False and cache.eager.overlap is False

    class Mismatch:
        overlap = True

    cache = train.GraphedStepCache(torch.nn.Linear(2, 2), None, Sync(), overlap=False, factory=lambda b: Mismatch())
    cache.bucket = staticmethod(lambda b: "k")
    with pytest.raises(RuntimeError, match="disagree on the gradient schedule"):
        cache.get(None)


def test_graphed_step_redelivers_gradients_somebody_cleared():
    """Advisor finding of round 4: GraphedStep._deliver samples the first and last parameter's `.grad`; a `.grad` cleared in
    the middle (partial zero_grad, a group frozen mid-run) is re-attached at the next full check (every VERIFY_EVERY-th step)."""
    from rlipv2_amd import train
    g = train.GraphedStep.__new__(train.GraphedStep)
    g.synchronizer, g.overlap = None, False
    g.params = [torch.nn.Parameter(torch.zeros(2)) for _ in range(5)]
    g.static_grads = [torch.ones(2) for _ in range(5)]
    g._deliver()
    assert all(p.grad is s for p, s in zip(g.params, g.static_grads))
    g.params[2].grad = None                                          # somebody cleared one in the middle
    for _ in range(train.GraphedStep.VERIFY_EVERY):
        g._deliver()
    assert g.params[2].grad is g.static_grads[2]
    g.params[0].grad = None                                          # the sampled ones are noticed at once
    g._deliver()
    assert g.params[0].grad is g.static_grads[0]


def _schedule_worker(rank, world, port, out_dir, case):
    """train.choose_dp_schedule (bench.py --dp-schedule auto) on two gloo ranks: the decision path itself -- self-test verdict,
    flat probe, overlapped probe, comparison, agreement -- with EagerSyncStep standing in for the captured steps (HIP graphs need
    a GPU; the eager bucketed schedule issues the same collectives and runs on any backend)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    model, crit, train = _build()
    batch = _batch(train, 2, seed=100 + rank)
    step = train.ParSeDATrainStep(model)
    train.freeze_parameters_without_gradient(step, crit, batch)
    params = [p for p in step.parameters() if p.requires_grad]
    built = []

    def build_step(overlap):
        built.append(overlap)
        sync = train.GradientSynchronizer(params, bucket_bytes=1 << 20)
        s = train.EagerSyncStep(step, crit, sync, overlap=overlap)
        if overlap:                                   # (a GraphedStep plans in its constructor: rank 0's arrival order for all)
            s.run(*batch)
            sync.plan_collectively(s.arrival)
            assert sync.planned and len(sync.buckets) >= 2
            if case == "overlapped_wrong_on_rank1" and rank == 1:
                real = s.run

                def wrong(*b):                        # a schedule that loses a bucket's worth of gradient on ONE rank
                    out = real(*b)
                    for p in params[:40]:
                        if p.grad is not None:
                            p.grad.mul_(0.0)
                    return out
                s.run = wrong
        return s

    selftest = {"agree": lambda device, group=None: True,
                "overlapped_wrong_on_rank1": lambda device, group=None: True,
                "selftest_fails_on_rank1": lambda device, group=None: rank == 0,
                "real_selftest_on_gloo": None}[case]
    before = torch.get_rng_state().clone()
    logs = []
    chosen, schedule, reason = train.choose_dp_schedule(build_step, batch, "cpu", selftest=selftest, log=logs.append)
    assert torch.equal(before, torch.get_rng_state()) and all(p.grad is None for p in params)
    # the chosen step works: its synchronised gradient is the average of the ranks' own gradients
    out = step(*batch)
    local = torch.autograd.grad(crit.weighted_sum(crit(out, batch[2])), params, allow_unused=True)
    flat = torch.cat([(torch.zeros_like(p) if g is None else g).reshape(-1) for p, g in zip(params, local)])
    both = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    chosen.run(*batch)
    got = torch.cat([p.grad.reshape(-1) for p in params]) * chosen.grad_scale
    err = float((got - sum(both) / world).abs().max()) / float((sum(both) / world).abs().max())
    torch.save({"schedule": schedule, "reason": reason, "built": built, "overlap": bool(chosen.overlap), "err": err, "logs": logs},
               os.path.join(out_dir, f"schedule_{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case", ["agree", "overlapped_wrong_on_rank1", "selftest_fails_on_rank1", "real_selftest_on_gloo"])
def test_auto_gradient_schedule_decision_two_ranks(tmp_path, case):
    """the schedule is `auto` -- overlapped iff captured collectives replay on ALL ranks and
    the first overlapped step's loss and gradient norm equal the flat schedule's to 1e-3 on ALL ranks, else flat; both ranks
    always reach the same decision, and the chosen step delivers the average of the ranks' gradients."""
    port = 29500 + os.getpid() % 1000 + 23 + ["agree", "overlapped_wrong_on_rank1", "selftest_fails_on_rank1", "real_selftest_on_gloo"].index(case)
    mp.spawn(_schedule_worker, args=(2, port, str(tmp_path), case), nprocs=2, join=True)
    r0, r1 = (torch.load(os.path.join(str(tmp_path), f"schedule_{r}.pt")) for r in range(2))
    assert r0["schedule"] == r1["schedule"] and r0["overlap"] == r1["overlap"]
    assert r0["err"] < 1e-5 and r1["err"] < 1e-5, (r0["err"], r1["err"])
    if case == "agree":
        assert r0["schedule"] == "overlapped" and r0["overlap"] and r0["built"] == [False, True] == r1["built"]
        assert "equal" in r0["logs"][0] and "equal" in r1["logs"][0]
    elif case == "overlapped_wrong_on_rank1":
        assert r0["schedule"] == "flat" and not r0["overlap"] and r0["built"] == [False, True]
        assert "another rank" in r0["reason"] and "differs from the flat one" in r1["reason"]
    else:                                             # no capture -> the overlapped step is never even built
        assert r0["schedule"] == "flat" and r0["built"] == [False] == r1["built"] and "self-test" in r0["reason"]
