"""The train step of RLIPv2-ParSeDA and its data-parallel wrapper (the protocol of the reference's
engine.train_one_epoch, engine.py:45-201, and main.py:505-539).

  step = phase A (backbone + ALIF encoder) -> phase B (decoders + heads) -> SetCriterionHOI ->
         weighted loss -> backward -> clip_grad_norm_(0.1) -> AdamW (3 parameter groups by name:
         rest / "backbone" / "text_encoder", main.py:523-539)

Data parallelism: one process per GPU, full replica, image batch sharded, gradient
all-reduce over RCCL (torch.distributed backend "nccl") bucketed and overlapped with backward by
DistributedDataParallel.  Two departures from the reference, neither changes the gradients:
  * both model phases run inside ONE wrapped forward, so the DDP hooks fire once per step (the
    reference calls the DDP module twice per backward);
  * the parameters that can never receive a gradient -- the box heads handed to the verb decoder,
    whose outputs only feed detached reference points -- are frozen up front, a static
    unused-parameter mask, instead of `find_unused_parameters=True` graph walks every step.
"""
from __future__ import annotations

import math
import os

import torch
import torch.distributed as dist
from torch import nn

from . import criterion as crit_mod
from .linear import swap_linears
from .optim import FusedMasterAdamW
from .backbone import build_r50_backbone
from .blocks import NestedTensor
from .parseda import build_parseda, default_args


class TextEncoderStub(nn.Module):
    """RoBERTa-base shaped text encoder with random weights (no checkpoints offline), built from
    the installed `transformers` package when available."""

    def __init__(self):
        super().__init__()
        from transformers import RobertaConfig, RobertaModel
        cfg = RobertaConfig(vocab_size=50265, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                            intermediate_size=3072, hidden_act="gelu", hidden_dropout_prob=0.1,
                            attention_probs_dropout_prob=0.1, max_position_embeddings=514, type_vocab_size=1,
                            layer_norm_eps=1e-5, pad_token_id=1, bos_token_id=0, eos_token_id=2)
        self.model = RobertaModel(cfg)

    def forward(self, input_ids, attention_mask):
        return self.model(input_ids=input_ids, attention_mask=attention_mask)


class ParSeDATrainStep(nn.Module):
    """Both phases of the model in one forward (what DDP wraps)."""

    def __init__(self, model):
        super().__init__()
        self.model = model

    def forward(self, samples, text, targets):
        memory_cache = self.model(samples, encode_and_save=True, text=text, targets=targets)
        return self.model(samples, encode_and_save=False, memory_cache=memory_cache, text=text, targets=targets)


OUT_KEYS = ("pred_sub_logits", "pred_obj_logits", "pred_verb_logits", "pred_sub_boxes", "pred_obj_boxes")


class GraphedTrainForward(nn.Module):
    """The two model phases behind a tensors-in / tensors-out signature, captured as HIP graphs
    (forward graph + backward graph, torch.cuda.make_graphed_callables).

    Why: the step issues ~8 400 kernel launches, most of them a few microseconds long; in eager
    mode the Python/HIP launch path (~10 us per launch) is slower than the GPU (measured: 103 ms
    wall vs 86 ms of kernel time at batch 4).  Shapes are static in training with fixed-size
    batches, the MSDA library only launches kernels and one memset on the current stream, and the
    pyramid metadata stays on the device -- so both phases replay as two graph launches.  The
    criterion (host-side Hungarian assignment) and the optimiser stay outside the graphs."""

    def __init__(self, step_module, obj_pred_names_sums, n_layers, pseudo_verb):
        super().__init__()
        self.step = step_module
        self.sums = obj_pred_names_sums
        self.n_layers = n_layers
        self.pseudo_verb = pseudo_verb
        self.no_padding = False

    def forward(self, images, mask, input_ids, attention_mask, verb_labels):
        text = {"input_ids": input_ids, "attention_mask": attention_mask, "obj_pred_names_sums": self.sums}
        out = self.step(NestedTensor(images, mask, self.no_padding), text, [{"verb_labels": verb_labels}])
        flat = [out[k] for k in OUT_KEYS]
        for aux in out.get("aux_outputs", []):
            flat += [aux[k] for k in OUT_KEYS]
        if self.pseudo_verb:
            flat.append(out["target_verb_sim"])
        return tuple(flat)

    def unflatten(self, flat):
        n = len(OUT_KEYS)
        out = dict(zip(OUT_KEYS, flat[:n]))
        aux = [dict(zip(OUT_KEYS, flat[n * (i + 1): n * (i + 2)])) for i in range(self.n_layers - 1)]
        if self.pseudo_verb:
            tvs = flat[-1]
            out["target_verb_sim"] = tvs
            for a in aux:
                a["target_verb_sim"] = tvs
        if aux:
            out["aux_outputs"] = aux
        return out


def effective_overlap(overlap, synchronizer):
    """The overlapped (bucketed, in-graph) schedule is only available when the collectives of the synchronizer's group can be
    captured into a HIP graph, i.e. on RCCL.  Resolved by ONE rule for every kind of step (GraphedStep, EagerSyncStep,
    GraphedStepCache): ranks that mix a replayed capture with an eager step must still issue the same collectives."""
    if not overlap or synchronizer is None:
        return False
    import torch.distributed as dist
    return bool(dist.is_available() and dist.is_initialized() and dist.get_backend(synchronizer.group) == "nccl")


class GraphedStep:
    """Forward graph + backward graph of the two model phases with the parameter gradients delivered
    directly, outside autograd.

    torch.cuda.make_graphed_callables routes the ~260 parameter gradients back through an autograd node;
    each one then passes an AccumulateGrad node whose recorded stream (the capture side stream) differs from
    the replay stream, so the engine creates / records / waits on an event per parameter and clones the
    static gradient -- measured with rocprofv3 --hip-trace: 747 hipEventRecord + 420 hipStreamWaitEvent and
    ~5 ms of idle GPU after every backward replay (hipGraphLaunch itself returns only when the graph is
    nearly done, so none of that host work is hidden).  Here the backward graph writes the gradients into
    static buffers and `backward()` simply points `p.grad` at them: stable addresses, no copies, no events.

    Protocol (train_step): `out = step(samples, text, targets)` returns detached leaf tensors; the criterion
    runs on them eagerly; `loss.backward()` fills their `.grad`; `step.backward()` replays the model's
    backward graph from those."""

    _delivered_by = None          # weak reference to the GraphedStep whose static buffers the parameters' `.grad` point at
    VERIFY_EVERY = 16             # deliveries between full checks that every parameter's `.grad` is still the static buffer

    def __init__(self, step_module, model, batch, warmup=3, synchronizer=None, criterion=None, overlap=False):
        """`overlap` (off until a run on >= 2 GPUs has verified it; bench.py --dp-overlap): the gradient all-reduce is
        bucketed and captured INSIDE the backward graph on a communication stream, each bucket starting as soon as its
        last gradient exists (GradientSynchronizer.hooked) -- averaging overlapped with the rest of the backward pass
        (reference main.py:515-517).  Off: one flat all-reduce after the backward replay.

        A capture issues NO collective of its own: the warm-up steps use this rank's interaction count, collectives inside
        the captured backward are recorded, not run.  A rank may therefore capture a new batch bucket while the other ranks
        replay theirs -- they simply wait at the step's first collective (GraphedStepCache).  The one exception is the bucket
        plan of the overlapped schedule (rank 0's gradient arrival order, broadcast once): if the synchronizer has no plan
        yet, every rank must construct its first GraphedStep together (GraphedStepCache plans at the end of step 0)."""
        samples, text, targets = batch
        self.step_module = step_module
        self.synchronizer = synchronizer
        self.criterion = criterion
        self.overlap = effective_overlap(overlap, synchronizer)    # only RCCL collectives can be captured into the graph
        self.wrapper = GraphedTrainForward(step_module, text["obj_pred_names_sums"],
                                           model.transformer.ho_decoder.num_layers, model.pseudo_verb)
        self.wrapper.no_padding = bool(getattr(samples, "no_padding", False))      # baked into the capture
        self.params = [p for p in step_module.parameters() if p.requires_grad]
        verbs = torch.cat([t["verb_labels"] for t in targets])
        self.static_in = [samples.tensors, samples.mask, text["input_ids"], text["attention_mask"], verbs]
        # With a criterion the graphs hold the whole step except the host-side assignment:
        #   forward graph  = model phases + stacked predictions + matcher cost matrices (criterion.prepare)
        #   host           = ONE device->host copy, scipy assignment, ONE host->device copy (criterion.assign)
        #   backward graph = losses (criterion.losses) + their backward + the model's backward (+ packing)
        # so the ~110 + ~150 launch-bound eager kernels of the criterion and its backward become graph nodes.
        # The target tensors are static inputs; the capture is specific to the number of targets per image.
        self.static_targets = [{k: v.clone() for k, v in t.items() if torch.is_tensor(v)} for t in targets]
        self.sizes = [len(t["obj_labels"]) for t in targets]

        def forward_part():
            outs = self.wrapper(*self.static_in)
            state = None
            if criterion is not None:
                state = criterion.prepare(self.wrapper.unflatten(outs), self.static_targets)
            return outs, state

        def backward_of(outputs, grad_outputs, mode):
            """mode "plain": gradients as autograd returns them; "record": additionally note their arrival order;
            "sync": deliver them through the synchronizer (bucketed + overlapped, or packed for the flat schedule)"""
            if mode == "record":
                with synchronizer.recording() as rec:
                    grads = torch.autograd.grad(outputs, self.params, grad_outputs, allow_unused=True)
                self.arrival = rec.order
                return grads
            if mode == "sync" and self.overlap:
                with synchronizer.hooked():
                    torch.autograd.grad(outputs, self.params, grad_outputs, allow_unused=True)
                return synchronizer.views
            grads = torch.autograd.grad(outputs, self.params, grad_outputs, allow_unused=True)
            return synchronizer.pack(grads) if (mode == "sync" and synchronizer is not None) else grads

        def loss_and_grads(outs, state, index, num, mode="plain"):
            if criterion is None:
                need = [o for o in outs if o.requires_grad]
                return None, None, backward_of(need, [torch.ones_like(o) for o in need], mode)
            ld = criterion.losses(state, index, num)
            total = criterion.weighted_sum(ld)
            return ld, total, backward_of(total, None, mode)

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for it in range(warmup):
                wmode = "record" if (self.overlap and not synchronizer.planned and it == warmup - 1) else "plain"
                outs, state = forward_part()
                if criterion is not None:
                    index = criterion.assign(state).to(samples.tensors.device)
                    # (local count: run() overwrites static_num with the all-reduced one on every step)
                    num = criterion._num_interactions(self.sizes, samples.tensors.device, local=True).reshape(1)
                    loss_and_grads(outs, state, index, num[0], wmode)
                else:
                    loss_and_grads(outs, None, None, None, wmode)
            del outs, state
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if self.overlap and not synchronizer.planned:
            # buckets in the order the gradients arrive; rank 0's order for everybody (the schedule of collectives
            # has to be the same on all ranks).  One layout per synchronizer: earlier captures have the flat buffer's
            # offsets baked into their copies and collectives -- a later capture keeps the plan, whatever order its own
            # gradients arrived in, and does not communicate.
            synchronizer.plan_collectively(self.arrival)
        pool = torch.cuda.graph_pool_handle()
        # "thread_local": other threads of the process may call HIP while we capture -- with a process group
        # alive, RCCL's watchdog thread polls events, which in the default "global" mode invalidates the capture
        # (observed: abort in capture_end, depending on timing)
        mode = "thread_local"
        self.fwd_graph, self.bwd_graph = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.fwd_graph, pool=pool, capture_error_mode=mode):
            self.static_out, self.state = forward_part()
        self.leaves = None
        if criterion is None:
            self.diff = [i for i, o in enumerate(self.static_out) if o.requires_grad]
            self.static_gout = [torch.zeros_like(self.static_out[i]) for i in self.diff]
            with torch.cuda.graph(self.bwd_graph, pool=pool, capture_error_mode=mode):
                # (with a synchronizer the packing copies -- and, overlapped, the collectives -- are graph nodes)
                grads = backward_of([self.static_out[i] for i in self.diff], self.static_gout, "sync")
            self.static_grads = grads
            return
        self.static_index = index.clone()                      # [2, K * matched pairs]: shape fixed by `sizes`
        self.static_num = num.clone()
        self.pinned_index = torch.empty(index.shape, dtype=index.dtype, pin_memory=True)
        # the cost matrices leave through a pinned staging buffer (a pageable .cpu() was 0.3 ms of idle GPU per step)
        C = self.state['C']
        self.pinned_cost = torch.empty(C.shape, dtype=torch.float32, pin_memory=True) if C.is_cuda else None
        with torch.cuda.graph(self.bwd_graph, pool=pool, capture_error_mode=mode):
            self.loss_dict, self.total, grads = loss_and_grads(self.static_out, self.state, self.static_index,
                                                                self.static_num[0], "sync")
        self.static_grads = grads

    def _cost_on_host(self):
        if self.pinned_cost is None:
            return None
        self.pinned_cost.copy_(self.state['C'], non_blocking=True)
        torch.cuda.current_stream().synchronize()
        return self.pinned_cost

    def _load_inputs(self, samples, text, targets):
        if bool(getattr(samples, "no_padding", False)) != self.wrapper.no_padding:
            raise RuntimeError("GraphedStep was captured for a batch with a different padding hint; capture again")
        verbs = torch.cat([t["verb_labels"] for t in targets])
        for dst, src in zip(self.static_in, (samples.tensors, samples.mask, text["input_ids"], text["attention_mask"], verbs)):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src)

    def __call__(self, samples, text, targets):
        """model phases only (criterion=None protocol): detached output leaves for an eager criterion"""
        self._load_inputs(samples, text, targets)
        self.fwd_graph.replay()
        self.leaves = [o.detach().requires_grad_(i in self.diff) for i, o in enumerate(self.static_out)]
        return self.wrapper.unflatten(self.leaves)

    def backward(self):
        """Replay the model's backward from the `.grad` the criterion left on the output leaves."""
        for buf, i in zip(self.static_gout, self.diff):
            g = self.leaves[i].grad
            if g is None:
                buf.zero_()
            else:
                buf.copy_(g)
        self.bwd_graph.replay()
        self._deliver()
        self.leaves = None

    def parameters(self):
        return self.step_module.parameters()

    @property
    def grad_scale(self):
        """factor the optimiser has to apply to `p.grad` (1 / world when the synchronizer leaves the SUM)"""
        s = self.synchronizer
        return s.grad_scale if (s is not None and s.scale_in_optimizer) else 1.0

    def _deliver(self):
        if self.synchronizer is not None and not self.overlap:
            self.synchronizer.all_reduce()
        # `p.grad` of every parameter points at this graph's static gradient buffers.  They still do from the previous step
        # unless somebody cleared them or another bucket's graph delivered in between (train_step does not zero the gradients
        # of a graphed step: the replay overwrites the buffers) -- first and last parameter tell; 750 attribute writes per
        # step were ~0.2 ms of Python between the backward graph and the optimiser.
        # Every VERIFY_EVERY-th delivery checks all of them: a partial zero_grad / a group frozen mid-run / a user hook that
        # replaced some `.grad` would otherwise leave those parameters without their gradient for the rest of the run.
        ps, gs = self.params, self.static_grads
        last = GraphedStep._delivered_by
        self._deliveries = getattr(self, "_deliveries", 0) + 1
        if ps and ps[0].grad is gs[0] and ps[-1].grad is gs[-1] and last is not None and last() is self:
            if self._deliveries % self.VERIFY_EVERY or all(p.grad is g for p, g in zip(ps, gs)):
                return
        for p, g in zip(ps, gs):
            p.grad = g
        import weakref
        GraphedStep._delivered_by = weakref.ref(self)

    def run(self, samples, text, targets):
        """Whole step up to the optimiser (criterion captured): forward graph, host assignment, backward
        graph; leaves the gradients in `p.grad` and returns (loss dict, weighted total) as static tensors."""
        if [len(t["obj_labels"]) for t in targets] != self.sizes:
            raise RuntimeError("GraphedStep was captured for different numbers of targets per image; capture again")
        self._load_inputs(samples, text, targets)
        for dst, src in zip(self.static_targets, targets):
            for k, v in dst.items():
                if v.data_ptr() != src[k].data_ptr():
                    v.copy_(src[k])
        self.fwd_graph.replay()
        self.pinned_index.copy_(self.criterion.assign(self.state, self._cost_on_host()))
        self.static_index.copy_(self.pinned_index, non_blocking=True)
        if dist.is_available() and dist.is_initialized():
            # (the other ranks' target counts change from step to step; alone, the captured value stays right)
            self.static_num.copy_(self.criterion._num_interactions(self.sizes, self.static_num.device).reshape(1))
        self.bwd_graph.replay()
        self._deliver()
        return self.loss_dict, self.total


class EagerSyncStep:
    """The train step without graphs, issuing exactly the collectives a GraphedStep replay issues (the interaction count,
    then the gradient average on the synchronizer's current schedule: bucketed in plan order, or flat) -- what a rank
    runs for a batch bucket it has not captured (yet) while other ranks replay theirs, and the step of a
    data-parallel run without graphs.  Same protocol as GraphedStep.run: gradients in `p.grad` (views of the flat
    buffer with a synchronizer), `grad_scale` for the optimiser, returns (loss dict, weighted total)."""

    def __init__(self, step_module, criterion, synchronizer=None, overlap=False, autocast_dtype=None):
        self.step_module, self.criterion, self.synchronizer = step_module, criterion, synchronizer
        # (the eager bucketed schedule runs on any backend; whoever pairs this step with captured ones passes the schedule
        #  those can run -- GraphedStepCache does)
        self.overlap = bool(overlap) and synchronizer is not None
        self.autocast_dtype = autocast_dtype
        self.params = synchronizer.params if synchronizer is not None else [p for p in step_module.parameters()
                                                                            if p.requires_grad]
        self.arrival = None

    @property
    def grad_scale(self):
        s = self.synchronizer
        return s.grad_scale if (s is not None and s.scale_in_optimizer) else 1.0

    def parameters(self):
        return self.step_module.parameters()

    def run(self, samples, text, targets):
        dt = self.autocast_dtype
        with torch.autocast(samples.tensors.device.type, dtype=dt, enabled=dt is not None):
            outputs = self.step_module(samples, text, targets)
        loss_dict = self.criterion(outputs, targets)                 # (all-reduces the interaction count)
        total = self.criterion.weighted_sum(loss_dict)
        s = self.synchronizer
        if s is None:
            grads = torch.autograd.grad(total, self.params, allow_unused=True)
        elif self.overlap and s.planned:
            with s.hooked():
                torch.autograd.grad(total, self.params, allow_unused=True)
            grads = s.views
        else:
            # flat schedule; an overlapped run without a bucket plan records the arrival order here (GraphedStepCache
            # turns it into the plan after this step, on all ranks at once)
            with s.recording() as rec:
                g = torch.autograd.grad(total, self.params, allow_unused=True)
            self.arrival = rec.order
            grads = s(g)
        for p, g in zip(self.params, grads):
            p.grad = g
        return loss_dict, total.detach()


class GraphedStepCache:
    """One GraphedStep per batch bucket, an eager step for everything else.  A capture is specific to the image tensor's
    shape, the padding hint and the number of targets per image (the matched-index tensor's shape); the reference pads
    every batch to its own maximum and has a variable number of triplets per image (util/misc.py:299-320,
    datasets/vg.py), and under data parallelism every rank pads its own shard -- the ranks meet different buckets at
    different steps.  The reference's DistributedDataParallel takes any shapes (main.py:515-517); so does this:

    * a bucket seen fewer than `capture_after` times runs as EagerSyncStep (any shape works, nothing is captured for
      one-off shapes), afterwards it is captured (a few eager warm-up steps + two captures) and replayed;
    * every decision is LOCAL -- hit, miss, capture, eviction of the least recently used capture beyond `max_buckets`.
      That is sound because all three ways of running a step issue the same sequence of collectives (interaction count,
      then the gradient buckets in plan order or the one flat all-reduce) and a capture issues none: a rank that
      captures is merely late for the step's first collective.  No per-step agreement exchange, no host sync;
    * the overlapped schedule needs one bucket plan shared by all ranks before anything is captured on it: step 0 (a
      miss on every rank by construction) runs eagerly on the flat schedule, records the gradient arrival order, and
      `after_step` broadcasts rank 0's order -- the only collective of the cache, at the same point on all ranks.

    `factory(batch) -> step` builds the captured step (default: GraphedStep); tests on the CPU pass a stand-in."""

    def __init__(self, step_module, model, synchronizer=None, criterion=None, max_buckets=8, capture_after=1,
                 overlap=False, autocast_dtype=None, factory=None):
        self.step_module, self.model = step_module, model
        self.synchronizer, self.criterion = synchronizer, criterion
        self.max_buckets, self.capture_after = max_buckets, max(1, int(capture_after))
        # ONE schedule for the captured steps and the eager step: GraphedStep can only run the overlapped one on RCCL
        # (a stand-in factory, CPU tests, runs whatever it is given)
        self.overlap = (effective_overlap(overlap, synchronizer) if factory is None
                        else bool(overlap) and synchronizer is not None)
        self.graphs = {}                    # key -> GraphedStep (insertion order = recency)
        self.seen = {}                      # key -> times met
        self.captures = self.hits = self.eager_steps = self.evictions = 0
        self.eager = EagerSyncStep(step_module, criterion, synchronizer, overlap=self.overlap, autocast_dtype=autocast_dtype)
        self.factory = factory or (lambda batch: GraphedStep(step_module, model, batch, synchronizer=synchronizer,
                                                             criterion=criterion, overlap=self.overlap))

    @staticmethod
    def bucket(batch):
        samples, text, targets = batch
        split = text.get("obj_pred_names_sums") if isinstance(text, dict) else None
        if split is not None:               # (a host-side list / CPU tensor: a device tensor here would cost a sync per step)
            split = tuple(int(v) for v in (split.flatten().tolist() if torch.is_tensor(split) else split))
        ids = text["input_ids"].shape if isinstance(text, dict) else tuple(t.shape for t in text if torch.is_tensor(t))
        return (tuple(samples.tensors.shape), str(samples.tensors.dtype), bool(getattr(samples, "no_padding", False)),
                tuple(ids), tuple(len(t["obj_labels"]) for t in targets),
                # the (objects, predicates) split of the text rows and the width of the verb labels are baked into a
                # capture as well: two batches that differ only there must not share a graph
                split or (), tuple(tuple(t["verb_labels"].shape) for t in targets))

    def _plan_pending(self):
        return self.overlap and not self.synchronizer.planned

    def plan(self, batch):
        """overlapped schedule without a bucket plan: one forward + backward of `batch` on the flat schedule (no optimiser
        step) records the gradient arrival order, then rank 0's order becomes the plan -- all ranks call this together.
        A training run does not need it (step 0 does the same as a side effect)."""
        if not self._plan_pending():
            return
        self.eager.run(*batch)
        self.after_step()
        for p in self.eager.params:
            p.grad = None

    def register(self, batches):
        """capture the buckets of `batches` now, whatever `capture_after` says (before a timed region).  With the
        overlapped schedule and no bucket plan yet this plans first -- then all ranks must call it together."""
        batches = list(batches)
        if batches:
            self.plan(batches[0])
        for b in batches:
            key = self.bucket(b)
            self.seen[key] = max(self.seen.get(key, 0), self.capture_after - 1)
            self.get(b)

    def get(self, batch):
        """the step object for this batch: a captured GraphedStep, or the eager step"""
        key = self.bucket(batch)
        g = self.graphs.pop(key, None)
        if g is not None:
            self.graphs[key] = g            # most recently used
            self.hits += 1
            return g
        self.seen[key] = self.seen.get(key, 0) + 1
        if self._plan_pending() or self.seen[key] < self.capture_after:
            self.eager_steps += 1
            return self.eager
        while len(self.graphs) >= self.max_buckets:
            self.graphs.pop(next(iter(self.graphs)))
            self.evictions += 1
        g = self.factory(batch)
        if getattr(g, "overlap", self.overlap) != self.overlap:
            raise RuntimeError("GraphedStepCache: the captured step and the eager step disagree on the gradient schedule")
        self.captures += 1
        self.graphs[key] = g
        return g

    def after_step(self):
        """end of a train step (train_step calls it): fixes the bucket plan of the overlapped schedule after step 0"""
        if self._plan_pending() and self.eager.arrival is not None:
            self.synchronizer.plan_collectively(self.eager.arrival)

    def parameters(self):
        return self.step_module.parameters()


class NonFiniteLoss(FloatingPointError):
    pass


class NonFiniteGuard:
    """The reference stops training on a non-finite loss (engine.py:123-128: `if not math.isfinite(loss_value)`:
    print the loss dict, exit) -- with a `.item()` sync every step.  Here the check rides along without a sync of its
    own: each step enqueues `isfinite(loss)` into a pinned byte (asynchronous copy + event); the NEXT step, whose
    matcher copy has synchronised the stream anyway, reads the byte and raises NonFiniteLoss.  Detection is one step
    late; the optimiser update of the bad step has then been applied, exactly as in the reference's ordering the
    update is skipped only because the process exits."""

    def __init__(self):
        self.flag = None
        self.event = None
        self.step = 0

    def submit(self, loss):
        self.check()
        if loss.is_cuda:
            if self.flag is None:
                self.flag = torch.ones(1, dtype=torch.uint8, pin_memory=True)
                self.event = torch.cuda.Event()
            self.flag.copy_(torch.isfinite(loss.detach()).reshape(1).to(torch.uint8), non_blocking=True)
            self.event.record()
            self.pending = True
        else:
            if not bool(torch.isfinite(loss.detach()).all()):
                raise NonFiniteLoss(f"loss is {float(loss)} at step {self.step}, stopping training")
        self.step += 1

    def check(self, wait=False):
        if getattr(self, "pending", False):
            if wait:
                self.event.synchronize()
            if self.event.query():
                self.pending = False
                if int(self.flag[0]) == 0:
                    raise NonFiniteLoss(f"loss was not finite at step {self.step - 1}, stopping training")


def graph_step_module(step_module, model, batch, synchronizer=None, criterion=None, overlap=False):
    """Capture `step_module` (both model phases, forward and backward) for the shapes of `batch`; returns a
    GraphedStep.  Raises if capture is not possible.  `synchronizer`: a GradientSynchronizer for data-parallel
    runs (the gradient all-reduce then follows the backward replay).  `criterion`: capture the criterion's
    device work into the two graphs as well (`GraphedStep.run`)."""
    return GraphedStep(step_module, model, batch, synchronizer=synchronizer, criterion=criterion, overlap=overlap)


def captured_collective_selftest(device, group=None):
    """Can this process group's all-reduce be captured into a HIP graph on a side stream and replayed?  A tiny graph
    (fork to a communication stream, SUM all-reduce, join) is captured, replayed twice and checked; every rank gets the
    same answer (the verdicts are combined with a MIN all-reduce), so all ranks pick the same schedule."""
    import torch.distributed as dist
    if dist.get_backend(group) != "nccl":
        # only RCCL's collectives can be stream-captured; a host-staged backend (gloo) raises INSIDE the capture and
        # leaves the stream in capture mode for good -- do not even try
        return False
    ok = 1
    try:
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        x = torch.zeros(1024, device=device)
        comm = torch.cuda.Stream(device=device)
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            y = x * 2.0
            comm.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(comm):
                work = dist.all_reduce(y, op=dist.ReduceOp.SUM, group=group, async_op=True)
            torch.cuda.current_stream().wait_stream(comm)
            work.wait()
            z = y + 1.0
        for k in (1, 2):
            x.fill_(float(rank + k))
            g.replay()
            torch.cuda.synchronize()
            expect = 2.0 * sum(r + k for r in range(world)) + 1.0
            if not bool(torch.all(z == expect)):
                ok = 0
    except Exception:                                       # noqa: BLE001 -- any failure means "do not use it"
        ok = 0
    flag = torch.tensor([ok], device=device, dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return bool(int(flag.item()))


SCHEDULE_RTOL = 1e-3          # choose_dp_schedule: loss and gradient norm of the overlapped step against the flat step's


def choose_dp_schedule(build_step, batch, device, group=None, rtol=SCHEDULE_RTOL, selftest=None, seed=20251004, log=None):
    """The `auto` gradient schedule of a data-parallel run (reference: DistributedDataParallel's bucketed all-reduce overlapped
    with the backward pass, main.py:515-517).  The OVERLAPPED schedule (buckets in arrival order, captured inside the backward
    graph on a communication stream) is taken when
      (1) this process group's collectives can be captured and replayed at all (`captured_collective_selftest`, on all ranks), and
      (2) one forward + backward of `batch` on it reproduces the FLAT schedule's step (one all-reduce after the backward graph):
          loss and norm of the synchronised gradient equal to `rtol` (or to four times the flat step's own run-to-run distance,
          measured with a second flat step, where that is wider), under the same random state, on every rank;
    otherwise the flat schedule runs.  Every rank reaches the same decision (MIN all-reduces); no optimiser step is taken and the
    parameters' `.grad` are left empty.

    `build_step(overlap) -> step`: a GraphedStep / EagerSyncStep-like object on its OWN GradientSynchronizer (`run(samples, text,
    targets) -> (loss dict, total)`, `parameters()`, `grad_scale`).  All ranks must call this function together.
    -> (step, "overlapped" | "flat", reason)."""
    import torch.distributed as dist
    dev = torch.device(device)

    def agree(flag):
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
        return bool(int(t.item()))

    def probe(step):
        params = [p for p in step.parameters() if p.requires_grad]
        for p in params:
            p.grad = None
        torch.manual_seed(seed)                       # (CPU and every device generator: the dropouts draw the same masks twice)
        _, total = step.run(*batch)
        grads = [p.grad for p in params if p.grad is not None]
        sq = torch.stack([g.detach().float().square().sum() for g in grads]).sum() if grads else torch.zeros((), device=dev)
        out = float(total.detach().float()), float(sq.sqrt()) * float(getattr(step, "grad_scale", 1.0))
        for p in params:
            p.grad = None
        return out

    cpu_state = torch.get_rng_state()
    cuda_state = torch.cuda.get_rng_state(dev) if dev.type == "cuda" else None
    try:
        can = agree((selftest or captured_collective_selftest)(device, group))
        flat = build_step(False)
        if not can:
            return flat, "flat", "the group's collectives cannot be captured into a graph here (self-test)"
        loss_f, norm_f = probe(flat)
        # the step's own run-to-run distance under the same random state (one MIOpen convolution of the step is not repeatable bit for
        # bit, profiles/r03_nondeterminism.txt): the bars below are `rtol`, or four times this noise where that is wider -- a schedule
        # that loses or doubles a bucket moves the gradient norm by that bucket's share, orders of magnitude more than either
        loss_2, norm_2 = probe(flat)
        tol_loss = max(rtol, 4.0 * abs(loss_2 - loss_f) / max(abs(loss_f), 1e-12))
        tol_norm = max(rtol, 4.0 * abs(norm_2 - norm_f) / max(norm_f, 1e-12))
        over, why = None, None
        try:
            over = build_step(True)
        except Exception as e:                                  # noqa: BLE001 -- the flat schedule is always available
            why = f"the overlapped step could not be built: {type(e).__name__}: {e}"
        if not agree(over is not None):
            return flat, "flat", why or "the overlapped step could not be built on another rank"
        loss_o, norm_o = probe(over)
        same = (loss_o == loss_o and norm_o == norm_o and abs(loss_o - loss_f) <= tol_loss * max(abs(loss_f), 1e-12)
                and abs(norm_o - norm_f) <= tol_norm * max(norm_f, 1e-12))
        if log is not None:
            log(f"[dp schedule] flat: loss {loss_f:.6g}, gradient norm {norm_f:.6g} (a second flat step: {loss_2:.6g}, {norm_2:.6g}); "
                f"overlapped: loss {loss_o:.6g}, gradient norm {norm_o:.6g} -> {'equal' if same else 'DIFFERENT'} at "
                f"{tol_loss:.2g} / {tol_norm:.2g} on this rank")
        if agree(same):
            return over, "overlapped", f"captured collectives replay and its first step equals the flat schedule's to {rtol:g}"
        del over
        return flat, "flat", (f"the overlapped step differs from the flat one (loss {loss_o:.6g} vs {loss_f:.6g}, gradient norm "
                              f"{norm_o:.6g} vs {norm_f:.6g})" if not same else "the overlapped step differs on another rank")
    finally:
        torch.set_rng_state(cpu_state)
        if cuda_state is not None:
            torch.cuda.set_rng_state(cuda_state, dev)


def broadcast_parameters(module, src=0):
    """Rank `src`'s parameters and buffers to every rank (what DistributedDataParallel does when it wraps)."""
    import torch.distributed as dist
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src)


class GradientSynchronizer:
    """Data-parallel gradient averaging without DistributedDataParallel, for the graphed step.

    The gradients live in ONE flat buffer (bf16 in the master-weight mode, 425 MB for 212.7 M parameters); the
    optimiser reads them as views of it, in the parameters' own memory layouts.  Two schedules:

    * bucketed and overlapped (`plan_buckets` + `hooked`, the default of GraphedStep): the flat buffer is cut into
      buckets in the order the gradients become available during the backward pass (recorded in the eager warm-up);
      a tensor hook on every parameter copies its gradient into the buffer as soon as autograd has it, and the hook
      that completes a bucket starts that bucket's all-reduce on a communication stream -- while autograd goes on
      with the earlier layers.  Under HIP-graph capture hooks, copies and collectives are all captured: the backward
      graph then holds the RCCL all-reduces on a forked branch, joined before the graph ends.  This is what
      DistributedDataParallel's reducer does for an eager backward (reference main.py:515-517), in a form that can
      be replayed.  xGMI is point-to-point, a ring all-reduce is per-link bound: few, large buckets (96 MB default).
    * flat (`pack` + `all_reduce`): one copy, ONE all-reduce after the backward pass, nothing overlapped.

    `reduce_dtype=torch.float32`: the reduction runs on a float32 copy of each bucket (the reference's DDP reduces
    float32 gradients) instead of in the gradients' own bfloat16."""

    def __init__(self, params, group=None, reduce_dtype=None, bucket_bytes=96 << 20):
        self.params = list(params)
        self.group = group
        self.reduce_dtype = reduce_dtype
        self.bucket_bytes = bucket_bytes
        self._avg = None
        total = sum((p.numel() + 7) // 8 * 8 for p in self.params)     # every view starts on a 16-byte boundary
        p0 = self.params[0]
        self.flat = torch.zeros(total, dtype=p0.dtype, device=p0.device)
        self.views = [None] * len(self.params)
        self.comm = torch.cuda.Stream(device=p0.device) if p0.is_cuda else None
        self.pending = []
        self.planned = False                 # plan_buckets() has fixed the layout (captured graphs depend on it)
        self.scale_in_optimizer = False      # leave the all-reduced SUM; the optimiser multiplies by `grad_scale`
        self.grad_scale = 1.0
        self.set_buckets([list(range(len(self.params)))])

    # ---- layout ---------------------------------------------------------------------------------------------------
    def set_buckets(self, buckets):
        """Partition of the parameter indices into buckets, each one contiguous stretch of the flat buffer."""
        assert sorted(i for b in buckets for i in b) == list(range(len(self.params)))
        self.buckets = [list(b) for b in buckets]
        self.bucket_of = [0] * len(self.params)
        self.ranges, off = [], 0
        for k, b in enumerate(self.buckets):
            start = off
            for i in b:
                p = self.params[i]
                n = p.numel()
                v = self.flat[off:off + n]
                # keep the parameter's memory layout (channels-last convolution weights) so that the fused
                # optimiser sees gradient and parameter in the same element order
                self.views[i] = v.as_strided(p.shape, p.stride()) if p.is_contiguous() or _dense(p) else v.view(p.shape)
                self.bucket_of[i] = k
                off += (n + 7) // 8 * 8
            self.ranges.append((start, off))
        self._stage = [None] * len(self.buckets)

    def plan_buckets(self, order):
        """`order`: parameter indices in the order their gradients arrive in the backward pass (every rank must pass
        the same order; parameters missing from it go last).  Greedy buckets of ~bucket_bytes in that order."""
        seen = set(order)
        order = list(order) + [i for i in range(len(self.params)) if i not in seen]
        buckets, cur, size = [], [], 0
        for i in order:
            cur.append(i)
            size += self.params[i].numel() * self.flat.element_size()
            if size >= self.bucket_bytes:
                buckets.append(cur)
                cur, size = [], 0
        if cur:
            buckets.append(cur)
        if self.planned:
            if [list(b) for b in buckets] != self.buckets:
                raise RuntimeError("GradientSynchronizer: the bucket layout is already planned (graphs captured on it hold "
                                   "its offsets); a different plan needs a new synchronizer")
            return self.buckets
        self.set_buckets(buckets)
        self.planned = True
        return buckets

    def plan_collectively(self, order):
        """rank 0's arrival order becomes everybody's bucket plan (a collective: all ranks call it at the same point)"""
        import torch.distributed as dist
        order = [list(order)]
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            dist.broadcast_object_list(order, src=0, group=self.group)
        return self.plan_buckets(order[0])

    # ---- flat schedule --------------------------------------------------------------------------------------------
    def pack(self, grads):
        """copy one gradient (or None = zero) per parameter into the flat buffer; returns the views"""
        have = [(v, g) for v, g in zip(self.views, grads) if g is not None]
        missing = [v for v, g in zip(self.views, grads) if g is None]
        if missing:
            torch._foreach_zero_(missing)
        # one mismatching (dst, src) layout pair sends the WHOLE _foreach_copy_ down its per-tensor slow path
        # (747 copy kernels = 4 ms inside the backward graph, measured): keep those pairs out of it.  They are
        # the ~20 convolution weights held channels-last whose gradients arrive contiguous.
        same = [(v, g) for v, g in have if v.stride() == g.stride() and v.dtype == g.dtype]
        other = [(v, g) for v, g in have if not (v.stride() == g.stride() and v.dtype == g.dtype)]
        if same:
            torch._foreach_copy_([v for v, _ in same], [g for _, g in same])
        for v, g in other:
            v.copy_(g)
        return self.views

    def _probe_avg(self):
        """ReduceOp.AVG is what RCCL/NCCL offer for this; gloo (CPU tests) does not -- find out once, with a
        collective every rank takes part in, and fall back to SUM followed by a scale."""
        import torch.distributed as dist
        try:
            dist.all_reduce(torch.zeros(1, dtype=self.flat.dtype, device=self.flat.device), op=dist.ReduceOp.AVG,
                            group=self.group)
            return True
        except (RuntimeError, ValueError):
            return False

    def all_reduce(self):
        import torch.distributed as dist
        if self.scale_in_optimizer:
            dist.all_reduce(self.flat, group=self.group)
            self.grad_scale = 1.0 / dist.get_world_size(self.group)
            return
        if self._avg is None:
            self._avg = self._probe_avg()
        if self._avg:
            dist.all_reduce(self.flat, op=dist.ReduceOp.AVG, group=self.group)
        else:
            dist.all_reduce(self.flat, group=self.group)
            self.flat.div_(dist.get_world_size(self.group))

    def __call__(self, grads):
        """grads: one tensor (or None) per parameter -> averaged gradients as views of the flat buffer"""
        self.pack(grads)
        self.all_reduce()
        return self.views

    # ---- bucketed, overlapped schedule ----------------------------------------------------------------------------
    def recording(self):
        """context: records the order in which the parameters' gradients arrive (`.order` afterwards)"""
        return _GradHooks(self, record_only=True)

    def hooked(self):
        """context around the backward pass (`torch.autograd.grad(loss, params)` or `loss.backward()`): every
        parameter's gradient is copied into its view the moment autograd has it, a complete bucket is handed to the
        collective at once.  On exit the parameters that received no gradient are zero-filled, the remaining buckets
        launched and the current stream made to wait for all of them: the views hold the averaged gradients."""
        return _GradHooks(self, record_only=False)

    def launch_bucket(self, k, streams=()):
        """Average bucket k on the communication stream, ordered after the work of `streams` (default: the current
        stream); returns at once, `finish()` joins.  The collective is a SUM followed by a scale on the same stream:
        ReduceOp.AVG captured into a HIP graph came back as garbage on replay (RCCL 2.26, measured on a 1-rank group;
        SUM replays correctly), and gloo has no AVG at all."""
        import torch.distributed as dist
        a, b = self.ranges[k]
        if a == b:
            return
        part = self.flat[a:b]
        world = dist.get_world_size(self.group)
        if self.comm is not None:
            for st in (streams or [torch.cuda.current_stream(self.flat.device)]):
                if st != self.comm:
                    self.comm.wait_stream(st)
        with (torch.cuda.stream(self.comm) if self.comm is not None else _nullcontext()):
            buf = part
            if self.reduce_dtype is not None and self.reduce_dtype != part.dtype:
                if self._stage[k] is None:
                    self._stage[k] = torch.empty(b - a, dtype=self.reduce_dtype, device=part.device)
                buf = self._stage[k]
                buf.copy_(part)
            work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        # (waiting for the work on the communication stream right here, inside the autograd hook, segfaulted in
        #  hipGraphInstantiate at the end of the capture: the join and the 1/world scale happen in finish())
        self.pending.append((work, part, buf, world))

    @staticmethod
    def _scale_back(part, buf, world):
        if buf is not part:
            part.copy_(buf if world == 1 else buf / world)
        elif world > 1:
            part.mul_(1.0 / world)

    def finish(self):
        """the current stream waits for every launched bucket; the averaged gradients are then in the views"""
        if self.comm is not None:
            torch.cuda.current_stream(self.flat.device).wait_stream(self.comm)
        for work, part, buf, world in self.pending:
            work.wait()
        if self.pending:
            world = self.pending[0][3]
            if self.scale_in_optimizer and all(buf is part for _, part, buf, _ in self.pending):
                self.grad_scale = 1.0 / world    # the optimiser's kernels multiply: no pass over the buffer here
            elif all(buf is part for _, part, buf, _ in self.pending):
                if world > 1:                    # one pass over the flat buffer: SUM -> mean
                    lo, hi = min(self.ranges)[0], max(self.ranges)[1]
                    self.flat[lo:hi].mul_(1.0 / world)
            else:
                for _, part, buf, _ in self.pending:
                    self._scale_back(part, buf, world)
        self.pending = []


class _GradHooks:
    """tensor hooks of GradientSynchronizer.recording() / .hooked()"""

    def __init__(self, sync, record_only):
        self.sync, self.record_only = sync, record_only
        self.order = []

    def __enter__(self):
        import threading
        s = self.sync
        self.lock = threading.Lock()
        self.left = [len(b) for b in s.buckets]
        self.next = 0                                    # buckets leave in INDEX order, whatever order they complete in:
        self.streams = [[] for _ in s.buckets]           # the sequence of collectives is the same on every rank
        self.got = [False] * len(s.params)
        self.held = [dict() for _ in s.buckets]          # bucket -> {stream: (views, gradients)} waiting for their copy
        self.handles = [p.register_hook(lambda g, i=i: self._arrived(i, g)) for i, p in enumerate(s.params)]
        return self

    def _arrived(self, i, g):
        s = self.sync
        with self.lock:                       # (autograd may run hooks from its device thread)
            if self.got[i]:
                return None
            self.got[i] = True
            self.order.append(i)
            if self.record_only:
                return None
            # A gradient is copied on the stream autograd hands it over on (valid there, its memory released in that
            # stream's order: no cross-stream lifetime to manage) -- but not one copy kernel per parameter (747 launch-
            # bound kernels inside the backward graph were +2 ms per step at world size 1): the gradients of a bucket
            # are kept until the bucket is complete and leave as ONE multi-tensor copy per contributing stream.
            k = s.bucket_of[i]
            cur = torch.cuda.current_stream(s.flat.device) if s.comm is not None else None
            if g.data_ptr() != s.views[i].data_ptr():                 # (already in place: nothing to copy)
                v = s.views[i]
                if v.stride() == g.stride() and v.dtype == g.dtype:
                    self.held[k].setdefault(cur, ([], []))
                    self.held[k][cur][0].append(v)
                    self.held[k][cur][1].append(g)
                else:
                    v.copy_(g)                                        # layout / dtype differs: its own copy, at once
            if cur is not None and all(cur != st for st in self.streams[k]):
                self.streams[k].append(cur)
            self.left[k] -= 1
            self._launch_ready()
        return None

    def _launch_ready(self):
        s = self.sync
        while self.next < len(self.left) and self.left[self.next] == 0:
            k = self.next
            self._flush(k)
            s.launch_bucket(k, self.streams[k])
            self.next += 1

    def _flush(self, k):
        """the held gradients of bucket k -> their views, one multi-tensor copy per stream they arrived on"""
        for st, (views, grads) in self.held[k].items():
            if not views:
                continue
            with (torch.cuda.stream(st) if st is not None else _nullcontext()):
                torch._foreach_copy_(views, grads)
        self.held[k] = {}

    def __exit__(self, *exc):
        s = self.sync
        for h in self.handles:
            h.remove()
        if self.record_only or exc[0] is not None:
            return False
        missing = [i for i, got in enumerate(self.got) if not got]
        if missing:
            torch._foreach_zero_([s.views[i] for i in missing])
        for k in sorted({s.bucket_of[i] for i in missing}):
            self.left[k] -= sum(1 for i in missing if s.bucket_of[i] == k)
            if s.comm is not None:
                cur = torch.cuda.current_stream(s.flat.device)
                if all(cur != st for st in self.streams[k]):
                    self.streams[k].append(cur)          # (the zero-fill above ran on this stream)
        self._launch_ready()
        assert all(n == 0 for n in self.left) and self.next == len(self.left), (self.left, self.next)
        s.finish()
        return False


class _nullcontext:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


def _dense(t):
    """non-overlapping and dense (any permutation of a contiguous layout)"""
    n, expect = t.numel(), 1
    for size, stride in sorted(zip(t.shape, t.stride()), key=lambda x: x[1]):
        if size == 1:
            continue
        if stride != expect:
            return False
        expect *= size
    return expect == n or n == 0


def freeze_statically_unused(model):
    """sub/obj_bbox_embed[n_pred:] only produce detached reference points in the verb decoder."""
    n_pred = model.transformer.ho_decoder.num_layers
    frozen = 0
    for heads in (model.sub_bbox_embed, model.obj_bbox_embed):
        for head in list(heads)[n_pred:]:
            for p in head.parameters():
                p.requires_grad_(False)
                frozen += p.numel()
    return frozen


def freeze_parameters_without_gradient(step_module, criterion, batch, autocast_dtype=None):
    """Dry run (forward + backward, no optimiser step) and freeze every trainable parameter that
    received no gradient: the static unused-parameter mask that lets DistributedDataParallel run
    with find_unused_parameters=False (a bucket holding a parameter that never gets a gradient would
    never be reduced).  Returns the frozen names.  Call BEFORE wrapping in DDP."""
    samples, text, targets = batch
    device_type = samples.tensors.device.type
    with torch.autocast(device_type, dtype=autocast_dtype, enabled=autocast_dtype is not None):
        outputs = step_module(samples, text, targets)
    criterion.weighted_sum(criterion(outputs, targets)).backward()
    frozen = []
    for n, p in step_module.named_parameters():
        if p.requires_grad and p.grad is None:
            p.requires_grad_(False)
            frozen.append(n)
        p.grad = None
    return frozen


def build_training(args=None, device="cuda:0", with_text_encoder=True, backbone_name="resnet50"):
    """model + criterion of the train step; `backbone_name`: "resnet50" (BASELINE configs 2-3) or a swin name
    ("swin_large": configs 4-5).  The GPU-only host routes are OFF in what this returns: `validate_host_routes` (below)
    switches on the ones that pass their self-check on the caller's first batch."""
    args = default_args() if args is None else args
    if torch.cuda.is_available() and str(device).startswith("cuda"):
        from .linear import use_tuned_library_gemms
        use_tuned_library_gemms()
    if "swin" in backbone_name:
        from .swin import build_swin_backbone
        backbone = build_swin_backbone(backbone_name, args.hidden_dim, num_feature_levels=3)
    else:
        backbone = build_r50_backbone(args.hidden_dim, train_backbone=True)
    text_encoder = TextEncoderStub() if with_text_encoder else None
    model = build_parseda(backbone, args, text_encoder=text_encoder).to(device)
    freeze_statically_unused(model)
    swap_linears(model)
    matcher = crit_mod.HungarianMatcherHOI(cost_obj_class=1, cost_verb_class=1, cost_bbox=2.5, cost_giou=1,
                                           subject_class=args.subject_class)
    criterion = crit_mod.SetCriterionHOI(matcher, crit_mod.build_weight_dict(args.dec_layers),
                                         eos_coef=0.1, subject_class=args.subject_class, giou_verb_label=True,
                                         pseudo_verb=args.pseudo_verb).to(device)
    return model, criterion


def validate_host_routes(step_module, criterion, batch, autocast_dtype=None, group=None, log=None):
    """Start-up self-check of the GPU-only routes (rlipv2_amd/routes.py): they are OFF after `build_training`; call this once
    with the step module, the criterion and one real batch before the first optimiser step (and before any graph capture) to
    switch on the ones that reproduce the plain step's loss and gradients.  Returns {route: "on" | "off (...)"}; all ranks
    of a data-parallel run must call it together (one MIN all-reduce).  bench.py does this and reports `config.host_routes`."""
    from . import routes
    return routes.validate(step_module, criterion, batch, autocast_dtype=autocast_dtype, group=group, log=log)


def build_optimizer(model, lr=1.41e-4, lr_backbone=1.41e-5, text_encoder_lr=1.41e-5, weight_decay=1e-4):
    """AdamW, three groups by parameter name (reference main.py:523-539)."""
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    groups = [
        {"params": [p for n, p in named if "backbone" not in n and "text_encoder" not in n]},
        {"params": [p for n, p in named if "backbone" in n], "lr": lr_backbone},
        {"params": [p for n, p in named if "text_encoder" in n], "lr": text_encoder_lr},
    ]
    try:
        return torch.optim.AdamW(groups, lr=lr, weight_decay=weight_decay, fused=True)
    except (RuntimeError, TypeError):
        return torch.optim.AdamW(groups, lr=lr, weight_decay=weight_decay)


class MasterWeightAdamW:
    """AdamW over float32 master copies of a model whose parameters live in bfloat16.

    bf16 precision policy of the train step: parameters, activations and
    gradients are bfloat16 (no per-op autocast casts, half the elementwise traffic, bf16 gradient
    all-reduce); the optimiser state and the weights it updates are float32; after each step the
    master weights are rounded back into the model.  Clipping (max-norm 0.1, engine.py:170-171) is
    applied to the float32 copies of the gradients.  Parameter groups by name as in main.py:523-539."""

    def __init__(self, model, lr=1.41e-4, lr_backbone=1.41e-5, text_encoder_lr=1.41e-5, weight_decay=1e-4):
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        self.names = [n for n, _ in named]
        self.params = [p for _, p in named]
        self.master = [p.detach().float().clone().requires_grad_(True) for p in self.params]
        self.gbuf = [torch.zeros_like(m) for m in self.master]
        for m, g in zip(self.master, self.gbuf):
            m.grad = g
        by = lambda pred: [m for (n, _), m in zip(named, self.master) if pred(n)]
        groups = [{"params": by(lambda n: "backbone" not in n and "text_encoder" not in n)},
                  {"params": by(lambda n: "backbone" in n), "lr": lr_backbone},
                  {"params": by(lambda n: "text_encoder" in n), "lr": text_encoder_lr}]
        groups = [g for g in groups if g["params"]]
        try:
            self.opt = torch.optim.AdamW(groups, lr=lr, weight_decay=weight_decay, fused=True)
        except (RuntimeError, TypeError):
            self.opt = torch.optim.AdamW(groups, lr=lr, weight_decay=weight_decay)
        self.param_groups = self.opt.param_groups

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            p.grad = None

    @torch.no_grad()
    def step(self, max_norm=0.1):
        have = [i for i, p in enumerate(self.params) if p.grad is not None]
        if len(have) != len(self.params):                       # a parameter without gradient this step
            for i in set(range(len(self.params))) - set(have):
                self.gbuf[i].zero_()
        torch._foreach_copy_([self.gbuf[i] for i in have], [self.params[i].grad for i in have])
        if max_norm > 0:
            torch.nn.utils.clip_grad_norm_(self.master, max_norm, foreach=True)
        self.opt.step()
        torch._foreach_copy_(self.params, self.master)

    def state_dict(self):
        return {"names": list(self.names), "master": [m.detach() for m in self.master], "opt": self.opt.state_dict()}

    @torch.no_grad()
    def load_state_dict(self, state):
        if list(state["names"]) != self.names:
            raise ValueError("optimizer state belongs to a different parameter set")
        for m, s in zip(self.master, state["master"]):
            m.copy_(s)
        self.opt.load_state_dict(state["opt"])
        torch._foreach_copy_(self.params, self.master)


def to_bf16(model):
    """Parameters and buffers to bfloat16; the backbone additionally to channels-last."""
    model.to(torch.bfloat16)
    if hasattr(model, "backbone"):
        model.backbone.to(memory_format=torch.channels_last)
    return model


def synthetic_batch(batch, height=800, width=1333, n_obj=43, n_verb=21, triplets=8, token_len=5, device="cuda:0",
                    seed=0, sizes=None):
    """images ~ N(0,1) (post-normalisation), no padding; 43 object labels (last = "no
    objects") + 21 relation labels = 64 texts as seeded token ids of length 5; 8 triplets per image."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    images = torch.randn(batch, 3, height, width, generator=g).to(device)
    mask = torch.zeros(batch, height, width, dtype=torch.bool, device=device)
    n_text = n_obj + n_verb
    ids = torch.randint(3, 50000, (n_text, token_len), generator=g)
    ids[:, 0], ids[:, -1] = 0, 2
    text = {"input_ids": ids.to(device), "attention_mask": torch.ones_like(ids).to(device),
            "obj_pred_names_sums": torch.tensor([[n_obj, n_verb]])}
    targets = []
    for _ in range(batch):
        c = torch.rand(triplets, 2, generator=g) * 0.6 + 0.2
        wh = torch.rand(triplets, 2, generator=g) * 0.35 + 0.05
        c2 = torch.rand(triplets, 2, generator=g) * 0.6 + 0.2
        wh2 = torch.rand(triplets, 2, generator=g) * 0.35 + 0.05
        verbs = torch.zeros(triplets, n_verb)
        verbs[torch.arange(triplets), torch.randint(0, n_verb, (triplets,), generator=g)] = 1
        verbs[torch.arange(0, triplets, 2), torch.randint(0, n_verb, ((triplets + 1) // 2,), generator=g)] = 1
        targets.append({"sub_labels": torch.randint(0, n_obj - 1, (triplets,), generator=g).to(device),
                        "obj_labels": torch.randint(0, n_obj - 1, (triplets,), generator=g).to(device),
                        "verb_labels": verbs.to(device),
                        "sub_boxes": torch.cat([c, wh], 1).to(device), "obj_boxes": torch.cat([c2, wh2], 1).to(device)})
    if sizes is not None:
        # padded variant (sizes (800,1333) and (736,1100) in one batch): image k has size
        # sizes[k % len(sizes)], zero-padded to (height, width), mask True on the padding -- the reference's
        # nested_tensor_from_tensor_list (util/misc.py:299-320)
        for k in range(batch):
            h, w = sizes[k % len(sizes)]
            images[k, :, h:, :] = 0
            images[k, :, :, w:] = 0
            mask[k, h:, :] = True
            mask[k, :, w:] = True
        return NestedTensor(images, mask, no_padding=all(tuple(s) == (height, width) for s in sizes)), text, targets
    return NestedTensor(images, mask, no_padding=True), text, targets


def _forward_backward(step_module, criterion, optimizer, batch, autocast_dtype):
    """forward + criterion + backward of one batch on whatever kind of step object this is; leaves this batch's gradients
    in `p.grad` and returns (weighted loss, factor the optimiser has to apply to the gradients)"""
    samples, text, targets = batch
    cache = step_module if isinstance(step_module, GraphedStepCache) else None
    if cache is not None:
        step_module = cache.get(batch)                      # a captured bucket, or the eager step for this batch
    whole = isinstance(step_module, GraphedStep) and step_module.criterion is not None
    if not whole:
        optimizer.zero_grad(set_to_none=True)               # (a graphed step overwrites its static gradient buffers)
    if isinstance(step_module, EagerSyncStep) or whole:
        _, loss = step_module.run(samples, text, targets)
    else:
        with torch.autocast(samples.tensors.device.type, dtype=autocast_dtype, enabled=autocast_dtype is not None):
            outputs = step_module(samples, text, targets)
        loss_dict = criterion(outputs, targets)
        loss = criterion.weighted_sum(loss_dict)
        loss.backward()
        if isinstance(step_module, GraphedStep):
            step_module.backward()
    if cache is not None:
        cache.after_step()
    return loss, float(getattr(step_module, "grad_scale", 1.0))


def _optimizer_step(optimizer, max_norm, grad_scale, params=None):
    """clip + update; `grad_scale` (1 / world when the synchronizer leaves the all-reduced SUM in the gradients) is applied
    inside FusedMasterAdamW's kernels -- no other optimiser knows about it, so anything else must see averaged gradients"""
    if isinstance(optimizer, FusedMasterAdamW):
        optimizer.step(max_norm, grad_scale=grad_scale)
        return
    if grad_scale != 1.0:
        raise RuntimeError("GradientSynchronizer.scale_in_optimizer leaves the SUM over ranks in the gradients; only "
                           "FusedMasterAdamW applies the 1 / world factor -- switch scale_in_optimizer off for this optimiser")
    if isinstance(optimizer, MasterWeightAdamW):
        optimizer.step(max_norm)
    else:
        if max_norm and max_norm > 0:
            params = params if params is not None else [p for g in optimizer.param_groups for p in g["params"]]
            torch.nn.utils.clip_grad_norm_(params, max_norm, foreach=True)
        optimizer.step()


def train_step(step_module, criterion, optimizer, batch, max_norm=0.1, autocast_dtype=torch.bfloat16, guard=None):
    """One optimisation step; returns the (device) loss.  No host synchronisation except the
    matcher's single device->host copy of the cost matrices.  `autocast_dtype=None` runs the model
    in whatever dtype its parameters have (float32, or bfloat16 with MasterWeightAdamW).  `guard`: a
    NonFiniteGuard (raises NonFiniteLoss one step after a non-finite loss, engine.py:123-128).
    `step_module`: a plain module (or DistributedDataParallel), a GraphedStep, an EagerSyncStep or a GraphedStepCache."""
    loss, grad_scale = _forward_backward(step_module, criterion, optimizer, batch, autocast_dtype)
    _optimizer_step(optimizer, max_norm, grad_scale)
    if guard is not None:
        guard.submit(loss)
    return loss.detach()


# ---- BASELINE config 5: mixed-dataset pre-training protocol ----------------------------------------------------------------
class BatchIterativeSampler:
    """Batch sampler of the mixed-dataset pre-training (reference datasets/mixed_dataset.py:48-214,
    `BatchIterativeDistributedSampler`): the concatenation of K datasets is walked in rounds of
    `len(paradigm)` batches, batch j of a round drawn from dataset `paradigm[j]` -- e.g. "0,1,2,2": one batch of
    Visual Genome, one of COCO, two of Objects365 -- and every batch is cut across the ranks by stride.  Dataset 0 is
    the anchor: an epoch is one pass over it (padded to a multiple of the world size, or cut with `drop_last`); the
    other datasets are reshuffled with the same generator and truncated to as many samples as their share of the
    rounds needs.  Yields lists of indices into the CONCATENATED dataset.

    Restated from the behaviour (golden: tests/golden/protocol.json, produced by executing the reference class)."""

    def __init__(self, dataset_sizes, batch_size, paradigm, num_replicas=1, rank=0, shuffle=True, seed=0, drop_last=False):
        if not 0 <= rank < num_replicas:
            raise ValueError(f"Invalid rank {rank}, rank should be in the interval [0, {num_replicas - 1}]")
        self.sizes = [int(n) for n in dataset_sizes]
        self.batch_size = int(batch_size)
        self.paradigm = [int(d) for d in paradigm.split(",")] if isinstance(paradigm, str) else [int(d) for d in paradigm]
        self.num_replicas, self.rank = int(num_replicas), int(rank)
        self.shuffle, self.seed, self.drop_last = bool(shuffle), int(seed), bool(drop_last)
        self.epoch = 0
        anchor = self.sizes[0]
        if self.drop_last and anchor % self.num_replicas != 0:
            self.num_samples = math.ceil((anchor - self.num_replicas) / self.num_replicas)
        else:
            self.num_samples = math.ceil(anchor / self.num_replicas)
        self.total_size = self.num_samples * self.num_replicas

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def __len__(self):
        rounds = self.num_samples // self.batch_size if self.drop_last else -(-self.num_samples // self.batch_size)
        return rounds * len(self.paradigm)

    def __iter__(self):
        anchor = self.sizes[0]
        g = torch.Generator()
        g.manual_seed(self.seed + self.epoch)
        first = torch.randperm(anchor, generator=g).tolist() if self.shuffle else list(range(anchor))
        if not self.drop_last:
            pad = self.total_size - len(first)
            first += first[:pad] if pad <= len(first) else (first * math.ceil(pad / len(first)))[:pad]
        else:
            first = first[:self.total_size]
        per_dataset, offset = [first], anchor
        for d, n in enumerate(self.sizes[1:], start=1):
            order = torch.randperm(n, generator=g).tolist() if self.shuffle else list(range(n))
            share = sum(1 for p in self.paradigm if p == d)
            per_dataset.append([i + offset for i in order][:anchor * share])
            offset += n
        cursor = [0] * len(per_dataset)
        out = []
        for _ in range(math.ceil(self.num_samples / self.batch_size)):
            take = min(self.num_replicas * self.batch_size, len(per_dataset[0]) - cursor[0])
            for d in self.paradigm:
                chunk = per_dataset[d][cursor[d]:cursor[d] + take]
                out.append(chunk[self.rank::self.num_replicas])
                cursor[d] += take
        return iter(out)


class AccumulatedUpdate:
    """`--gradient_strategy gradient_accumulation` of the reference's train loop (engine.py:136-153): the weighted losses
    of the `len(paradigm)` consecutive batches of a round are SUMMED (not averaged) and one backward pass, one clipping
    and one optimiser step follow on the last batch of the round; `vanilla` updates on every batch (:161-166).

    Two equivalent forms of the same update:
      * `add(loss)` -- the reference's own form: keeps the round's autograd graphs alive and runs one backward over the
        summed loss (what the CPU golden pins, bit for bit in float32);
      * `add_gradients()` -- for steps whose backward has already run (the graphed step: its backward graph leaves the
        batch's gradients in static buffers): the parameters' `.grad` are added into float32 accumulators, and the
        last batch of the round hands the sums to the optimiser.  d(sum of losses) = sum of d(losses): same update,
        one batch of activations alive at a time."""

    def __init__(self, params, optimizer, paradigm, strategy="gradient_accumulation", max_norm=0.1, clip=None):
        self.params = [p for p in params if p.requires_grad]
        self.optimizer = optimizer
        self.n = len(paradigm.split(",")) if isinstance(paradigm, str) else len(paradigm)
        if strategy not in ("gradient_accumulation", "vanilla"):
            raise ValueError(f"unknown gradient_strategy {strategy!r}")
        if strategy == "gradient_accumulation" and self.n <= 1:
            raise AssertionError("gradient_accumulation needs an iterative paradigm of more than one dataset")   # engine.py:139
        self.strategy, self.max_norm = strategy, max_norm
        self.clip = clip or (lambda ps, m: torch.nn.utils.clip_grad_norm_(ps, m))
        self.i = 0
        self.pending = None
        self.acc = None

    def _step(self):
        if self.max_norm and self.max_norm > 0:
            self.clip(self.params, self.max_norm)
        self.optimizer.step()

    def add(self, loss):
        """one batch's weighted loss (a scalar with its autograd graph); returns True when the optimiser stepped"""
        self.i += 1
        if self.strategy == "vanilla":
            self.optimizer.zero_grad()
            loss.backward()
            self._step()
            return True
        pos = self.i % self.n
        if pos == 1:
            self.pending = loss
            return False
        self.pending = self.pending + loss
        if pos != 0:
            return False
        self.optimizer.zero_grad()
        self.pending.backward()
        self.pending = None
        self._step()
        return True

    def add_gradients(self, step=None):
        """the current `.grad` of the parameters is this batch's gradient; `step(params)`: custom clip + optimiser call"""
        self.i += 1
        last = self.strategy == "vanilla" or self.i % self.n == 0
        first = self.strategy == "vanilla" or self.i % self.n == 1
        if self.strategy == "gradient_accumulation":
            # float32 sums; a parameter that received no gradient in ANY batch of the round keeps `.grad = None`, so the
            # optimiser skips it as torch.optim.AdamW does after the reference's single backward (engine.py:136-153)
            if first:
                self.acc = [None] * len(self.params)
            for k, p in enumerate(self.params):
                if p.grad is None:
                    continue
                g = p.grad.detach().float()
                self.acc[k] = g.clone() if self.acc[k] is None else self.acc[k].add_(g)
            if not last:
                return False
            for p, a in zip(self.params, self.acc):
                p.grad = None if a is None else a.to(p.dtype)
            self.acc = None
        if step is not None:
            step(self.params)
        else:
            self._step()
        return True


def train_round(step_module, criterion, optimizer, batches, paradigm, strategy="gradient_accumulation", max_norm=0.1,
                autocast_dtype=None, state=None):
    """One round of the mixed-dataset protocol (BASELINE config 5; reference engine.py:99-166 with
    --gradient_strategy gradient_accumulation): `batches` = the `len(paradigm)` batches of the round in paradigm order
    (one per dataset slot, their image sizes and text lists may differ).  Each batch runs forward + backward on its own
    (one batch of activations alive at a time), the gradients are summed in float32 and ONE clipping + optimiser step
    closes the round -- the same update as the reference's backward over the summed losses (AccumulatedUpdate).
    `step_module` may be any step object train_step takes (plain / DDP module, GraphedStep, EagerSyncStep,
    GraphedStepCache): data-parallel gradients are averaged per batch, a SUM left for the optimiser (`grad_scale`) is scaled
    once, on the summed gradients.  Returns the list of the batches' weighted losses (device tensors)."""
    params = [p for p in step_module.parameters() if p.requires_grad]
    upd = state if state is not None else AccumulatedUpdate(params, optimizer, paradigm, strategy=strategy, max_norm=max_norm)
    losses = []
    for batch in batches:
        loss, grad_scale = _forward_backward(step_module, criterion, optimizer, batch, autocast_dtype)
        upd.add_gradients(step=lambda ps, gs=grad_scale: _optimizer_step(optimizer, max_norm, gs, params=ps))
        losses.append(loss.detach())
    return losses
