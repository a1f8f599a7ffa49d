"""GPU parity tests of the modules and the full model: the reference-generated goldens of
tests/test_modules_cpu.py, run on cuda:0 with the real HIP op (no oracle substitution).

Tolerances (north star): logits 1e-3 rel in float32, boxes 1e-4 abs.  The MSDeformAttn module case
runs in float64 (generic HIP kernel); everything else in float32 (quad / scatter kernels), and the
full model additionally in bfloat16 autocast-free mode against a looser, documented tolerance.
"""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import test_modules_cpu as C  # noqa: E402
from model_fill import fill_closed_form  # noqa: E402

from rlipv2_amd import alif, blocks, decoder, deform_attn, encoder, msda, parseda  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def dev(g):
    return {k: v.to(DEV) for k, v in g.items()}


def test_hip_op_is_the_one_in_use():
    assert deform_attn.msda_function is msda.MSDeformAttnFunction


@pytest.mark.parametrize("nd", [2, 4])
def test_msdeformattn_module_f64(nd):
    g = dev(C.load(f"msdeformattn_{nd}d"))
    m = deform_attn.MSDeformAttn(256, 4, 8, 4).double()
    fill_closed_form(m)
    with torch.no_grad():
        m.sampling_offsets.weight.mul_(0.3)
    m = m.to(DEV)
    shapes, starts = [t.to(DEV) for t in C.level_meta()]
    query = g["query"].clone().requires_grad_(True)
    inp = g["inp"].clone().requires_grad_(True)
    out = m(query, g["ref"], inp, shapes, starts, g["mask"])
    C.close(out.cpu(), g["out"].cpu(), 1e-9, 1e-11, "out")
    out.backward(g["go"])
    C.close(query.grad.cpu(), g["g_query"].cpu(), 1e-7, 1e-9, "g_query")
    C.close(inp.grad.cpu(), g["g_inp"].cpu(), 1e-7, 1e-9, "g_inp")


@pytest.mark.parametrize("last_vis", [1, 0])
def test_encoder_f32(last_vis):
    g = dev(C.load(f"encoder_lastvis{last_vis}"))
    enc = C._encoder(bool(last_vis)).to(DEV)
    shapes, starts = [t.to(DEV) for t in C.level_meta()]
    src = g["src"].clone().requires_grad_(True)
    lang = g["lang"].clone().requires_grad_(True)
    img, lng = enc(src, shapes, starts, g["valid_ratios"], g["pos"], g["mask"], lang_hidden=lang,
                   lang_masks=g["lmask"])
    C.close(img.cpu(), g["img"].cpu(), what="img_memory")
    C.close(lng.cpu(), g["lng"].cpu(), what="lang")
    (img * g["gi"]).sum().add((lng * g["gl"]).sum()).backward()
    C.close(src.grad.cpu(), g["g_src"].cpu(), 1e-3, 1e-5, "g_src")
    C.close(lang.grad.cpu(), g["g_lang"].cpu(), 1e-3, 1e-5, "g_lang")


@pytest.mark.parametrize("parse", [1, 0])
def test_dab_decoder_f32(parse):
    g = dev(C.load(f"decoder_parse{parse}"))
    layer = decoder.DeformableTransformerDecoderLayer(256, 512, 0.0, "relu", 4, 8, 4)
    dec = decoder.DABDeformableTransformerDecoderHOI(layer, 2, True, use_dab=True, d_model=256,
                                                     ParSe=bool(parse)).eval()
    dec.sub_bbox_embed = encoder._clones(blocks.MLP(256, 256, 4, 3), 2)
    dec.obj_bbox_embed = encoder._clones(blocks.MLP(256, 256, 4, 3), 2)
    fill_closed_form(dec)
    with torch.no_grad():
        for l in dec.layers:
            l.cross_attn.sampling_offsets.weight.mul_(0.3)
    dec = dec.to(DEV)
    shapes, starts = [t.to(DEV) for t in C.level_meta()]
    tgt = g["tgt"].clone().requires_grad_(True)
    src = g["src"].clone().requires_grad_(True)
    hs, inter = dec(tgt, (g["ref_sub"], g["ref_obj"]), src, shapes, starts, g["valid_ratios"], query_pos=None,
                    src_padding_mask=g["mask"])
    C.close(hs.cpu(), g["hs"].cpu(), what="hs")
    C.close(inter.cpu(), g["inter"].cpu(), 0.0, 1e-4, "refined boxes")
    (hs * g["gh"]).sum().backward()
    C.close(tgt.grad.cpu(), g["g_tgt"].cpu(), 1e-3, 1e-5, "g_tgt")
    C.close(src.grad.cpu(), g["g_src"].cpu(), 1e-3, 1e-5, "g_src")


def test_full_parseda_f32():
    g = C.load("parseda")
    model, bb = C.build_small_parseda()
    model = model.to(DEV)
    mc, out, feats, _ = C.run_small_parseda(model, bb, g, device=DEV)
    C.close(mc["img_memory"].cpu(), g["img_memory"], what="img_memory")
    loss = 0
    for k in C.KEYS:
        box = "boxes" in k
        C.close(out[k].cpu(), g[k], 1e-3 if not box else 0.0, 1e-5 if not box else 1e-4, k)
        C.close(out["aux_outputs"][0][k].cpu(), g["aux0_" + k], 1e-3 if not box else 0.0, 1e-5 if not box else 1e-4,
                "aux " + k)
        loss = loss + (out[k] * g["g_" + k].to(DEV)).sum() + (out["aux_outputs"][0][k] * g["g_" + k].to(DEV)).sum() * 0.5
    loss.backward()
    for i, (t, _) in enumerate(feats):
        C.close(t.grad.cpu(), g[f"g_feat{i}"], 1e-3, 1e-5, f"g_feat{i}")
    params = dict(model.named_parameters(remove_duplicate=False))
    for key in g:
        if key.startswith("gparam_") and g[key].numel():
            name = key[len("gparam_"):].replace("__", ".")
            C.close(params[name].grad.cpu(), g[key], 1e-3, 1e-5, "grad " + name)


def test_full_parseda_bf16_against_the_f32_reference_golden():
    """The headline dtype at model level, OUTPUTS against the float32 reference golden (two images of different size, 5 outputs +
    first auxiliary layer; measured with tools/bf16_parity_probe.py on MI355X in round 3, margins ~1.5-3x).  bf16 policy:
    weights / activations / value in bfloat16; sampling geometry, softmax, bilinear weights and accumulators in float32.
      logits  max |err| <= 2e-2 x max |reference|     (measured: subject / object 4.2e-3, verb 1.24e-2)
      boxes   max |err| <= 5e-3 absolute               (measured: 8.0e-4 / 1.8e-3)
    and the GRADIENTS of the golden's loss against the reference golden at round 3's hardware-validated bar (cosine >= 0.88,
    below); tests/test_zz_round4_gpu.py adds the tighter check against a float32 run of this model on the bf16-rounded weights
    with the sampling pinned (cosine >= 0.95).  Parity proper is claimed in float32 (test_full_parseda_f32: logits 1e-3 rel, boxes 1e-4 abs)."""
    g = C.load("parseda")
    model, bb = C.build_small_parseda()
    model = model.to(DEV).to(torch.bfloat16)
    gb = {k: (v.to(torch.bfloat16) if v.dtype == torch.float32 else v) for k, v in g.items()}
    gb["img_mask"] = g["img_mask"]
    mc, out, feats, _ = C.run_small_parseda(model, bb, gb, device=DEV)
    loss = 0
    for k in C.KEYS:
        for got, ref, what in ((out[k], g[k], k), (out["aux_outputs"][0][k], g["aux0_" + k], "aux " + k)):
            err = (got.float().cpu() - ref).abs().max().item()
            tol = 5e-3 if "boxes" in k else 2e-2 * ref.abs().max().item()
            assert err <= tol, (what, err, tol)
        loss = loss + (out[k].float() * g["g_" + k].to(DEV)).sum() + (out["aux_outputs"][0][k].float() * g["g_" + k].to(DEV)).sum() * 0.5
    loss.backward()

    # GRADIENTS against the REFERENCE golden (independent of this repository's other code paths): the hardware-validated bar of
    # round 3, cosine >= 0.88 for the three feature maps and every sentinel parameter (measured then: features 0.920-0.990,
    # parameters 0.905-1.000; the low ones pass the sampling locations' floor() kinks, which one bf16 rounding moves across).
    # The tighter check with the sampling pinned (>= 0.95, against a float32 run of this model) is tests/test_zz_round4_gpu.py;
    # this one stays because a systematic error shared by two of this repository's routes would pass that and fail here.
    def cosine(a, b):
        a, b = a.double().flatten(), b.double().flatten()
        return float((a @ b) / (a.norm() * b.norm() + 1e-300))

    for i, (t, _) in enumerate(feats):
        assert torch.isfinite(t.grad.float()).all()
        c = cosine(t.grad.float().cpu(), g[f"g_feat{i}"])
        assert c >= 0.88, (f"g_feat{i}", c)
    params = dict(model.named_parameters(remove_duplicate=False))
    for key in g:
        if key.startswith("gparam_") and g[key].numel():
            name = key[len("gparam_"):].replace("__", ".")
            c = cosine(params[name].grad.float().cpu(), g[key])
            assert c >= 0.88, (name, c)


@pytest.mark.parametrize("nd", [2, 4])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fused_sampling_geometry_equals_module_arithmetic(nd, dtype):
    """csrc/msda_prep.hip against the module's own PyTorch arithmetic (the reference's formula,
    ms_deform_attn.py:101-112): outputs and the gradients of query, value input and reference points."""
    g = dev(C.load(f"msdeformattn_{nd}d"))
    m = deform_attn.MSDeformAttn(256, 4, 8, 4)
    fill_closed_form(m)
    with torch.no_grad():
        m.sampling_offsets.weight.mul_(0.3)
    m = m.to(DEV).to(dtype)
    shapes, starts = [t.to(DEV) for t in C.level_meta()]
    res = {}
    for fused in (False, True):
        deform_attn.fused_geometry = fused
        try:
            query = g["query"].to(dtype).clone().requires_grad_(True)
            inp = g["inp"].to(dtype).clone().requires_grad_(True)
            ref = g["ref"].float().clone().requires_grad_(True)
            out = m(query, ref, inp, shapes, starts, g["mask"])
            out.backward(g["go"].to(dtype))
            res[fused] = [t.float() for t in (out, query.grad, inp.grad, ref.grad)]
        finally:
            deform_attn.fused_geometry = True
    tol = 3e-2 if dtype == torch.bfloat16 else 1e-4       # bf16: the unfused route rounds offsets/normaliser to bf16
    for name, a, b in zip(("out", "g_query", "g_inp", "g_ref"), res[True], res[False]):
        err = (a - b).abs().max().item() / max(1e-6, b.abs().max().item())
        assert err < tol, (name, err)


def test_full_parsed_v2_f32():
    g = C.load("parsed")
    model, bb = C.build_small_parsed()
    model = model.to(DEV)
    out, feats = C.run_small_parsed(model, bb, g, device=DEV)
    loss = 0
    for k in C.KEYS:
        box = "boxes" in k
        C.close(out[k].cpu(), g[k], 1e-3 if not box else 0.0, 1e-5 if not box else 1e-4, k)
        loss = loss + (out[k] * g["g_" + k].to(DEV)).sum() + (out["aux_outputs"][0][k] * g["g_" + k].to(DEV)).sum() * 0.5
    loss.backward()
    for i, (t, _) in enumerate(feats):
        C.close(t.grad.cpu(), g[f"g_feat{i}"], 1e-3, 1e-5, f"g_feat{i}")


def test_graphed_step_delivers_the_same_gradients_as_eager_autograd():
    """train.GraphedStep (forward graph + backward graph, parameter gradients handed over outside autograd)
    against the eager step on the same model and batch: outputs, loss and every parameter gradient.
    eval() mode: the hidden dropouts (VLFuse / RoBERTa / FeatureResizer) would draw different masks."""
    from rlipv2_amd import parseda, train
    torch.manual_seed(0)
    margs = parseda.default_args(num_queries=40, enc_layers=4, dec_layers=2)       # n_fusions == dec_layers
    model, criterion = train.build_training(margs, device="cuda:0", with_text_encoder=True)
    train.to_bf16(model)
    batch = train.synthetic_batch(2, 256, 320, device="cuda:0", triplets=3)
    batch[0].tensors = batch[0].tensors.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    step = train.ParSeDATrainStep(model)
    model.eval()
    train.freeze_parameters_without_gradient(step, criterion, batch)
    params = [(n, p) for n, p in step.named_parameters() if p.requires_grad]

    def run(step_module):
        for _, p in params:
            p.grad = None
        out = step_module(*batch)
        loss = criterion.weighted_sum(criterion(out, batch[2]))
        loss.backward()
        if isinstance(step_module, train.GraphedStep):
            step_module.backward()
        return loss.detach().float(), {n: p.grad.detach().float().clone() for n, p in params}

    loss_e, grads_e = run(step)
    loss_e2, grads_e2 = run(step)                               # run-to-run noise floor of the eager step itself

    def distance(ga, gb):
        """relative L2 distance of the whole gradient, and the worst per-parameter one among parameters whose
        gradient is not pure noise (key biases have an exactly zero true gradient: softmax shift invariance)"""
        num = sum(float((ga[n] - gb[n]).norm()) ** 2 for n in ga) ** 0.5
        den = sum(float(gb[n].norm()) ** 2 for n in gb) ** 0.5
        return num / den

    noise = distance(grads_e2, grads_e)
    graphed = train.graph_step_module(step, model, batch)
    for _ in range(2):                                           # replays must be repeatable
        loss_g, grads_g = run(graphed)
        torch.testing.assert_close(loss_g, loss_e, rtol=2e-2, atol=1e-3)
        assert set(grads_g) == set(grads_e)
        # same kernels in the same order: only the atomics-order noise of the MSDA scatter (which also
        # separates two eager runs) may separate the graphed from the eager gradients
        d = distance(grads_g, grads_e)
        assert d <= 2.0 * noise + 1e-2, (d, noise)

    # the criterion captured as well: forward graph ends with the cost matrices, backward graph starts with the
    # losses; same loss dict, same gradients
    ref_ld = criterion(step(*batch), batch[2])
    full = train.graph_step_module(step, model, batch, criterion=criterion)
    for _ in range(2):
        for _, p in params:
            p.grad = None
        ld, total = full.run(*batch)
        torch.testing.assert_close(total.float(), loss_e, rtol=2e-2, atol=1e-3)
        assert set(ld.keys()) == set(ref_ld.keys())
        for k in ref_ld:
            if k.startswith("loss_"):           # (class / cardinality errors are argmax counts: bf16 noise flips them)
                torch.testing.assert_close(ld[k].float(), ref_ld[k].float(), rtol=5e-2, atol=2e-2,
                                           msg=lambda m: f"{k}: {m}")
        grads_f = {n: p.grad.detach().float().clone() for n, p in params}
        d = distance(grads_f, grads_e)
        assert d <= 2.0 * noise + 1e-2, (d, noise)


def test_swin_backbone_on_gpu_matches_reference_golden():
    """swin.SwinTransformer on the GPU (float32: ROCm's scaled_dot_product_attention with the additive
    relative-position / shift mask) against the reference golden; bf16 within a bf16 band of the float32 run."""
    sys.path.insert(0, C.GOLD)
    from make_model_golden import SWIN_CFG, swin_input
    from rlipv2_amd import swin
    g = C.load("swin")
    m = swin.SwinTransformer(**{k: (tuple(v) if isinstance(v, list) else v) for k, v in SWIN_CFG.items()}).eval()
    fill_closed_form(m)
    m = m.to(DEV)
    x = swin_input().to(DEV).requires_grad_(True)
    outs = m(x)
    total = 0
    for i, (k, v) in enumerate(sorted(outs.items())):
        C.close(v.cpu(), g[k], 1e-3, 1e-4, k)
        gen = torch.Generator().manual_seed(90 + i)
        total = total + (v * torch.randn(*v.shape, generator=gen, dtype=torch.float64).float().to(DEV)).sum()
    total.backward()
    C.close(x.grad.cpu(), g["g_x"], 5e-3, 1e-4, "g_x")
    mb = m.to(torch.bfloat16)
    ob = mb(x.detach().to(torch.bfloat16))
    for k, v in ob.items():
        ref = g[k]
        rel = float((v.float().cpu() - ref).norm() / ref.norm())
        assert rel < 0.05, (k, rel)


def test_graphed_data_parallel_step_on_one_rank_rccl():
    """The data-parallel flavour of the graphed step on a 1-rank RCCL group: parameters broadcast, gradients
    packed into the flat bf16 buffer inside the backward graph, one all-reduce, `p.grad` = views of the buffer
    in the parameters' own layouts, and a fused optimiser step on them.  (Rank counts > 1 are covered on CPU
    over gloo, tests/test_dp_cpu.py; this checks the RCCL / graph plumbing on the device.)"""
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
        model.eval()
        train.broadcast_parameters(model, 0)
        train.freeze_parameters_without_gradient(step, criterion, batch)
        params = [p for p in step.parameters() if p.requires_grad]
        sync = train.GradientSynchronizer(params)
        graphed = train.graph_step_module(step, model, batch, synchronizer=sync, criterion=criterion)
        opt = train.FusedMasterAdamW(model)
        before = [p.detach().float().clone() for p in params[:20]]
        for _ in range(2):
            loss = train.train_step(graphed, criterion, opt, batch, autocast_dtype=None)
        assert torch.isfinite(loss)
        lo, hi = sync.flat.data_ptr(), sync.flat.data_ptr() + sync.flat.numel() * 2
        for p in params:
            assert p.grad is not None and lo <= p.grad.data_ptr() < hi          # views of the flat buffer
            assert p.grad.shape == p.shape and p.grad.stride() == p.stride()
        assert float(sync.flat.float().abs().sum()) > 0
        assert any(not torch.equal(b, p.detach().float()) for b, p in zip(before, params[:20]))
    finally:
        if created:
            dist.destroy_process_group()


def test_overlapped_bucketed_all_reduce_inside_the_backward_graph():
    """GraphedStep(overlap=True): tensor hooks + bucketed RCCL all-reduces captured INSIDE the backward graph on a
    communication stream (GradientSynchronizer.hooked) against the flat schedule (one all-reduce after the replay), on
    a 1-rank RCCL group in eval mode.  Same autograd pass, only the delivery of the gradients differs (exactness of
    the schedule itself: tests/test_dp_cpu.py on 2 gloo ranks).

    Round 2 compared with a 2 % band because two runs of the SAME step differed by 3.6 %; tools/nondet_modules.py has
    since named the source: not "atomics of the attention backward" but ONE MIOpen convolution per model size (forward of
    `input_proj.3.0` at the bench size, of `backbone...layer2.0.conv2` at this size) -- every hand-written kernel,
    hipBLASLt GEMM and fused attention in the step is bit-repeatable.  With MIOpen restricted to its deterministic
    solvers this small step is repeatable bit for bit (profiles/r03_nondeterminism.txt), so the schedules are compared
    to 1e-3 and a second capture of the same schedule must reproduce the first exactly."""
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
        det_before = torch.backends.cudnn.deterministic
        torch.backends.cudnn.deterministic = True                 # MIOpen: deterministic solvers only (see above)
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
        got = {}
        for overlap in (False, "again", True):
            sync = train.GradientSynchronizer(params, bucket_bytes=32 << 20)
            graphed = train.GraphedStep(step, model, batch, synchronizer=sync, criterion=criterion, overlap=overlap is True)
            assert graphed.overlap == (overlap is True)
            for _ in range(3):
                _, total = graphed.run(*batch)
            torch.cuda.synchronize()
            got[overlap] = ([p.grad.detach().float().clone() for p in params], float(total))
            if overlap is True:
                assert len(sync.buckets) >= 4, [len(b) for b in sync.buckets]
                # arrival order: heads / decoders first; the text encoder (issued after the backbone in the forward
                # pass, so that its backward comes first) before the backbone, whose first convolution is last
                assert not any("backbone" in names[i] or "text_encoder" in names[i] for i in sync.buckets[0])
                first_text = min(k for k, b in enumerate(sync.buckets) if any("text_encoder" in names[i] for i in b))
                first_bb = min(k for k, b in enumerate(sync.buckets) if any("backbone" in names[i] for i in b))
                assert first_text <= first_bb and "backbone" in names[sync.buckets[-1][-1]]
                lo, hi = sync.flat.data_ptr(), sync.flat.data_ptr() + sync.flat.numel() * 2
                assert all(lo <= p.grad.data_ptr() < hi and p.grad.data_ptr() % 16 == 0 for p in params)
        assert abs(got[True][1] - got[False][1]) <= 1e-3 * abs(got[False][1])
        def distance(x, y):
            num = sum(float((a - b).pow(2).sum()) for a, b in zip(x, y))
            return (num / sum(float(b.pow(2).sum()) for b in y)) ** 0.5

        noise = distance(got["again"][0], got[False][0])        # a second capture of the SAME flat schedule
        assert noise <= 1e-6, noise                               # (deterministic convolutions: nothing left to differ)
        d = distance(got[True][0], got[False][0])
        assert d <= 1e-3, (d, noise)                              # a wrong or missing bucket would be >= its share of the norm
        worst = max(float((a - b).abs().max()) / max(float(b.abs().max()), 1e-12) for a, b in zip(got[True][0], got[False][0]))
        assert worst <= 1e-2, worst                               # and no single parameter's gradient is off
    finally:
        torch.backends.cudnn.deterministic = det_before
        if created:
            dist.destroy_process_group()


def test_criterion_and_matcher_match_reference_on_gpu():
    """The stacked, stage-split criterion that the bench captures (prepare -> assign -> losses, criterion.py) on
    the GPU against the reference-generated golden: every loss entry, the weighted total and the gradients of
    every prediction (reference: hoi.py:3627-4766, matcher.py:95-270)."""
    sys.path.insert(0, C.GOLD)
    from make_model_golden import criterion_case
    from rlipv2_amd import criterion as MC
    g = C.load("criterion")
    main, aux, targets, _ = criterion_case()
    main = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in main.items()}
    aux = [{k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in o.items()} for o in aux]
    targets = [{k: v.to(DEV) for k, v in t.items()} for t in targets]
    for o in [main] + aux:
        for k in o:
            if k.startswith("pred_"):
                o[k].requires_grad_(True)
    out = dict(main)
    out["aux_outputs"] = aux
    crit = MC.SetCriterionHOI(MC.HungarianMatcherHOI(1, 1, 2.5, 1, subject_class=True), MC.build_weight_dict(2)).to(DEV)
    state = crit.prepare(out, targets)
    index = crit.assign(state).to(DEV)
    ld = crit.losses(state, index, crit._num_interactions(state["sizes"], state["dev"]))
    ref_keys = {k[len("loss_"):] for k in g if k.startswith("loss_")}
    assert set(ld.keys()) == ref_keys
    for k in ref_keys:
        C.close(ld[k].reshape(()).cpu(), g["loss_" + k].reshape(()), 1e-5, 1e-6, k)
    total = crit.weighted_sum(ld)
    C.close(total.reshape(()).cpu(), g["total"].reshape(()), 1e-5, 1e-6, "total")
    total.backward()
    for li, o in enumerate([main] + aux):
        for k in o:
            if k.startswith("pred_"):
                C.close(o[k].grad.cpu(), g[f"g_L{li}_{k}"], 1e-4, 1e-6, f"grad L{li} {k}")


@pytest.mark.parametrize("nd", [2, 4])
def test_msdeformattn_module_f32_fused_route_vs_golden(nd):
    """The fused sampling-geometry route of the module (msda_prep.hip + quad forward + K1 + destination-stationary
    backward) against the float64 reference golden, in float32."""
    g = dev(C.load(f"msdeformattn_{nd}d"))
    m = deform_attn.MSDeformAttn(256, 4, 8, 4)
    fill_closed_form(m)
    with torch.no_grad():
        m.sampling_offsets.weight.mul_(0.3)
    m = m.to(DEV)
    shapes, starts = [t.to(DEV) for t in C.level_meta()]
    query = g["query"].float().clone().requires_grad_(True)
    inp = g["inp"].float().clone().requires_grad_(True)
    out = m(query, g["ref"].float(), inp, shapes, starts, g["mask"])
    C.close(out.cpu(), g["out"].float().cpu(), 1e-3, 1e-5, "out")
    out.backward(g["go"].float())
    C.close(query.grad.cpu(), g["g_query"].float().cpu(), 2e-3, 1e-4, "g_query")
    C.close(inp.grad.cpu(), g["g_inp"].float().cpu(), 2e-3, 1e-4, "g_inp")


def test_swin_large_small_train_step_smoke():
    """BASELINE config 4's model family (Swin-L backbone, batch 2) through one eager bf16 master-weight train step
    at a small image size: finite loss, finite non-zero gradients on backbone, encoder and decoder parameters."""
    from rlipv2_amd import train
    torch.manual_seed(0)
    margs = parseda.default_args(num_queries=40)
    model, criterion = train.build_training(margs, device=DEV, with_text_encoder=True, backbone_name="swin_large")
    batch = train.synthetic_batch(2, 256, 320, n_obj=13, n_verb=7, triplets=3, device=DEV, seed=0)
    train.to_bf16(model)
    batch[0].tensors = batch[0].tensors.to(torch.bfloat16)
    step = train.ParSeDATrainStep(model)
    model.train()
    out = step(*batch)
    ld = criterion(out, batch[2])
    loss = criterion.weighted_sum(ld)
    assert torch.isfinite(loss)
    loss.backward()
    seen = {"backbone": 0, "encoder": 0, "decoder": 0}
    for n, p in model.named_parameters():
        if p.grad is None:
            continue
        assert torch.isfinite(p.grad).all(), n
        for k in seen:
            if k in n and float(p.grad.abs().max()) > 0:
                seen[k] += 1
    assert all(v > 0 for v in seen.values()), seen


def test_non_finite_guard_raises_one_step_late_without_sync():
    from rlipv2_amd import train
    guard = train.NonFiniteGuard()
    guard.submit(torch.tensor(2.0, device=DEV))
    guard.submit(torch.tensor(float("nan"), device=DEV))     # the previous (finite) step is checked here
    torch.cuda.synchronize()
    with pytest.raises(train.NonFiniteLoss):
        guard.submit(torch.tensor(1.0, device=DEV))
    with pytest.raises(train.NonFiniteLoss):
        g2 = train.NonFiniteGuard()
        g2.submit(torch.tensor(float("inf"), device=DEV))
        g2.check(wait=True)


def test_graphed_step_cache_recaptures_per_bucket():
    """A second bucket (other target counts, padded batch) is captured on first sight instead of raising; replays of
    either bucket give the loss of the eager step on the same batch."""
    from rlipv2_amd import train
    torch.manual_seed(0)
    margs = parseda.default_args(num_queries=40)
    model, criterion = train.build_training(margs, device=DEV, with_text_encoder=True)
    train.to_bf16(model)
    step = train.ParSeDATrainStep(model)
    model.train()
    for mod in model.modules():                                  # dropout off: eager and replay must agree
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0

    def make(triplets, pad, seed):
        b = train.synthetic_batch(2, 256, 320, n_obj=13, n_verb=7, triplets=triplets, device=DEV, seed=seed)
        b[0].tensors = b[0].tensors.to(torch.bfloat16)
        if pad:
            b[0].mask[1, :, 280:] = True
            b[0].tensors[1, :, :, 280:] = 0
            b[0].no_padding = False
        return b

    train.freeze_parameters_without_gradient(step, criterion, make(3, False, 0))
    cache = train.GraphedStepCache(step, model, criterion=criterion)
    for batch in (make(3, False, 0), make(5, True, 1), make(3, False, 2), make(5, True, 3)):
        with torch.no_grad():
            eager = criterion.weighted_sum(criterion(step(*batch), batch[2])).float()
        torch.cuda.synchronize()
        g = cache.get(batch)
        _, total = g.run(*batch)
        torch.testing.assert_close(total.float(), eager, rtol=3e-2, atol=3e-2)
    assert cache.captures == 2 and len(cache.graphs) == 2


# ---- ALIF attention core on the HIP kernel (csrc/alif_attention.hip) -----------------------------------------------------
def _vlfuse_bf16(gating):
    g = dev(C.load(f"vlfuse_{gating}"))
    m = alif.RLIPv2_VLFuse(parseda.default_args(gating_mechanism=gating)).eval()
    fill_closed_form(m)
    return g, m.to(DEV).to(torch.bfloat16)


def _run_vlfuse(m, g, fused):
    alif.fused_attention = fused
    try:
        v = g["v"].to(torch.bfloat16).clone().requires_grad_(True)
        l = g["l"].to(torch.bfloat16).clone().requires_grad_(True)
        out = m({"visual": {"src": v, "padding_mask": g["vmask"], "pos": g["pos"].to(torch.bfloat16)},
                 "lang": {"hidden": l, "masks": g["lmask"]}})
        ov, ol = out["visual"]["src"], out["lang"]["hidden"]
        (ov.float() * g["gv"]).sum().add((ol.float() * g["gl"]).sum()).backward()
        grads = {n: p.grad.float().clone() for n, p in m.named_parameters() if p.grad is not None}
        for p in m.parameters():
            p.grad = None
        torch.cuda.synchronize()
        return ov.float(), ol.float(), v.grad.float(), l.grad.float(), grads
    finally:
        alif.fused_attention = True


@pytest.mark.parametrize("gating", ["VXAc", "XGating"])
def test_alif_fused_attention_kernel_vs_reference_golden_and_pytorch_route(gating):
    """RLIPv2_VLFuse in bfloat16 with the attention core on the HIP kernel (one launch: q k^T, both softmaxes, both
    value products on MFMA) against (a) the reference-generated float32 golden (models/fuse_helper.py:983-1096) within
    the bfloat16 band the PyTorch route itself keeps, and (b) the PyTorch route of the same bfloat16 module: outputs
    and input / parameter gradients (the backward of the fused route is hand-written from the saved probabilities)."""
    g, m = _vlfuse_bf16(gating)
    assert m.b_attn.attn._fused_ok(g["v"].to(torch.bfloat16), g["l"].to(torch.bfloat16))
    fo = _run_vlfuse(m, g, True)
    po = _run_vlfuse(m, g, False)
    refs = (g["out_v"], g["out_l"], g["g_v"], g["g_l"])
    for name, a, b, ref in zip(("out_v", "out_l", "g_v", "g_l"), fo[:4], po[:4], refs):
        scale = float(ref.abs().max())
        err_f, err_p = float((a - ref).abs().max()) / scale, float((b - ref).abs().max()) / scale
        assert err_f <= max(2.0 * err_p, 2e-2), (name, err_f, err_p)      # as close to the reference as bf16 PyTorch is
        assert float((a - b).abs().max()) / scale <= 3e-2, (name, float((a - b).abs().max()) / scale)
    # parameter gradients: every tensor the PyTorch route differentiates, same values to bf16 accuracy
    assert set(fo[4]) == set(po[4])
    gmax = max(float(t.abs().max()) for t in po[4].values())
    for n in po[4]:
        err = float((fo[4][n] - po[4][n]).abs().max())
        # (per tensor, or -- for the scalar gates, whose gradient is one long cancelling bf16 sum -- on the global scale)
        assert err <= max(4e-2 * float(po[4][n].abs().max()), 5e-3 * gmax), (n, err)


def test_alif_fused_attention_kernel_direct_vs_float64_and_dropout():
    """The kernel alone (C ABI through AlifAttentionFunction) at the train step's shape (B=4, H=8, Tv=273, Tl=64):
    outputs against a float64 restatement of fuse_helper.py:395-462 on the same bfloat16 inputs; with dropout the kept
    probabilities are scaled by 1 / (1 - p), the dropped ones contribute nothing (checked through the saved masks)."""
    torch.manual_seed(3)
    B, H, Tv, Tl, hd = 4, 8, 273, 64, 256
    E = H * hd
    q = (torch.randn(B, Tv, E, device=DEV) * 0.08).bfloat16()
    k = torch.randn(B, Tl, E, device=DEV).bfloat16()
    vl = torch.randn(B, Tl, E, device=DEV).bfloat16()
    vv = torch.randn(B, Tv, E, device=DEV).bfloat16()
    Tvp = (Tv + 31) // 32 * 32
    vlt = vl.transpose(1, 2).contiguous()
    vvt = torch.nn.functional.pad(vv, (0, 0, 0, Tvp - Tv), value=7.0).transpose(1, 2).contiguous()   # junk in the padding
    heads = lambda t, T: t.double().view(B, T, H, hd).transpose(1, 2)
    S = heads(q, Tv) @ heads(k, Tl).transpose(-1, -2)
    pv, pl = S.softmax(-1), S.transpose(-1, -2).softmax(-1)
    ref_v = (pv @ heads(vl, Tl)).transpose(1, 2).reshape(B, Tv, E)
    ref_l = (pl @ heads(vv, Tv)).transpose(1, 2).reshape(B, Tl, E)
    out_v, out_l = alif.AlifAttentionFunction.apply(q, k, vlt, vvt, H, 0.1, False)
    torch.cuda.synchronize()
    for got, ref in ((out_v, ref_v), (out_l, ref_l)):
        assert float((got.double() - ref).abs().max()) <= 2e-2 * float(ref.abs().max())
    # dropout: reproduce the kernel's output from the probabilities and masks it saved
    qg = q.clone().requires_grad_(True)
    out_v, out_l = alif.AlifAttentionFunction.apply(qg, k, vlt, vvt, H, 0.5, True)
    _, _, _, _, p_v, p_l, keep_v, keep_l = out_v.grad_fn.saved_tensors
    assert 0.4 < float(keep_v.float().mean()) < 0.6 and 0.4 < float(keep_l.float().mean()) < 0.6
    exp_v = ((p_v.double() * keep_v * 2.0) @ heads(vl, Tl)).transpose(1, 2).reshape(B, Tv, E)
    exp_l = ((p_l.double() * keep_l * 2.0) @ heads(vv, Tv)).transpose(1, 2).reshape(B, Tl, E)
    assert float((out_v.double() - exp_v).abs().max()) <= 2e-2 * float(exp_v.abs().max())
    assert float((out_l.double() - exp_l).abs().max()) <= 2e-2 * float(exp_l.abs().max())
    (out_v.float().sum() + out_l.float().sum()).backward()
    assert torch.isfinite(qg.grad).all()


# ---- f4 on the device: inference tail and checkpoint interop ------------------------------------------------------------
@pytest.mark.parametrize("variant", ["plain", "temperature", "zeroshot"])
def test_postprocess_hoi_on_gpu_matches_reference(variant):
    """PostProcessHOI (hoi.py:4769-4873) with the model outputs resident on the GPU (whole batch on the device, one
    D2H transfer) against the reference-generated golden: labels / ids exact, scores and boxes to float32 rounding."""
    sys.path.insert(0, C.GOLD)
    from make_model_golden import postprocess_case
    from rlipv2_amd.postprocess import PostProcessHOI
    g = C.load("postprocess")
    out, sizes = postprocess_case()
    out = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in out.items()}
    pp = PostProcessHOI(0, temperature=(variant == "temperature"), zero_shot_hoi_eval=(variant == "zeroshot"))
    res = pp(out, sizes.to(DEV))
    assert len(res) == 3
    for i, r in enumerate(res):
        for k in ("labels", "sub_ids", "obj_ids"):
            assert torch.equal(r[k].cpu(), torch.as_tensor(g[f"{variant}_{i}_{k}"])), (i, k)
        C.close(r["boxes"].cpu(), g[f"{variant}_{i}_boxes"], 1e-5, 1e-3, "boxes")
        C.close(r["verb_scores"].cpu(), g[f"{variant}_{i}_verb_scores"], 1e-5, 1e-7, "verb_scores")


def test_checkpoint_round_trip_with_the_fused_optimiser_on_gpu(tmp_path):
    """checkpoint.py on the device with the bf16 / float32-master policy of the train step: 'model' of the saved file
    holds the float32 masters in the reference's layout and names (main.py:599-629), a `--resume` into a fresh model +
    FusedMasterAdamW followed by one more step equals the uninterrupted run bit for bit; `--pretrained` semantics cut the
    learned queries (util/misc.py:479-490)."""
    from rlipv2_amd import checkpoint as CK, train

    def make():
        torch.manual_seed(2)
        margs = parseda.default_args(num_queries=20, enc_layers=4, dec_layers=2, pseudo_verb=False)
        model, criterion = train.build_training(margs, device=DEV, with_text_encoder=False)
        train.to_bf16(model)
        model.eval()
        return model, criterion

    mem = torch.tanh(torch.randn(10, 1, 768, generator=torch.Generator().manual_seed(9))).repeat(1, 2, 1).to(DEV).bfloat16()
    text = (~(mem.sum(-1) > 0), mem, torch.tensor([[6, 4]]))
    samples, _, targets = train.synthetic_batch(2, 128, 160, n_obj=6, n_verb=4, triplets=2, device=DEV, seed=4)
    samples.tensors = samples.tensors.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    batch = (samples, text, targets)

    m1, crit = make()
    step1 = train.ParSeDATrainStep(m1)
    train.freeze_parameters_without_gradient(step1, crit, batch)
    o1 = train.FusedMasterAdamW(m1)
    for _ in range(2):
        train.train_step(step1, crit, o1, batch, autocast_dtype=None)
    path = str(tmp_path / "ck.pth")
    CK.save_checkpoint(path, m1, o1, epoch=7)
    ck = torch.load(path, map_location="cpu", weights_only=False)
    assert ck["epoch"] == 7 and all(v.dtype == torch.float32 for k, v in ck["model"].items() if v.is_floating_point())
    assert set(ck["model"]) == set(m1.state_dict())
    train.train_step(step1, crit, o1, batch, autocast_dtype=None)
    torch.cuda.synchronize()

    m2, _ = make()
    step2 = train.ParSeDATrainStep(m2)
    train.freeze_parameters_without_gradient(step2, crit, batch)
    o2 = train.FusedMasterAdamW(m2)
    ck = torch.load(path, map_location=DEV, weights_only=False)
    CK.load_resume(m2, {"model": {k: (v.to(torch.bfloat16) if v.is_floating_point() else v) for k, v in ck["model"].items()}})
    o2.load_state_dict(ck["optimizer"])                     # float32 masters + moments: the state the update continues from
    train.train_step(step2, crit, o2, batch, autocast_dtype=None)
    torch.cuda.synchronize()
    worst = max(float((a.float() - b.float()).abs().max()) for a, b in zip(m1.parameters(), m2.parameters()))
    assert worst <= 1e-3, worst          # (the device backward is not bit-repeatable: attention atomics)


@pytest.mark.parametrize("parse", [True, False])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_decoder_glue_kernels_match_the_op_sequence(parse, dtype):
    """csrc/decoder_glue.hip against the PyTorch statement of the same float32 arithmetic (decoder.py's fallback path,
    reference deformable_transformer.py:1490-1541): the reference points and the refined boxes exactly (same operations,
    same order, same device math library), the sine features to one bf16 / float32 rounding of sin / cos."""
    from rlipv2_amd.blocks import inverse_sigmoid, sine_embed_for_position
    g = torch.Generator().manual_seed(5)
    N, n, L = 3, 37, 4
    sub = torch.rand(N, n, 4, generator=g).cuda()
    obj = torch.rand(N, n, 4, generator=g).cuda()
    obj[0, 0] = torch.tensor([0.0, 1.0, 1e-7, 1 - 1e-7])                       # clamp / eps branches of inverse_sigmoid
    ratios = (0.5 + 0.5 * torch.rand(N, L, 2, generator=g)).cuda()
    ratios4 = torch.cat([ratios, ratios], -1)[:, None]
    ref_in, feat = decoder.reference_embed(sub, obj, ratios, parse, dtype)
    if parse:
        want = torch.cat((sub[:, :, None] * ratios4, obj[:, :, None] * ratios4), dim=1)
    else:
        want = 0.5 * (sub + obj)[:, :, None] * ratios4
    assert torch.equal(ref_in, want)
    want_feat = sine_embed_for_position(want[:, :, 0, :])
    tol = 2.0 ** -8 if dtype == torch.bfloat16 else 2e-6
    assert (feat.float() - want_feat).abs().max() <= tol
    delta = (torch.randn(N, n, 4, generator=g) * 2).to(dtype).cuda()
    got = decoder.refine_boxes(delta, obj)
    want_box = (delta.float() + inverse_sigmoid(obj)).sigmoid()
    assert (got - want_box).abs().max() <= 1e-6
    assert got.dtype == torch.float32 and not got.requires_grad


@pytest.mark.gpu
def test_step_roofline_probe_runs_on_the_small_model():
    """bench.py's step-level roofline (T_mem, T_mfma of one eager train step) is produced by
    rlipv2_amd.roofline.probe under a TorchDispatchMode + FlopCounterMode; round 2 shipped a bench line without it
    because the probe raised inside a custom autograd function and the failure was only printed.  Here the probe runs
    on the small model (eager master-weight step, as bench.py issues it) and must return plausible numbers."""
    from rlipv2_amd import parseda, roofline, train
    torch.manual_seed(0)
    margs = parseda.default_args(num_queries=40, enc_layers=4, dec_layers=2)
    model, criterion = train.build_training(margs, device="cuda:0", with_text_encoder=True)
    train.to_bf16(model)
    batch = train.synthetic_batch(2, 256, 320, device="cuda:0", triplets=3)
    batch[0].tensors = batch[0].tensors.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    step = train.ParSeDATrainStep(model)
    train.freeze_parameters_without_gradient(step, criterion, batch)
    optimizer = train.FusedMasterAdamW(model)
    train.train_step(step, criterion, optimizer, batch, autocast_dtype=torch.bfloat16)
    r = roofline.probe(lambda: train.train_step(step, criterion, optimizer, batch, autocast_dtype=torch.bfloat16))
    assert r["bytes"] > 1e8 and r["flops"] > 1e9 and r["aten_ops"] > 100
    assert r["library_bytes"] > 0                      # the ctypes-bound kernels reported their operands
    assert 0 < r["T_mfma_s"] < r["T_mem_s"] < 1.0      # a memory-bound step


@pytest.mark.gpu
def test_config5_mixed_dataset_round_swin_large_with_text_encoder():
    """BASELINE config 5 (Swin-L + RoBERTa-shaped text encoder, mixed pseudo-SGG pre-training) at a small size: one round
    of the iterative paradigm "0,1,2" -- three batches of different image sizes and text-list lengths, as three datasets
    deliver them -- with gradient accumulation (train.train_round; protocol pinned against the reference on the CPU,
    tests/test_protocol_cpu.py): one optimiser step per round, finite losses, parameters of backbone, text encoder,
    encoder and decoders all move."""
    from rlipv2_amd import train
    torch.manual_seed(0)
    margs = parseda.default_args(num_queries=40, enc_layers=4, dec_layers=2)       # (fusion layers == decoder layers)
    model, criterion = train.build_training(margs, device=DEV, with_text_encoder=True, backbone_name="swin_large")
    train.to_bf16(model)
    batches = [train.synthetic_batch(2, 256, 320, n_obj=13, n_verb=7, triplets=3, device=DEV, seed=0),
               train.synthetic_batch(2, 224, 288, n_obj=9, n_verb=5, triplets=2, device=DEV, seed=1),
               train.synthetic_batch(2, 192, 256, n_obj=17, n_verb=4, triplets=4, device=DEV, seed=2)]
    for b in batches:
        b[0].tensors = b[0].tensors.to(torch.bfloat16)
    step = train.ParSeDATrainStep(model)
    model.train()
    train.freeze_parameters_without_gradient(step, criterion, batches[0])
    opt = train.FusedMasterAdamW(model)
    watch = {k: None for k in ("backbone", "text_encoder", "transformer.encoder", "ho_decoder", "verb_decoder")}
    for n, p in model.named_parameters():
        for k in watch:
            if watch[k] is None and k in n and p.requires_grad and p.numel() > 64:
                watch[k] = (n, p, p.detach().float().clone())
    assert all(v is not None for v in watch.values()), watch
    steps_before = opt.t
    losses = train.train_round(step, criterion, opt, batches, "0,1,2")
    torch.cuda.synchronize()
    assert len(losses) == 3 and all(bool(torch.isfinite(l)) for l in losses)
    assert opt.t == steps_before + 1                      # ONE update for the round
    for k, (n, p, before) in watch.items():
        assert not torch.equal(p.detach().float(), before), f"{n} did not move"
        assert bool(torch.isfinite(p.detach().float()).all())
