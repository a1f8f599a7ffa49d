"""Where the train step's time goes, by phase (profiling aid): CUDA events around the phases of one
eager step, averaged over a few steps."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlipv2_amd import parseda, train
from rlipv2_amd.blocks import NestedTensor

margs = parseda.default_args(num_queries=300)
model, criterion = train.build_training(margs, device="cuda:0", with_text_encoder=True)
train.to_bf16(model)
batch = train.synthetic_batch(4, 800, 1333, device="cuda:0")
batch[0].tensors = batch[0].tensors.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
opt = train.FusedMasterAdamW(model)
model.train()
samples, text, targets = batch
marks = {}
def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e
def run(record):
    t = [("start", ev())]
    # phase A pieces, re-implemented inline to place the marks
    features, pos = model.backbone(samples); t.append(("backbone", ev()))
    srcs, masks = [], []
    for l, feat in enumerate(features):
        src, mask = feat.decompose(); srcs.append(model.input_proj[l](src)); masks.append(mask)
    src = model.input_proj[3](features[-1].tensors)
    import torch.nn.functional as F
    mask = F.interpolate(samples.mask[None].float(), size=src.shape[-2:]).to(torch.bool)[0]
    pos.append(model.backbone[1](NestedTensor(src, mask)).to(src.dtype)); srcs.append(src); masks.append(mask)
    t.append(("input_proj+pos", ev()))
    tr = model.transformer
    ids, am = text["input_ids"], text["attention_mask"]
    pooled = tr.text_encoder(input_ids=ids, attention_mask=am).pooler_output; t.append(("text_encoder", ev()))
    mc = tr(srcs=srcs, masks=masks, pos_embeds=pos, query_embed=model._query_embeds(), text=text, encode_and_save=True)
    t.append(("transformer phase A (incl. 2nd text enc)", ev()))
    out = model(samples, encode_and_save=False, memory_cache=mc, text=text, targets=targets); t.append(("phase B decoders+heads", ev()))
    out = {k: ([{kk: vv.float() for kk, vv in a.items()} for a in v] if k == "aux_outputs" else v.float()) for k, v in out.items()}
    loss = criterion.weighted_sum(criterion(out, targets)); t.append(("criterion", ev()))
    opt.zero_grad(); (loss + 0 * pooled.float().sum()).backward(); t.append(("backward", ev()))
    opt.step(0.1); t.append(("clip+optimizer", ev()))
    torch.cuda.synchronize()
    if record:
        for (n0, e0), (n1, e1) in zip(t[:-1], t[1:]):
            marks.setdefault(n1, []).append(e0.elapsed_time(e1))
for i in range(6):
    run(i >= 2)

# kernel launches and GPU-busy time per phase: one profiler session per phase of one more eager step
from torch.profiler import profile, ProfilerActivity
counts = {}
class Phase:
    def __init__(self, name): self.name = name
    def __enter__(self):
        torch.cuda.synchronize(); self.p = profile(activities=[ProfilerActivity.CUDA]); self.p.__enter__(); return self
    def __exit__(self, *a):
        torch.cuda.synchronize(); self.p.__exit__(*a)
        ev_ = [e for e in self.p.events() if str(e.device_type).endswith("CUDA")]
        counts[self.name] = (len(ev_), sum((e.device_time if hasattr(e, "device_time") else e.cuda_time) for e in ev_) / 1e3)
import torch.nn.functional as F
with Phase("backbone"):
    features, pos = model.backbone(samples)
with Phase("input_proj+pos"):
    srcs, masks = [], []
    for l, feat in enumerate(features):
        src, mask = feat.decompose(); srcs.append(model.input_proj[l](src)); masks.append(mask)
    src = model.input_proj[3](features[-1].tensors)
    mask = F.interpolate(samples.mask[None].float(), size=src.shape[-2:]).to(torch.bool)[0]
    pos.append(model.backbone[1](NestedTensor(src, mask)).to(src.dtype)); srcs.append(src); masks.append(mask)
tr = model.transformer
with Phase("text_encoder"):
    pooled = tr.text_encoder(input_ids=text["input_ids"], attention_mask=text["attention_mask"]).pooler_output
with Phase("transformer phase A (incl. 2nd text enc)"):
    mc = tr(srcs=srcs, masks=masks, pos_embeds=pos, query_embed=model._query_embeds(), text=text, encode_and_save=True)
with Phase("phase B decoders+heads"):
    out = model(samples, encode_and_save=False, memory_cache=mc, text=text, targets=targets)
with Phase("criterion"):
    loss = criterion.weighted_sum(criterion(out, targets))
opt.zero_grad()
with Phase("backward"):
    (loss + 0 * pooled.float().sum()).backward()
with Phase("clip+optimizer"):
    opt.step(0.1)
tot = 0
for k, v in marks.items():
    m = sum(v) / len(v); tot += m
    c = counts.get(k, (0, 0.0))
    print(f"{k:45s} {m:8.2f} ms eager wall   {c[0]:5d} launches  {c[1]:7.2f} ms GPU busy")
print(f"{'sum':45s} {tot:8.2f} ms")
