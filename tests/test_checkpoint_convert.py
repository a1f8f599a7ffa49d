"""DAB-Deformable-DETR -> RLIPv2-ParSeDA checkpoint key conversion (rlipv2_amd.checkpoint.convert_dab_ddetr) against the
behaviour of the reference's convert_parameters/convert_parameters_DABDDETR.py (:48-170), on a closed-form-filled state
dict (the reference script itself needs a CUDA device for its fresh rows and writes to disk: it cannot run here)."""
import torch

from rlipv2_amd import checkpoint as C


def _fill(name, shape):
    g = torch.Generator().manual_seed(sum(map(ord, name)) % 9973)
    return torch.rand(*shape, generator=g) + (hash(name) % 7)


def _dab_ddetr_state(box_refine=False):
    sd = {}
    for i in range(6):
        for j, (o, c) in enumerate(((256, 256), (256, 256), (4, 256))):
            sd[f"bbox_embed.{i}.layers.{j}.weight"] = _fill(f"bb{i}{j}w", (o, c))
            sd[f"bbox_embed.{i}.layers.{j}.bias"] = _fill(f"bb{i}{j}b", (o,))
            if box_refine:
                sd[f"transformer.decoder.bbox_embed.{i}.layers.{j}.weight"] = _fill(f"tbb{i}{j}w", (o, c))
                sd[f"transformer.decoder.bbox_embed.{i}.layers.{j}.bias"] = _fill(f"tbb{i}{j}b", (o,))
        sd[f"class_embed.{i}.weight"] = _fill(f"ce{i}w", (91, 256))
        sd[f"class_embed.{i}.bias"] = _fill(f"ce{i}b", (91,))
    for i in range(2):
        sd[f"transformer.encoder.layers.{i}.linear1.weight"] = _fill(f"enc{i}", (8, 4))
        sd[f"transformer.decoder.layers.{i}.cross_attn.value_proj.weight"] = _fill(f"dec{i}", (4, 4))
    sd["transformer.decoder.ref_point_head.layers.0.weight"] = _fill("rph", (4, 8))
    sd["tgt_embed.weight"] = _fill("tgt", (300, 256))
    sd["refpoint_embed.weight"] = _fill("rp", (300, 4))
    sd["backbone.0.body.conv1.weight"] = _fill("conv1", (4, 3, 7, 7))
    return sd


def test_parse_conversion_of_a_dab_ddetr_checkpoint():
    sd = _dab_ddetr_state(box_refine=True)
    before = {k: v.clone() for k, v in sd.items()}
    out = C.convert_dab_ddetr({"model": sd, "epoch": 49}, with_box_refine=True, generator=torch.Generator().manual_seed(1))
    m = out["model"]
    assert out["epoch"] == 49
    assert all(torch.equal(sd[k], before[k]) for k in before) and set(sd) == set(before)       # input untouched
    # encoder / decoder duplication (originals stay)
    for i in range(2):
        e = f"layers.{i}.linear1.weight"
        assert torch.equal(m["transformer.ho_encoder." + e], before["transformer.encoder." + e])
        assert "transformer.encoder." + e in m
        d = f"layers.{i}.cross_attn.value_proj.weight"
        for dec in ("ho_decoder", "verb_decoder"):
            assert torch.equal(m[f"transformer.{dec}." + d], before["transformer.decoder." + d])
    assert torch.equal(m["transformer.verb_decoder.ref_point_head.layers.0.weight"], before["transformer.decoder.ref_point_head.layers.0.weight"])
    # box heads: outer copies from bbox_embed, the decoders' inner copies from transformer.decoder.bbox_embed
    for i in range(6):
        for j in range(3):
            for part in ("weight", "bias"):
                src = before[f"bbox_embed.{i}.layers.{j}.{part}"]
                inner = before[f"transformer.decoder.bbox_embed.{i}.layers.{j}.{part}"]
                for head in ("sub_bbox_embed", "obj_bbox_embed"):
                    assert torch.equal(m[f"{head}.{i}.layers.{j}.{part}"], src)
                    for dec in ("ho_decoder", "verb_decoder"):
                        assert torch.equal(m[f"transformer.{dec}.{head}.{i}.layers.{j}.{part}"], inner)
        # class head: the 80 COCO rows by category id + one fresh row
        w, b = m[f"obj_class_embed.{i}.weight"], m[f"obj_class_embed.{i}.bias"]
        assert w.shape == (81, 256) and b.shape == (81,)
        assert torch.equal(w[:80], before[f"class_embed.{i}.weight"][list(C.COCO_OBJECT_IDS)])
        assert torch.equal(b[:80], before[f"class_embed.{i}.bias"][list(C.COCO_OBJECT_IDS)])
        assert float(w[80].abs().max()) <= 1.0 / 16 + 1e-6                                       # Linear(256, 1) init range
    # ONE fresh "no pair" row shared by the six layers (the reference draws background_class once, :60-63)
    assert all(torch.equal(m[f"obj_class_embed.{i}.weight"][80], m["obj_class_embed.0.weight"][80]) for i in range(6))
    assert all(torch.equal(m[f"obj_class_embed.{i}.bias"][80], m["obj_class_embed.0.bias"][80]) for i in range(6))
    assert len(C.COCO_OBJECT_IDS) == 80 and 12 not in C.COCO_OBJECT_IDS and 91 not in C.COCO_OBJECT_IDS
    assert torch.equal(m["verb_tgt_embed.weight"], before["tgt_embed.weight"])
    assert torch.equal(m["backbone.0.body.conv1.weight"], before["backbone.0.body.conv1.weight"])


def test_detreg_vcoco_and_dropped_class_heads():
    sd = _dab_ddetr_state(box_refine=False)
    m = C.convert_dab_ddetr({"model": sd}, with_box_refine=True, detreg=True, dataset="vcoco")["model"]
    for dec in ("ho_decoder", "verb_decoder"):                     # DETReg: the inner copies come from bbox_embed too
        assert torch.equal(m[f"transformer.{dec}.sub_bbox_embed.3.layers.2.weight"], sd["bbox_embed.3.layers.2.weight"])
    w = m["obj_class_embed.0.weight"]
    assert w.shape == (82, 256)                                    # V-COCO: one more fresh row in front of the last
    assert torch.equal(w[:80], sd["class_embed.0.weight"][list(C.COCO_OBJECT_IDS)])
    m2 = C.convert_dab_ddetr({"model": sd}, drop_class_embed=True)["model"]
    assert not any(k.startswith("obj_class_embed") for k in m2)
    assert "transformer.ho_decoder.sub_bbox_embed.0.layers.0.weight" not in m2      # no box refinement asked for


def test_mmdetection_checkpoint_only_gets_the_duplication():
    sd = {"bbox_head.transformer.encoder.layers.0.linear1.weight": _fill("a", (4, 4)),
          "bbox_head.transformer.decoder.layers.0.linear1.weight": _fill("b", (4, 4)),
          "backbone.conv1.weight": _fill("c", (4, 3, 3, 3))}
    m = C.convert_dab_ddetr({"state_dict": sd, "meta": 1})["model"]
    assert torch.equal(m["transformer.ho_encoder.layers.0.linear1.weight"], sd["bbox_head.transformer.encoder.layers.0.linear1.weight"])
    assert torch.equal(m["transformer.verb_decoder.layers.0.linear1.weight"], sd["bbox_head.transformer.decoder.layers.0.linear1.weight"])
    assert "verb_tgt_embed.weight" not in m and not any("bbox_head" in k for k in m)


def test_mmdetection_checkpoint_given_as_a_path(tmp_path):
    """the kind of checkpoint is decided on the LOADED dict, not on the path argument"""
    sd = {"bbox_head.transformer.encoder.layers.0.linear1.weight": _fill("a", (4, 4)),
          "bbox_head.transformer.decoder.layers.0.linear1.weight": _fill("b", (4, 4))}
    path = tmp_path / "mmdet.pth"
    torch.save({"state_dict": sd, "meta": 1}, path)
    out = C.convert_dab_ddetr(str(path))
    m = out["model"]
    assert out["meta"] == 1 and not any("bbox_head" in k for k in m)
    assert torch.equal(m["transformer.ho_decoder.layers.0.linear1.weight"], sd["bbox_head.transformer.decoder.layers.0.linear1.weight"])
    torch.save({"model": _dab_ddetr_state(), "epoch": 3}, tmp_path / "dab.pth")
    out = C.convert_dab_ddetr(tmp_path / "dab.pth")
    assert out["epoch"] == 3 and out["model"]["obj_class_embed.5.weight"].shape == (81, 256)


def test_converted_checkpoint_loads_into_the_model_non_strictly():
    """the converted keys are the model's own names: after `load_pretrained` the seeded parameters hold the source values"""
    from rlipv2_amd import parseda, train
    margs = parseda.default_args(num_queries=16, enc_layers=2, dec_layers=2)
    model, _ = train.build_training(margs, device="cpu", with_text_encoder=False)
    own = model.state_dict()
    sd = {}
    for i in range(6):
        for j in range(3):
            for part in ("weight", "bias"):
                k = f"sub_bbox_embed.{min(i, 1)}.layers.{j}.{part}"
                sd[f"bbox_embed.{i}.layers.{j}.{part}"] = _fill(f"x{i}{j}{part}", tuple(own[k].shape)) if k in own else torch.zeros(1)
        sd[f"class_embed.{i}.weight"] = _fill(f"c{i}", (91, 256))
        sd[f"class_embed.{i}.bias"] = _fill(f"cb{i}", (91,))
    sd["tgt_embed.weight"] = _fill("tgt", tuple(own["tgt_embed.weight"].shape))
    conv = C.convert_dab_ddetr({"model": sd}, drop_class_embed=True)
    missing, unexpected = C.load_pretrained(model, conv, num_queries=16)
    assert "verb_tgt_embed.weight" not in missing and "sub_bbox_embed.0.layers.0.weight" not in missing
    assert torch.equal(model.state_dict()["sub_bbox_embed.1.layers.2.bias"], sd["bbox_embed.1.layers.2.bias"])
    assert torch.equal(model.state_dict()["verb_tgt_embed.weight"], sd["tgt_embed.weight"][:model.state_dict()["verb_tgt_embed.weight"].shape[0]])
