"""Import harness for the reference checkout (build container only; nothing here ships to the GPU box).

It makes the reference's model code importable in this container so that golden vectors can be
generated from it (tests/golden/make_model_golden.py).  The reference expects torch 1.10 /
torchvision / timm / transformers 4.5.1 / a CUDA extension; the shims below stand in for what
is absent (SURVEY.md section 8c lists them and why each is needed):

  1. `transformers` is imported before any fake torchvision is visible;
  2. stub packages `torchvision` (version string, resnet names, RoIAlign ...) and
     `timm.models.layers.{DropPath,to_2tuple,trunc_normal_}`;
  3. `transformers.modeling_utils.{apply_chunking_to_forward, prune_linear_layer,
     find_pruneable_heads_and_indices}` re-exported; `RobertaConfig.from_pretrained` ->
     roberta-base constants; tokenizer / RobertaModel.from_pretrained -> small stubs (no weights
     exist offline): models are driven through the pre-encoded text tuple;
  4. `RobertaLayer.get_extended_attention_mask` patched to the transformers-4.5.1 formula
     `(1 - mask[:, None, None, :]) * -10000.0`;
  5. a stub `MultiScaleDeformableAttention` module and `MSDeformAttnFunction` replaced, in both
     copies of the op package, by a shim whose `.apply` calls the reference's own
     `ms_deform_attn_core_pytorch` (differentiable) -- BASELINE config 1's pure-PyTorch path;
  6. `models/__init__.py` is NOT executed (it imports every model family and the datasets'
     dependencies): a namespace stand-in for the `models` package is registered instead;
  7. the argparse namespace comes from `main.get_args_parser`, extracted with `ast` (importing
     `main` would import the datasets);
  8. a stand-in backbone object (strides, num_channels, [1] = position encoder).
"""
import ast
import os
import sys
import types

import torch
from torch import nn

REF = "/root/reference"


def available():
    return os.path.isdir(os.path.join(REF, "models"))


_installed = False


def install():
    """Idempotent.  After this, `import models.hoi` etc. work."""
    global _installed
    if _installed:
        return
    if not available():
        raise RuntimeError("reference checkout not mounted at /root/reference")
    import transformers  # noqa: F401  (1)
    import transformers.modeling_utils as mu
    import transformers.pytorch_utils as pu
    from transformers import RobertaConfig

    # (3)
    for name in ("apply_chunking_to_forward", "prune_linear_layer"):
        if not hasattr(mu, name):
            setattr(mu, name, getattr(pu, name))
    if not hasattr(mu, "find_pruneable_heads_and_indices"):
        mu.find_pruneable_heads_and_indices = lambda *a, **k: (set(), torch.zeros(0, dtype=torch.long))

    def _roberta_base_config(cls, *a, **k):
        return RobertaConfig(vocab_size=50265, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                             intermediate_size=3072, hidden_act="gelu", hidden_dropout_prob=0.1,
                             attention_probs_dropout_prob=0.1, max_position_embeddings=514, type_vocab_size=1,
                             layer_norm_eps=1e-5, pad_token_id=1, bos_token_id=0, eos_token_id=2)
    RobertaConfig.from_pretrained = classmethod(_roberta_base_config)

    class _StubTextEncoder(nn.Module):            # stands in for RobertaModel (weights unavailable offline)
        def __init__(self):
            super().__init__()
            self.config = _roberta_base_config(None)
            self.dummy = nn.Parameter(torch.zeros(1))

    class _StubTokenizer:
        pass

    import transformers as tf
    tf.RobertaModel.from_pretrained = classmethod(lambda cls, *a, **k: _StubTextEncoder())
    tf.RobertaTokenizerFast.from_pretrained = classmethod(lambda cls, *a, **k: _StubTokenizer())

    # (2) torchvision / timm stubs
    tv = types.ModuleType("torchvision")
    tv.__version__ = "0.15.0"
    tv.__path__ = []
    tv_models = types.ModuleType("torchvision.models")
    tv_models.__path__ = []
    tv_models_utils = types.ModuleType("torchvision.models._utils")
    tv_models_utils.IntermediateLayerGetter = type("IntermediateLayerGetter", (nn.Module,), {})
    tv_models_resnet = types.ModuleType("torchvision.models.resnet")
    tv_models_resnet.ResNet = type("ResNet", (nn.Module,), {})
    tv_models_resnet.Bottleneck = type("Bottleneck", (nn.Module,), {})
    tv_models.resnet50 = lambda *a, **k: None
    tv_ops = types.ModuleType("torchvision.ops")
    tv_ops.__path__ = []
    tv_ops.RoIAlign = type("RoIAlign", (nn.Module,), {})
    tv_ops.DeformConv2d = type("DeformConv2d", (nn.Module,), {})
    tv_ops.deform_conv2d = lambda *a, **k: None
    tv_ops_boxes = types.ModuleType("torchvision.ops.boxes")

    def box_area(boxes):
        return (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    tv_ops_boxes.box_area = box_area
    tv_ops_misc = types.ModuleType("torchvision.ops.misc")
    tv.models, tv.ops = tv_models, tv_ops
    tv_models._utils, tv_models.resnet = tv_models_utils, tv_models_resnet
    tv_ops.boxes, tv_ops.misc = tv_ops_boxes, tv_ops_misc
    for name, mod in (("torchvision", tv), ("torchvision.models", tv_models),
                      ("torchvision.models._utils", tv_models_utils), ("torchvision.models.resnet", tv_models_resnet),
                      ("torchvision.ops", tv_ops), ("torchvision.ops.boxes", tv_ops_boxes),
                      ("torchvision.ops.misc", tv_ops_misc)):
        sys.modules[name] = mod

    timm = types.ModuleType("timm"); timm.__path__ = []
    timm_models = types.ModuleType("timm.models"); timm_models.__path__ = []
    timm_layers = types.ModuleType("timm.models.layers")

    class DropPath(nn.Module):
        def __init__(self, p=0.0):
            super().__init__()
            self.p = p

        def forward(self, x):
            assert self.p == 0.0 or not self.training
            return x
    timm_layers.DropPath = DropPath
    timm_layers.to_2tuple = lambda x: (x, x) if not isinstance(x, tuple) else x
    timm_layers.trunc_normal_ = nn.init.trunc_normal_
    timm.models, timm_models.layers = timm_models, timm_layers
    for name, mod in (("timm", timm), ("timm.models", timm_models), ("timm.models.layers", timm_layers)):
        sys.modules[name] = mod

    # (5) the op extension
    sys.modules["MultiScaleDeformableAttention"] = types.ModuleType("MultiScaleDeformableAttention")

    # (6) namespace stand-ins so that package __init__ files are not executed
    if REF not in sys.path:
        sys.path.insert(0, REF)
    for pkg, rel in (("models", "models"), ("models.dab_deformable", "models/dab_deformable"),
                     ("models.swin", "models/swin")):
        m = types.ModuleType(pkg)
        m.__path__ = [os.path.join(REF, rel)]
        sys.modules[pkg] = m

    import models.dab_deformable.ops.functions.ms_deform_attn_func as f2
    import models.ops.functions.ms_deform_attn_func as f1

    def _make_shim(core):
        class _CoreShim:
            @staticmethod
            def apply(value, shapes, starts, loc, aw, im2col_step):
                return core(value, shapes, loc, aw)
        return _CoreShim
    import models.dab_deformable.ops.modules.ms_deform_attn as m2
    import models.ops.modules.ms_deform_attn as m1
    m1.MSDeformAttnFunction = _make_shim(f1.ms_deform_attn_core_pytorch)
    m2.MSDeformAttnFunction = _make_shim(f2.ms_deform_attn_core_pytorch)

    # (4)
    import models.modeling_roberta as mr

    def _ext_mask_4_5_1(self, attention_mask, input_shape, device=None, dtype=None):
        m = attention_mask[:, None, None, :].to(torch.float32)
        return (1.0 - m) * -10000.0
    mr.RobertaLayer.get_extended_attention_mask = _ext_mask_4_5_1
    _installed = True


def reference_args(**overrides):
    """argparse namespace of main.get_args_parser() (7), with the script flags of
    scripts/RLIP_ParSeDA/train_RLIP_ParSeDA_v2_mixed_vgcoco_resnet.sh applied, then `overrides`."""
    import argparse
    src = open(os.path.join(REF, "main.py")).read()
    tree = ast.parse(src)
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "get_args_parser")
    mod = ast.Module(body=[fn], type_ignores=[])
    ns = {"argparse": argparse}
    exec(compile(mod, "main_get_args_parser", "exec"), ns)
    args = ns["get_args_parser"]().parse_args([])
    script = dict(enc_layers=6, dec_layers=3, num_queries=200, dim_feedforward=2048, dropout=0.0,
                  num_feature_levels=4, with_box_refine=True, subject_class=True, use_no_obj_token=True,
                  fusion_type="GLIP_attn", gating_mechanism="VXAc", verb_query_tgt_type="vanilla_MBF",
                  fusion_interval=2, fusion_last_vis=True, lang_aux_loss=True, giou_verb_label=True,
                  pseudo_verb=True, RLIP_ParSeDA_v2=True, use_dab=True)
    for k, v in {**script, **overrides}.items():
        setattr(args, k, v)
    return args


class StandInBackbone(nn.Module):
    """(8) what RLIP_ParSeDA needs of a backbone: strides, num_channels, [1] = position encoder,
    and __call__(NestedTensor) -> ([NestedTensor x3], [pos x3]).  Features are supplied."""

    def __init__(self, num_channels=(512, 1024, 2048)):
        super().__init__()
        from models.position_encoding import PositionEmbeddingSine
        self.strides = [8, 16, 32]
        self.num_channels = list(num_channels)
        self.pos = PositionEmbeddingSine(128, normalize=True)
        self.features = None        # list of (tensor [N,C,H,W], mask [N,H,W])

    def __getitem__(self, i):
        assert i == 1
        return self.pos

    def forward(self, samples):
        from util.misc import NestedTensor
        out = [NestedTensor(t, m) for t, m in self.features]
        pos = [self.pos(x).to(x.tensors.dtype) for x in out]
        return out, pos


def fill_closed_form(module, scale=1.0):
    """Deterministic, storage-free weights: every parameter / buffer element is
    a * sin(b * i + c_k) with (a, b, c_k) derived from the tensor's NAME and fan-in, so that both
    sides (reference and rlipv2_amd) can fill identically named tensors identically."""
    import zlib
    with torch.no_grad():
        for name, t in list(module.named_parameters()) + list(module.named_buffers()):
            if not t.is_floating_point():
                continue
            h = zlib.crc32(name.encode()) & 0xffffffff
            c = (h % 10007) / 10007.0 * 6.283185307179586
            b = 0.37 + (h % 97) / 97.0
            n = t.numel()
            i = torch.arange(n, dtype=torch.float64)
            if t.dim() >= 2:
                fan_in = t[0].numel()
                a = scale * (3.0 / fan_in) ** 0.5
                v = a * torch.sin(b * i + c)
            elif name.endswith("weight") and ("norm" in name.lower() or "LayerNorm" in name):
                v = 1.0 + 0.1 * torch.sin(b * i + c)
            elif "gamma" in name:
                v = 0.25 + 0.05 * torch.sin(b * i + c)
            else:
                v = 0.05 * scale * torch.sin(b * i + c)
            t.copy_(v.reshape(t.shape).to(t.dtype))
