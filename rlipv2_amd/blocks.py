"""Small host-side building blocks of the RLIPv2-ParSeDA path (PyTorch-ROCm plumbing).

Each block keeps the parameter names of its reference counterpart so reference checkpoints load:
  MLP                         models/dab_deformable/deformable_transformer.py:1763-1775
  sine_embed_for_position     gen_sineembed_for_position, same file :1777-1803
  MultiBranchFusion           same file :1025-1068
  FeatureResizer              models/ParSetransformer.py:1909-1928
  inverse_sigmoid             util/misc.py:460-464
  PositionEmbeddingSine       models/position_encoding.py:22-58
  NestedTensor / nested_tensor_from_tensor_list   util/misc.py:299-341
"""
from __future__ import annotations

import math
from typing import List, Optional

import torch
import torch.nn.functional as F
from torch import Tensor, nn

from .linear import token_linear


def inverse_sigmoid(x: Tensor, eps: float = 1e-5) -> Tensor:
    x = x.clamp(min=0, max=1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


class MLP(nn.Module):
    """Linear-ReLU stack; `layers.N` parameter names as in the reference."""

    def __init__(self, input_dim: int, hidden_dim: int, output_dim: int, num_layers: int):
        super().__init__()
        dims = [input_dim] + [hidden_dim] * (num_layers - 1) + [output_dim]
        self.num_layers = num_layers
        self.layers = nn.ModuleList(nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:]))

    def forward(self, x: Tensor) -> Tensor:
        for layer in self.layers[:-1]:
            x = token_linear(x, layer.weight, layer.bias, relu=True)
        return self.layers[-1](x)


_SINE_DIMS = {}


def _sine_dim_t(device, n=128, temperature=10000.0):
    key = (str(device), n, temperature)
    if key not in _SINE_DIMS:
        i = torch.arange(n, dtype=torch.float32, device=device)
        _SINE_DIMS[key] = temperature ** (2 * (i // 2) / n)
    return _SINE_DIMS[key]


def sine_embed_for_position(pos: Tensor) -> Tensor:
    """[N, nq, 2|4] normalised (x, y[, w, h]) -> [N, nq, 256|512]: 128 sin/cos features per
    coordinate, interleaved (sin on even, cos on odd frequencies), ordered (y, x, w, h)."""
    dim_t = _sine_dim_t(pos.device)
    ang = (pos * (2 * math.pi))[..., None] / dim_t                       # [N, nq, C, 128]
    emb = torch.stack((ang[..., 0::2].sin(), ang[..., 1::2].cos()), dim=-1).flatten(-2)
    order = [1, 0] if pos.shape[-1] == 2 else [1, 0, 2, 3]
    if pos.shape[-1] not in (2, 4):
        raise ValueError("Unknown pos_tensor shape(-1):{}".format(pos.shape[-1]))
    parts = emb.unbind(-2)                      # (one node: a select per coordinate is a zero-fill + copy + add each in the backward)
    return torch.cat([parts[k] for k in order], dim=-1)


class MultiBranchFusion(nn.Module):
    """relu(sum_k fc_3[k](relu(fc_1[k](a) * fc_2[k](b)))) over `cardinality` branches."""

    def __init__(self, appearance_size: int, spatial_size: int, representation_size: int, cardinality: int):
        super().__init__()
        self.cardinality = cardinality
        sub = representation_size // cardinality
        assert sub * cardinality == representation_size, \
            "The given representation size should be divisible by cardinality"
        self.fc_1 = nn.ModuleList(nn.Linear(appearance_size, sub) for _ in range(cardinality))
        self.fc_2 = nn.ModuleList(nn.Linear(spatial_size, sub) for _ in range(cardinality))
        self.fc_3 = nn.ModuleList(nn.Linear(sub, representation_size) for _ in range(cardinality))

    def forward(self, appearance: Tensor, spatial: Tensor) -> Tensor:
        # the 16 branches are three grouped GEMMs: concatenated fc_1 / fc_2, then a block-diagonal fc_3
        w1 = torch.cat([m.weight for m in self.fc_1], 0)
        b1 = torch.cat([m.bias for m in self.fc_1], 0)
        w2 = torch.cat([m.weight for m in self.fc_2], 0)
        b2 = torch.cat([m.bias for m in self.fc_2], 0)
        h = F.relu(F.linear(appearance, w1, b1) * F.linear(spatial, w2, b2))           # [..., card*sub]
        w3 = torch.cat([m.weight for m in self.fc_3], 1)                                 # [rep, card*sub]
        b3 = torch.stack([m.bias for m in self.fc_3], 0).sum(0)
        return F.relu(F.linear(h, w3, b3))


class FeatureResizer(nn.Module):
    """Linear + LayerNorm(eps 1e-12) + dropout."""

    def __init__(self, input_feat_size: int, output_feat_size: int, dropout: float, do_ln: bool = True):
        super().__init__()
        self.do_ln = do_ln
        self.fc = nn.Linear(input_feat_size, output_feat_size, bias=True)
        self.layer_norm = nn.LayerNorm(output_feat_size, eps=1e-12)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x: Tensor) -> Tensor:
        x = self.fc(x)
        if self.do_ln:
            x = self.layer_norm(x)
        return self.dropout(x)


class NestedTensor:
    """`no_padding` is a host-side hint (set by whoever built the batch) that the mask is all False: the
    model then skips the value masking inside MSDeformAttn (identical results, no device round trip)."""

    def __init__(self, tensors: Tensor, mask: Optional[Tensor], no_padding: bool = False):
        self.tensors = tensors
        self.mask = mask
        self.no_padding = no_padding

    def to(self, device):
        return NestedTensor(self.tensors.to(device), None if self.mask is None else self.mask.to(device),
                            self.no_padding)

    def decompose(self):
        return self.tensors, self.mask

    def __repr__(self):
        return str(self.tensors)


def nested_tensor_from_tensor_list(tensor_list: List[Tensor]) -> NestedTensor:
    """Pad [C, H, W] images to the batch maximum; mask is True on padding."""
    if tensor_list[0].ndim != 3:
        raise ValueError("not supported")
    c = tensor_list[0].shape[0]
    h = max(t.shape[1] for t in tensor_list)
    w = max(t.shape[2] for t in tensor_list)
    t0 = tensor_list[0]
    batch = torch.zeros((len(tensor_list), c, h, w), dtype=t0.dtype, device=t0.device)
    mask = torch.ones((len(tensor_list), h, w), dtype=torch.bool, device=t0.device)
    for img, pad, m in zip(tensor_list, batch, mask):
        pad[:, : img.shape[1], : img.shape[2]].copy_(img)
        m[: img.shape[1], : img.shape[2]] = False
    return NestedTensor(batch, mask, no_padding=all(t.shape[1] == h and t.shape[2] == w for t in tensor_list))


cache_padding_free = True     # (tools/r04_host_ab.py flips the attribute for its A/B)


class PositionEmbeddingSine(nn.Module):
    """Image sine position encoding (normalised to 2*pi, temperature 10000), [N, 2*F, H, W]."""

    def __init__(self, num_pos_feats: int = 64, temperature: float = 10000, normalize: bool = False, scale=None):
        super().__init__()
        if scale is not None and not normalize:
            raise ValueError("normalize should be True if scale is passed")
        self.num_pos_feats = num_pos_feats
        self.temperature = temperature
        self.normalize = normalize
        self.scale = 2 * math.pi if scale is None else scale

    def forward(self, tensor_list: NestedTensor, out_dtype=None) -> Tensor:
        """`out_dtype`: the dtype the caller converts to anyway (the feature maps').  For a batch WITHOUT padding
        (`NestedTensor.no_padding`, a host-side hint) the encoding is a function of the shape alone: it is computed once per
        (shape, dtype, device) -- through the code below, so the values are the ones an uncached call returns -- and kept in
        channels-last memory, so that the transformer's `flatten(2).transpose(1, 2)` is a view of contiguous tokens instead of
        a strided read.  Per train step that was ~60 launches and ~0.5 GB of float32 traffic for constants."""
        mask = tensor_list.mask
        assert mask is not None
        if getattr(tensor_list, "no_padding", False) and cache_padding_free:
            key = (tuple(mask.shape), str(mask.device), out_dtype, self.num_pos_feats, float(self.temperature),
                   self.normalize, float(self.scale))
            cache = self.__dict__.setdefault("_no_padding_cache", {})
            hit = cache.get(key)
            if hit is None:
                capturing = mask.is_cuda and torch.cuda.is_current_stream_capturing()
                with torch.no_grad():
                    hit = self._encode(mask)
                    if out_dtype is not None:
                        hit = hit.to(out_dtype)
                    hit = hit.contiguous(memory_format=torch.channels_last)
                # Entries are never evicted: a captured HIP graph that read one keeps reading its memory on every replay.
                # A full table (a run with very many padding-free shapes) stops caching new shapes instead; memory of a
                # capture's private pool is not kept beyond this call either.
                if capturing or len(cache) >= 16:
                    return hit
                cache[key] = hit
            return hit
        out = self._encode(mask)
        return out if out_dtype is None else out.to(out_dtype)

    def _encode(self, mask):
        not_mask = ~mask
        y_embed = not_mask.cumsum(1, dtype=torch.float32)
        x_embed = not_mask.cumsum(2, dtype=torch.float32)
        if self.normalize:
            eps = 1e-6
            y_embed = y_embed / (y_embed[:, -1:, :] + eps) * self.scale
            x_embed = x_embed / (x_embed[:, :, -1:] + eps) * self.scale
        dim_t = _sine_dim_t(mask.device, self.num_pos_feats, float(self.temperature))
        px = x_embed[..., None] / dim_t
        py = y_embed[..., None] / dim_t
        px = torch.stack((px[..., 0::2].sin(), px[..., 1::2].cos()), dim=-1).flatten(-2)
        py = torch.stack((py[..., 0::2].sin(), py[..., 1::2].cos()), dim=-1).flatten(-2)
        return torch.cat((py, px), dim=-1).permute(0, 3, 1, 2)
