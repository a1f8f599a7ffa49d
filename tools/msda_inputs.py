"""Synthetic MSDA inputs of the model's shapes, generated on the GPU.

mode "uniform"   : sampling locations ~ U(0,1) -- the recipe of the reference's test
                   (models/ops/test.py:38), worst locality.
mode "model"     : encoder-like: reference point = pixel centre of the query's own pyramid cell
                   (reference: models/deformable_transformer.py:803-815) + offsets
                   k * dir_m / (W_l, H_l), k = 1..P, dir_m the 8 directions of the module's
                   initialisation (models/ops/modules/ms_deform_attn.py:66-74) + N(0, 1 px) jitter.
mode "init"      : "model" without the jitter: the module exactly as initialised (sampling_offsets.weight = 0, bias = the
                   direction grid, ms_deform_attn.py:66-74) -- what the random-init train step of bench.py produces.
mode "decoder"   : Lq box queries: centre ~ U(0.1,0.9), size ~ U(0.05,0.5), offsets scaled by
                   size / (2P) (the 4-d reference-point branch, ms_deform_attn.py:110-112).
"""
import math

import torch

PYRAMID_800x1333 = [(100, 167), (50, 84), (25, 42), (13, 21)]   # R50 strides 8/16/32/64
PYRAMID_640x640 = [(80, 80), (40, 40), (20, 20), (10, 10)]


def level_tensors(pyramid, device):
    shapes = torch.tensor(pyramid, dtype=torch.long, device=device)
    starts = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    return shapes, starts


def make_inputs(N, pyramid=PYRAMID_800x1333, M=8, D=32, P=4, Lq=None, mode="model", dtype=torch.float32,
                device="cuda:0", seed=0):
    g = torch.Generator(device=device).manual_seed(seed)
    shapes, starts = level_tensors(pyramid, device)
    L = len(pyramid)
    S = int(shapes.prod(1).sum())
    value = (torch.rand(N, S, M, D, device=device, generator=g) * 0.01).to(dtype)      # test.py:37
    norm = torch.stack([shapes[:, 1], shapes[:, 0]], -1).float()                       # (W, H) per level
    thetas = torch.arange(M, dtype=torch.float32, device=device) * (2.0 * math.pi / M)
    dirs = torch.stack([thetas.cos(), thetas.sin()], -1)
    dirs = dirs / dirs.abs().max(-1, keepdim=True)[0]                                  # [M, 2]
    steps = torch.arange(1, P + 1, device=device, dtype=torch.float32)                 # k = 1..P
    if mode == "uniform":
        Lq = S if Lq is None else Lq
        loc = torch.rand(N, Lq, M, L, P, 2, device=device, generator=g)
    elif mode in ("model", "init"):
        Lq = S
        ref = []
        for (H, W) in pyramid:
            ys, xs = torch.meshgrid((torch.arange(H, device=device) + 0.5) / H,
                                    (torch.arange(W, device=device) + 0.5) / W, indexing="ij")
            ref.append(torch.stack([xs.reshape(-1), ys.reshape(-1)], -1))
        ref = torch.cat(ref, 0)                                                        # [S, 2]
        off = dirs[None, None, :, None, None, :] * steps[None, None, None, None, :, None]
        jitter = torch.randn(N, Lq, M, L, P, 2, device=device, generator=g)            # pixels
        off = off + (jitter if mode == "model" else torch.zeros_like(jitter))           # (full [N, Lq, M, L, P, 2] shape)
        loc = ref[None, :, None, None, None, :] + off / norm[None, None, None, :, None, :]
    elif mode == "decoder":
        Lq = 300 if Lq is None else Lq
        c = torch.rand(N, Lq, 2, device=device, generator=g) * 0.8 + 0.1
        wh = torch.rand(N, Lq, 2, device=device, generator=g) * 0.45 + 0.05
        off = dirs[None, None, :, None, None, :] * steps[None, None, None, None, :, None]
        off = off + torch.randn(N, Lq, M, L, P, 2, device=device, generator=g)
        loc = c[:, :, None, None, None, :] + off / P * wh[:, :, None, None, None, :] * 0.5
    else:
        raise ValueError(mode)
    aw = torch.softmax(torch.randn(N, Lq, M, L * P, device=device, generator=g), -1).view(N, Lq, M, L, P)
    grad_out = torch.randn(N, Lq, M * D, device=device, generator=g).to(dtype)
    return dict(value=value, shapes=shapes, starts=starts, loc=loc.contiguous(), aw=aw.contiguous(),
                grad_out=grad_out, dims=(N, S, M, D, L, Lq, P))
