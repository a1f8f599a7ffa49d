"""oracle/msda_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Python face of the CPU oracle for multi-scale deformable attention.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this module; the product package ``rlipv2_amd`` never does.

Two independent restatements live here:

* ``forward`` / ``backward``      -> ctypes calls into ``msda_oracle.c`` (the C
  restatement; see that file's header for the reference file:line map).
* ``forward_numpy`` / ``backward_numpy`` -> a vectorised numpy restatement of the
  same algorithm (reference: models/ops/functions/ms_deform_attn_func.py:45-65
  for the math, models/ops/src/cuda/ms_deform_im2col_cuda.cuh:33-159 for the
  bounds rules and gradient formulas).  It exists so the C code is checked by
  something that shares no code with it.

Pinning: both are checked against tests/golden/msda_*.npz, vectors generated in
the build container by importing the reference's ``ms_deform_attn_core_pytorch``
and differentiating it with autograd (tests/golden/make_msda_golden.py).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}


def build(force: bool = False) -> None:
    """Compile the C restatement (gcc, no GPU involved)."""
    targets = [os.path.join(_HERE, n) for n in ("libmsda_oracle.so", "libmsda_oracle_omp.so")]
    src = os.path.join(_HERE, "msda_oracle.c")
    stale = force or any(
        (not os.path.exists(t)) or os.path.getmtime(t) < os.path.getmtime(src) for t in targets
    )
    if stale:
        subprocess.run(["make", "-s", "-C", _HERE, "-B", "all"], check=True)


def _lib(omp: bool = False):
    key = "omp" if omp else "serial"
    if key not in _LIBS:
        path = os.path.join(_HERE, "libmsda_oracle_omp.so" if omp else "libmsda_oracle.so")
        if not os.path.exists(path):
            build()
        lib = ctypes.CDLL(path)
        lib.msda_oracle_threads.restype = ctypes.c_int
        _LIBS[key] = lib
    return _LIBS[key]


def threads(omp: bool = True) -> int:
    return int(_lib(omp).msda_oracle_threads())


def _suffix(dtype) -> str:
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return "f32"
    if dtype == np.float64:
        return "f64"
    raise TypeError(f"oracle supports float32/float64, got {dtype}")


def _prep(value, shapes, starts, loc, aw):
    value = np.ascontiguousarray(value)
    dt = value.dtype
    loc = np.ascontiguousarray(loc, dtype=dt)
    aw = np.ascontiguousarray(aw, dtype=dt)
    shapes = np.ascontiguousarray(shapes, dtype=np.int64)
    starts = np.ascontiguousarray(starts, dtype=np.int64)
    N, S, M, D = value.shape
    _, Lq, M2, L, P, two = loc.shape
    assert M2 == M and two == 2 and shapes.shape == (L, 2) and starts.shape == (L,)
    assert aw.shape == (N, Lq, M, L, P)
    assert int((shapes[:, 0] * shapes[:, 1]).sum()) == S
    return value, shapes, starts, loc, aw, (N, S, M, D, L, Lq, P)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def forward(value, shapes, starts, loc, aw, omp: bool = False):
    """out[N, Lq, M*D] via the C restatement."""
    value, shapes, starts, loc, aw, dims = _prep(value, shapes, starts, loc, aw)
    N, S, M, D, L, Lq, P = dims
    out = np.empty((N, Lq, M * D), dtype=value.dtype)
    fn = getattr(_lib(omp), "msda_oracle_forward_" + _suffix(value.dtype))
    fn.restype = None
    fn(_p(value), _p(shapes), _p(starts), _p(loc), _p(aw),
       *(ctypes.c_int(x) for x in dims), _p(out))
    return out


def backward(value, shapes, starts, loc, aw, grad_out, omp: bool = False):
    """(grad_value, grad_loc, grad_aw) via the C restatement."""
    value, shapes, starts, loc, aw, dims = _prep(value, shapes, starts, loc, aw)
    N, S, M, D, L, Lq, P = dims
    grad_out = np.ascontiguousarray(grad_out, dtype=value.dtype).reshape(N, Lq, M * D)
    g_value = np.empty_like(value)
    g_loc = np.empty_like(loc)
    g_aw = np.empty_like(aw)
    fn = getattr(_lib(omp), "msda_oracle_backward_" + _suffix(value.dtype))
    fn.restype = None
    fn(_p(value), _p(shapes), _p(starts), _p(loc), _p(aw), _p(grad_out),
       *(ctypes.c_int(x) for x in dims), _p(g_value), _p(g_loc), _p(g_aw))
    return g_value, g_loc, g_aw


# --------------------------------------------------------------------------------------
# numpy restatement (independent of the C code)
# --------------------------------------------------------------------------------------
def _corners(value, shapes, starts, loc):
    """Per level: corner indices, bilinear pieces and validity for every (n,q,m,p)."""
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = loc.shape
    dt = value.dtype
    for l in range(L):
        H, W = int(shapes[l, 0]), int(shapes[l, 1])
        x = loc[:, :, :, l, :, 0] * dt.type(W) - dt.type(0.5)   # w_im  [N,Lq,M,P]
        y = loc[:, :, :, l, :, 1] * dt.type(H) - dt.type(0.5)   # h_im
        with np.errstate(invalid="ignore"):
            inside = (y > -1) & (x > -1) & (y < H) & (x < W)      # .cuh:285 (NaN -> False)
        xs = np.where(inside, x, 0)
        ys = np.where(inside, y, 0)
        x0 = np.floor(xs).astype(np.int64)
        y0 = np.floor(ys).astype(np.int64)
        x1, y1 = x0 + 1, y0 + 1
        lw, lh = xs - x0, ys - y0
        hw, hh = 1 - lw, 1 - lh
        ok = [
            inside & (y0 >= 0) & (x0 >= 0),
            inside & (y0 >= 0) & (x1 <= W - 1),
            inside & (y1 <= H - 1) & (x0 >= 0),
            inside & (y1 <= H - 1) & (x1 <= W - 1),
        ]
        idx = [
            starts[l] + np.clip(y0, 0, H - 1) * W + np.clip(x0, 0, W - 1),
            starts[l] + np.clip(y0, 0, H - 1) * W + np.clip(x1, 0, W - 1),
            starts[l] + np.clip(y1, 0, H - 1) * W + np.clip(x0, 0, W - 1),
            starts[l] + np.clip(y1, 0, H - 1) * W + np.clip(x1, 0, W - 1),
        ]
        yield l, H, W, inside, (hh, hw, lh, lw), ok, idx


def _gather(value, idx, ok):
    """value[n, idx[n,q,m,p], m, :] with zeros where not ok -> [N,Lq,M,P,D]."""
    N, S, M, D = value.shape
    n_i = np.arange(N)[:, None, None, None]
    m_i = np.arange(M)[None, None, :, None]
    v = value[n_i, idx, m_i]                      # [N,Lq,M,P,D]
    return np.where(ok[..., None], v, 0)


def forward_numpy(value, shapes, starts, loc, aw):
    value, shapes, starts, loc, aw, dims = _prep(value, shapes, starts, loc, aw)
    N, S, M, D, L, Lq, P = dims
    out = np.zeros((N, Lq, M, D), dtype=value.dtype)
    for l, H, W, inside, (hh, hw, lh, lw), ok, idx in _corners(value, shapes, starts, loc):
        w = [hh * hw, hh * lw, lh * hw, lh * lw]
        val = 0
        for k in range(4):
            val = val + w[k][..., None] * _gather(value, idx[k], ok[k])
        out += (val * aw[:, :, :, l, :, None]).sum(axis=3)
    return out.reshape(N, Lq, M * D)


def backward_numpy(value, shapes, starts, loc, aw, grad_out):
    value, shapes, starts, loc, aw, dims = _prep(value, shapes, starts, loc, aw)
    N, S, M, D, L, Lq, P = dims
    dt = value.dtype
    go = np.ascontiguousarray(grad_out, dtype=dt).reshape(N, Lq, M, 1, D)
    g_value = np.zeros_like(value)
    g_loc = np.zeros_like(loc)
    g_aw = np.zeros_like(aw)
    n_i = np.broadcast_to(np.arange(N)[:, None, None, None], (N, Lq, M, P))
    m_i = np.broadcast_to(np.arange(M)[None, None, :, None], (N, Lq, M, P))
    for l, H, W, inside, (hh, hw, lh, lw), ok, idx in _corners(value, shapes, starts, loc):
        w = [hh * hw, hh * lw, lh * hw, lh * lw]
        v = [_gather(value, idx[k], ok[k]) for k in range(4)]
        a = aw[:, :, :, l, :]
        val = sum(w[k][..., None] * v[k] for k in range(4))
        g_aw[:, :, :, l, :] = (go * val).sum(-1)
        gw = hh[..., None] * (v[1] - v[0]) + lh[..., None] * (v[3] - v[2])
        gh = hw[..., None] * (v[2] - v[0]) + lw[..., None] * (v[3] - v[1])
        g_loc[:, :, :, l, :, 0] = dt.type(W) * a * (go * gw).sum(-1)
        g_loc[:, :, :, l, :, 1] = dt.type(H) * a * (go * gh).sum(-1)
        for k in range(4):
            contrib = np.where(ok[k][..., None], (w[k] * a)[..., None] * go, 0)
            np.add.at(g_value, (n_i, idx[k], m_i), contrib)
    return g_value, g_loc, g_aw


# --------------------------------------------------------------------------------------
# PyTorch-CPU restatement: per-level grid_sample, the formulation of the reference's own CPU
# path (models/ops/functions/ms_deform_attn_func.py:45-65).  Differentiable through autograd;
# BASELINE.json asks for this path's time on the host cores as the CPU baseline.  It keeps a
# zero-weight corner's location gradient at the exclusion boundary (tests/conftest.py), so
# it is the timed baseline, the C port above stays the parity checker.
# --------------------------------------------------------------------------------------
def forward_torch(value, shapes, loc, aw):
    """value [N,S,M,D], shapes [(H,W)...], loc [N,Lq,M,L,P,2] in [0,1], aw [N,Lq,M,L,P] (torch CPU tensors)
    -> [N, Lq, M*D].  One bilinear, zero-padded, align_corners=False grid_sample per level."""
    import torch
    import torch.nn.functional as F
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = loc.shape
    sizes = [int(h) * int(w) for h, w in shapes]
    grid_all = loc * 2.0 - 1.0                                   # grid_sample's [-1, 1] convention
    per_level = []
    for lvl, (chunk, (H, W)) in enumerate(zip(value.split(sizes, dim=1), shapes)):
        H, W = int(H), int(W)
        img = chunk.permute(0, 2, 3, 1).reshape(N * M, D, H, W)                      # one image per (n, head)
        grid = grid_all[:, :, :, lvl].permute(0, 2, 1, 3, 4).reshape(N * M, Lq, P, 2)
        per_level.append(F.grid_sample(img, grid, mode="bilinear", padding_mode="zeros", align_corners=False))
    sampled = torch.stack(per_level, dim=3).reshape(N * M, D, Lq, L * P)            # [N*M, D, Lq, L*P]
    weights = aw.permute(0, 2, 1, 3, 4).reshape(N * M, 1, Lq, L * P)
    out = (sampled * weights).sum(-1).reshape(N, M * D, Lq)
    return out.transpose(1, 2).contiguous()
