"""Fused optimiser step of the bf16 train step: gradient-norm clipping (max-norm 0.1, reference
engine.py:170-172) + AdamW with the reference's three parameter groups (main.py:523-541) on float32 master
weights, writing the bf16 parameters back -- two HIP launches (csrc/fused_adamw.hip, C ABI
include/rlipv2_optim.h) instead of ~90 PyTorch launches."""
from __future__ import annotations

import ctypes
import math

import numpy as np
import torch

from . import _lib, roofline


def _same_layout(a, b):
    """same element order in memory (strides of size-1 dimensions are arbitrary and ignored)"""
    return a.shape == b.shape and all(sa == sb for n, sa, sb in zip(a.shape, a.stride(), b.stride()) if n > 1)


class _Group(ctypes.Structure):
    _fields_ = [(n, ctypes.c_float) for n in ("beta1", "beta2", "one_minus_beta1", "one_minus_beta2", "eps", "decay",
                                              "step_size", "bias_correction2_sqrt")]


class FusedMasterAdamW:
    """AdamW over float32 master copies of bf16 parameters; same update rule as torch.optim.AdamW and the
    same clipping as torch.nn.utils.clip_grad_norm_ (tests/test_optim_gpu.py compares them step by step)."""

    def __init__(self, model, lr=1.41e-4, lr_backbone=1.41e-5, text_encoder_lr=1.41e-5, weight_decay=1e-4,
                 betas=(0.9, 0.999), eps=1e-8):
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        if not named:
            raise ValueError("no trainable parameters")
        for n, p in named:
            if not p.is_cuda:
                raise RuntimeError("Not implemented on the CPU")
            if p.dtype != torch.bfloat16:
                raise RuntimeError(f"FusedMasterAdamW expects bfloat16 parameters ({n} is {p.dtype})")
        self.names = [n for n, _ in named]
        self.params = [p for _, p in named]
        self.master = [p.detach().float().clone() for p in self.params]
        self.exp_avg = [torch.zeros_like(m) for m in self.master]
        self.exp_avg_sq = [torch.zeros_like(m) for m in self.master]
        group_of = [1 if "backbone" in n else 2 if "text_encoder" in n else 0 for n in self.names]
        lrs = [lr, lr_backbone, text_encoder_lr]
        used = sorted(set(group_of))
        self.group_index = [used.index(g) for g in group_of]
        self.param_groups = [{"lr": lrs[g], "betas": tuple(betas), "eps": eps, "weight_decay": weight_decay,
                              "params": [p for p, gi in zip(self.params, group_of) if gi == g]} for g in used]
        self.t = 0
        self.steps = [0] * len(self.params)     # per-parameter step count, as torch.optim keeps it
        self.device = self.params[0].device
        self.sqnorm = torch.zeros(1, dtype=torch.float32, device=self.device)
        self._last_grad_scale = 1.0
        sizes = [ctypes.c_int() for _ in range(4)]
        _lib.lib().adamw_abi_sizes(*[ctypes.byref(s) for s in sizes])
        assert (sizes[0].value, sizes[1].value, sizes[2].value) == (56, 8, ctypes.sizeof(_Group)), "ABI mismatch"
        self.chunk = sizes[3].value
        self._tables = {}        # tuple of gradient pointers -> (tensor table, chunk table, n_chunks)
        self._relayout = {}
        self._checked = [None] * len(self.params)    # per parameter: the gradient OBJECT whose dtype / layout passed last

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            p.grad = None

    def _grads(self):
        """(indices, gradients) of the parameters that have one.  The graphed step hands over the SAME gradient tensors every
        step (static buffers): a gradient object that passed the dtype / layout check once is not checked again -- the
        per-parameter shape / stride comparison was ~1 ms of Python per step for ~750 parameters, between the end of the
        backward graph and the optimiser's two launches."""
        idx, grads = [], []
        checked = self._checked
        for i, p in enumerate(self.params):
            g = p.grad
            if g is None:
                continue                                   # torch.optim skips parameters without gradient
            if g is checked[i]:
                idx.append(i)
                grads.append(g)
                continue
            if g.dtype != torch.bfloat16 or not _same_layout(g, p):
                # e.g. a convolution weight kept channels-last whose gradient arrives contiguous: re-lay it
                # out into a persistent buffer (stable address -> the pointer table stays cached)
                buf = self._relayout.get(i)
                if buf is None:
                    buf = self._relayout[i] = torch.empty_like(p)
                g = buf.copy_(g)
                self.grad_copies = getattr(self, "grad_copies", 0) + 1
            else:
                checked[i] = g
            idx.append(i)
            grads.append(g)
        return idx, grads

    def _table(self, idx, grads, eff_group):
        # (parameter addresses are part of the key: modules may re-lay their parameters out, deform_attn.adjacent_cat)
        key = (tuple(idx), tuple(g.data_ptr() for g in grads), tuple(eff_group), tuple(self.params[i].data_ptr() for i in idx))
        hit = self._tables.get(key)
        if hit is not None:
            return hit
        rows = np.empty((len(idx), 7), dtype=np.int64)
        chunks = []
        for r, (i, g) in enumerate(zip(idx, grads)):
            n = self.params[i].numel()
            rows[r] = (g.data_ptr(), self.master[i].data_ptr(), self.exp_avg[i].data_ptr(),
                       self.exp_avg_sq[i].data_ptr(), self.params[i].data_ptr(), n, eff_group[r])
            chunks += [(r, c) for c in range((n + self.chunk - 1) // self.chunk)]
        t_dev = torch.from_numpy(rows).to(self.device)
        c_dev = torch.from_numpy(np.asarray(chunks, dtype=np.int32).reshape(-1, 2)).to(self.device)
        self.table_builds = getattr(self, "table_builds", 0) + 1
        if len(self._tables) > 8:
            self._tables.clear()
        self._tables[key] = (t_dev, c_dev, len(chunks))
        return self._tables[key]

    @torch.no_grad()
    def step(self, max_norm=0.1, grad_scale=1.0):
        """`grad_scale`: factor on every gradient, applied inside the kernels (norm and update): the data-parallel step
        hands over the all-reduced SUM with grad_scale = 1 / world instead of scaling its 425 MB buffer first."""
        idx, grads = self._grads()
        if not idx:
            return
        self._last_grad_scale = float(grad_scale)
        self.t += 1
        # bias corrections depend on how many updates a parameter has had; a parameter that skipped a step
        # (no gradient) lags behind, so the kernel's "group" is (parameter group, update count)
        for i in idx:
            self.steps[i] += 1
        pairs = sorted({(self.group_index[i], self.steps[i]) for i in idx})
        if len(pairs) > 8:
            raise RuntimeError("FusedMasterAdamW: more than 8 distinct (group, step-count) combinations")
        eff_group = [pairs.index((self.group_index[i], self.steps[i])) for i in idx]
        t_dev, c_dev, n_chunks = self._table(idx, grads, eff_group)
        groups = (_Group * len(pairs))()
        for gi, (pg, t) in enumerate(pairs):
            g = self.param_groups[pg]
            b1, b2 = g["betas"]
            groups[gi] = _Group(b1, b2, 1.0 - b1, 1.0 - b2, g["eps"], 1.0 - g["lr"] * g["weight_decay"],
                                g["lr"] / (1.0 - b1 ** t), math.sqrt(1.0 - b2 ** t))
        L = _lib.lib()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        clip = max_norm is not None and max_norm > 0
        if clip:
            st = L.adamw_grad_sqnorm_bf16(t_dev.data_ptr(), c_dev.data_ptr(), n_chunks, self.sqnorm.data_ptr(), stream)
            if st:
                raise RuntimeError("adamw_grad_sqnorm: " + _lib.strerror(st))
        st = L.adamw_step_scaled_bf16(t_dev.data_ptr(), c_dev.data_ptr(), n_chunks, self.sqnorm.data_ptr(),
                                      float(max_norm) if clip else 0.0, float(grad_scale), groups, len(pairs), stream)
        if st:
            raise RuntimeError("adamw_step: " + _lib.strerror(st))
        # gradient (bf16) read twice (norm + update), master / two moments read + written (f32), bf16 parameter written
        n = sum(self.params[i].numel() for i in idx)
        roofline.add(n * (2 * (2 if clip else 1) + 3 * 8 + 2))

    def grad_norm(self):
        """norm of the last step's gradients as the update saw them, i.e. with that step's `grad_scale` applied -- the
        value the reference logs from clip_grad_norm_ (engine.py:170); a device tensor, no sync."""
        return self.sqnorm.sqrt() * self._last_grad_scale

    def state_dict(self):
        return {"t": self.t, "steps": list(self.steps), "names": list(self.names), "master": self.master, "exp_avg": self.exp_avg,
                "exp_avg_sq": self.exp_avg_sq, "param_groups": [{k: v for k, v in g.items() if k != "params"}
                                                                for g in self.param_groups]}

    def load_state_dict(self, state):
        if list(state["names"]) != self.names:
            raise ValueError("optimizer state belongs to a different parameter set")
        self.t = int(state["t"])
        self.steps = list(state.get("steps", [self.t] * len(self.params)))
        with torch.no_grad():
            for dst, src in ((self.master, state["master"]), (self.exp_avg, state["exp_avg"]),
                             (self.exp_avg_sq, state["exp_avg_sq"])):
                for d, s in zip(dst, src):
                    d.copy_(s)
            for p, m in zip(self.params, self.master):
                p.copy_(m)
        for g, s in zip(self.param_groups, state["param_groups"]):
            g.update(s)
