"""Step-level roofline accounting: algorithmic bytes and FLOPs of ONE eager train step.

    T_mem  = sum over ops of (inputs + outputs, each tensor once per op, in the dtype it has) / HBM peak
    T_mfma = sum of matrix FLOPs (GEMMs, convolutions, attention products; forward and backward) / dense bf16 MFMA peak

ATen ops are counted by a TorchDispatchMode (views move no bytes) and torch's FlopCounterMode; the library's own
kernels (bound through ctypes, invisible to the dispatcher) report their operands through `add()` at their call sites.
The probe runs one extra eager step after the timed region of bench.py -- it is an accounting of the algorithm, not a
measurement of traffic (rocprofv3 PMC passes under profiles/ measure traffic).
"""
from __future__ import annotations

import torch
from torch.utils._python_dispatch import TorchDispatchMode
from torch.utils.flop_counter import FlopCounterMode

HBM_PEAK_BPS = 8.0e12          # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_BF16_PEAK = 2.5e15        # dense bf16 MFMA peak

active = False
_extra_bytes = 0
_extra_flops = 0


def add(nbytes: int, flops: int = 0) -> None:
    """Called by the ctypes-bound kernels' wrappers (msda, linear, norm, optim) while a probe is running."""
    global _extra_bytes, _extra_flops
    if active:
        _extra_bytes += int(nbytes)
        _extra_flops += int(flops)


def tensor_bytes(*tensors) -> int:
    return sum(t.numel() * t.element_size() for t in tensors if torch.is_tensor(t))


class _ByteCounter(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.bytes = 0
        self.ops = 0

    @staticmethod
    def _is_view(func) -> bool:
        rets = getattr(func, "_schema", None)
        if rets is None:
            return False
        rets = func._schema.returns
        return bool(rets) and all(r.alias_info is not None and not r.alias_info.is_write for r in rets)

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        if not self._is_view(func):
            seen, total = set(), 0

            def visit(x):
                nonlocal total
                if torch.is_tensor(x) and x.is_cuda:
                    key = (x.data_ptr(), x.numel(), x.element_size())
                    if key not in seen:
                        seen.add(key)
                        # an expanded (stride-0) operand is read once from its storage
                        total += min(x.numel() * x.element_size(), x.untyped_storage().nbytes())
                elif isinstance(x, (list, tuple)):
                    for y in x:
                        visit(y)

            visit(args)
            visit(list((kwargs or {}).values()))
            visit(out)
            self.bytes += total
            self.ops += 1
        return out


def probe(step_fn):
    """Runs step_fn() once under the accounting modes; returns the step's algorithmic bytes / matrix FLOPs and the two
    roofline times in seconds."""
    global active, _extra_bytes, _extra_flops
    _extra_bytes = _extra_flops = 0
    active = True
    try:
        with FlopCounterMode(display=False) as fc, _ByteCounter() as bc:
            step_fn()
        torch.cuda.synchronize()
    finally:
        active = False
    nbytes = bc.bytes + _extra_bytes
    flops = fc.get_total_flops() + _extra_flops
    return {"bytes": nbytes, "flops": flops, "aten_ops": bc.ops, "library_bytes": _extra_bytes,
            "library_flops": _extra_flops, "T_mem_s": nbytes / HBM_PEAK_BPS, "T_mfma_s": flops / MFMA_BF16_PEAK}
