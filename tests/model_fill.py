"""Closed-form, storage-free weights shared by the golden generator and the tests: every float
parameter / buffer element is a * sin(b * i + c_k), with (a, b, c_k) derived from the tensor's
NAME and fan-in (same rule as tests/golden/ref_harness.fill_closed_form)."""
import zlib

import torch


def fill_closed_form(module, scale=1.0):
    with torch.no_grad():
        for name, t in list(module.named_parameters()) + list(module.named_buffers()):
            if not t.is_floating_point():
                continue
            h = zlib.crc32(name.encode()) & 0xffffffff
            c = (h % 10007) / 10007.0 * 6.283185307179586
            b = 0.37 + (h % 97) / 97.0
            i = torch.arange(t.numel(), dtype=torch.float64)
            if t.dim() >= 2:
                a = scale * (3.0 / t[0].numel()) ** 0.5
                v = a * torch.sin(b * i + c)
            elif name.endswith("weight") and ("norm" in name.lower() or "LayerNorm" in name):
                v = 1.0 + 0.1 * torch.sin(b * i + c)
            elif "gamma" in name:
                v = 0.25 + 0.05 * torch.sin(b * i + c)
            else:
                v = 0.05 * scale * torch.sin(b * i + c)
            t.copy_(v.reshape(t.shape).to(t.dtype))
