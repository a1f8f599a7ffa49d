"""TEST INFRASTRUCTURE: an autograd Function with MSDeformAttnFunction's signature whose forward and
backward are the CPU oracle (oracle/msda_oracle.c).  CPU tests substitute it for the HIP op
(`rlipv2_amd.deform_attn.msda_function`) to check the HOST logic of the modules against
reference-generated goldens without a GPU.  Never imported by the product package."""
import numpy as np
import torch
from torch.autograd import Function

from oracle import msda_oracle as O


class OracleMSDeformAttnFunction(Function):
    @staticmethod
    def forward(ctx, value, shapes, starts, loc, aw, im2col_step):
        dt = np.float64 if value.dtype == torch.float64 else np.float32
        tdt = torch.float64 if value.dtype == torch.float64 else torch.float32       # bf16 is upcast (exactly)
        a = [value.detach().to(tdt).numpy(), shapes.numpy(), starts.numpy(), loc.detach().to(tdt).numpy(),
             aw.detach().to(tdt).numpy()]
        ctx.args = a
        ctx.dtypes = (value.dtype, loc.dtype, aw.dtype)
        return torch.from_numpy(O.forward(*a)).to(value.dtype)

    @staticmethod
    def backward(ctx, grad_out):
        a = ctx.args
        gv, gl, ga = O.backward(*a, grad_out.contiguous().to(torch.from_numpy(a[0]).dtype).numpy())
        dv, dl, da = ctx.dtypes
        return (torch.from_numpy(gv).to(dv), None, None, torch.from_numpy(gl).to(dl), torch.from_numpy(ga).to(da),
                None)
