"""rlipv2_amd -- MI355X-native hot path of RLIPv2-ParSeDA.

Only what the path needs lives here: ``csrc/`` (hand-written HIP kernels for gfx950 behind the
C ABI of ``include/rlipv2_msda.h``) and the host-side mirror of the reference's operator /
module interface.  There is no CPU fallback: every op raises if the HIP library is missing or
a tensor is not on the GPU.
"""
from .msda import (  # noqa: F401
    MSDeformAttnFunction,
    ms_deform_attn_backward,
    ms_deform_attn_forward,
)

__all__ = ["MSDeformAttnFunction", "ms_deform_attn_forward", "ms_deform_attn_backward"]
