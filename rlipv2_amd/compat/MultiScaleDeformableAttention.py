"""Stand-in for the reference's compiled extension module ``MultiScaleDeformableAttention``
(models/ops/src/vision.cpp:13-16: two functions, `ms_deform_attn_forward` and `ms_deform_attn_backward`,
same argument order and return types), bound to the gfx950 library through the C ABI of
include/rlipv2_msda.h.  The reference's autograd function calls exactly these two names
(models/ops/functions/ms_deform_attn_func.py:22, :29-30, :39-40), so

    import rlipv2_amd.compat; rlipv2_amd.compat.install()

lets `import MultiScaleDeformableAttention as MSDA` in unmodified reference code resolve here.
"""
from ..msda import ms_deform_attn_backward, ms_deform_attn_forward  # noqa: F401

__all__ = ["ms_deform_attn_forward", "ms_deform_attn_backward"]
