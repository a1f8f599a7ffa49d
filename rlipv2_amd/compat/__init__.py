"""Import-path compatibility with the reference tree.

``rlipv2_amd.compat.install()`` registers

* ``MultiScaleDeformableAttention`` -- the reference's native extension module
  (models/ops/src/vision.cpp:13-16), bound to librlipv2_msda.so;
* ``models.ops.functions`` / ``models.ops.modules`` and their twins under ``models.dab_deformable.ops`` --
  the two import paths the reference's transformers use (ParSetransformer.py:27, deformable_transformer.py:24,
  dab_deformable/deformable_transformer.py:28),

in ``sys.modules`` so that the reference's own ``models/ops/test.py`` and model files resolve the op and the
module to this package without edits.  Nothing is registered when a real module of that name is already present.
"""
from __future__ import annotations

import sys
import types

from . import MultiScaleDeformableAttention as _ext


def install(force: bool = False) -> None:
    from .. import deform_attn, msda

    def put(name, mod):
        if force or name not in sys.modules:
            sys.modules[name] = mod

    put("MultiScaleDeformableAttention", _ext)
    for root in ("models.ops", "models.dab_deformable.ops"):
        fn_pkg = types.ModuleType(root + ".functions")
        fn_pkg.MSDeformAttnFunction = msda.MSDeformAttnFunction
        fn_mod = types.ModuleType(root + ".functions.ms_deform_attn_func")
        fn_mod.MSDeformAttnFunction = msda.MSDeformAttnFunction
        fn_mod.MSDA = _ext
        mod_pkg = types.ModuleType(root + ".modules")
        mod_pkg.MSDeformAttn = deform_attn.MSDeformAttn
        mod_mod = types.ModuleType(root + ".modules.ms_deform_attn")
        mod_mod.MSDeformAttn = deform_attn.MSDeformAttn
        pkg = types.ModuleType(root)
        pkg.functions, pkg.modules = fn_pkg, mod_pkg
        parts = root.split(".")
        for k in range(1, len(parts)):                       # parent packages ("models", "models.dab_deformable")
            put(".".join(parts[:k]), types.ModuleType(".".join(parts[:k])))
        put(root, pkg)
        put(root + ".functions", fn_pkg)
        put(root + ".functions.ms_deform_attn_func", fn_mod)
        put(root + ".modules", mod_pkg)
        put(root + ".modules.ms_deform_attn", mod_mod)
