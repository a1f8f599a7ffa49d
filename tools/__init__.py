"""Measurement and evidence tools (never imported by the product package).

Tools that A/B another build of the C-ABI library (`make -C rlipv2_amd/csrc ablation | timeline`, the emulated builds of
tools/emu) select it with RLIPV2_LIB_PATH / RLIPV2_CPU_LIB_PATH.  Those variables are read HERE, by the tools package, and handed
to the product through its explicit entry point `rlipv2_amd._lib.use_library(path)`: the product package itself reads no library
location from the environment.  A tool run as a script gets this by `import tools` (any `from tools.x import y` does it)."""
import os as _os


def apply_library_overrides():
    from rlipv2_amd import _lib
    path = _os.environ.get("RLIPV2_LIB_PATH")
    if path:
        _lib.use_library(path)
    path = _os.environ.get("RLIPV2_CPU_LIB_PATH")
    if path:
        _lib.use_cpu_library(path)


apply_library_overrides()
