"""rlipv2_amd -- MI355X-native hot path of RLIPv2-ParSeDA.

Only what the path needs lives here: ``csrc/`` (hand-written HIP kernels for gfx950 behind the
C ABI of ``include/rlipv2_msda.h``) and the host-side mirror of the reference's operator /
module interface.  There is no fallback for GPU tensors: every op on them raises if the HIP
library is missing.  CPU tensors are a different device, served by the op's CPU twins
(``include/rlipv2_msda_cpu.h``, the reference's own device dispatch) -- never a substitute for
the GPU path.
"""
import os as _os


def use_tuned_miopen_db() -> str | None:
    """Points MIOpen (the library behind the 3x3 / 7x7 / strided convolutions of the R50 trunk) at the Find results
    recorded on an MI355X by `tools/tune_miopen.sh` (`rlipv2_amd/tuned/miopen/`, two small text files keyed by MIOpen
    version and device).  Without them PyTorch's immediate mode picks split-K solvers with float32 workspace
    zero / cast passes for most of these shapes: 41.8 against 39.8 ms per train step on the same box.  Lookup only
    -- nothing is searched at run time unless the caller turns on `torch.backends.cudnn.benchmark`.  The files are
    copied to a scratch directory (MIOpen writes into its user-db directory; the tree stays as committed).
    A MIOPEN_USER_DB_PATH set by the caller wins; RLIPV2_TUNED_MIOPEN=0 disables.  Must run before the first
    convolution of the process (import time)."""
    if _os.environ.get("RLIPV2_TUNED_MIOPEN", "1") == "0" or _os.environ.get("MIOPEN_USER_DB_PATH"):
        return None
    import shutil
    import tempfile
    src = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "tuned", "miopen")
    if not _os.path.isdir(src):
        return None
    import hashlib
    names = sorted(n for n in _os.listdir(src) if n.endswith(".txt"))
    digest = hashlib.md5()
    for name in names:
        with open(_os.path.join(src, name), "rb") as f:
            digest.update(f.read())
    # a per-user cache directory (not a guessable path under a shared /tmp that somebody else could have created):
    # $XDG_CACHE_HOME or ~/.cache, falling back to the temporary directory; in every case the directory must belong to
    # this user and be writable by nobody else, otherwise the recorded results are not used
    base = _os.environ.get("XDG_CACHE_HOME") or _os.path.join(_os.path.expanduser("~"), ".cache")
    if not _os.path.isdir(_os.path.dirname(base)) or not _os.access(_os.path.dirname(base), _os.W_OK):
        base = tempfile.gettempdir()
    dst = _os.path.join(base, "rlipv2_amd", "rlipv2_miopen_db_%d_%s" % (_os.getuid(), digest.hexdigest()[:10]))
    try:
        _os.makedirs(dst, mode=0o700, exist_ok=True)
        st = _os.stat(dst)
        if st.st_uid != _os.getuid() or (st.st_mode & 0o022):
            return None
        for name in names:
            target = _os.path.join(dst, name)
            if not _os.path.exists(target):
                tmp = "%s.%d.tmp" % (target, _os.getpid())
                shutil.copyfile(_os.path.join(src, name), tmp)
                _os.replace(tmp, target)                    # atomic: the other ranks of the node race for this
    except OSError:
        return None
    _os.environ["MIOPEN_USER_DB_PATH"] = dst
    return dst


def miopen_db_status() -> dict:
    """What can be known without a profiler about whether MIOpen is really using the recorded Find results: the db files
    are keyed by device and MIOpen version in their NAMES (gfx950100.HIP.3_5_0_<build>.ufdb.txt) and MIOpen silently
    ignores files of another version -- then its immediate mode picks the split-K solvers again (about 2 ms per step).
    Returns {"path", "recorded_for", "library", "version_match"}; version_match False means the db is NOT in effect."""
    path = _os.environ.get("MIOPEN_USER_DB_PATH", "")
    out = {"path": path or None, "recorded_for": None, "library": None, "version_match": None}
    try:
        import torch
        v = int(torch.backends.cudnn.version() or 0)             # MIOpen: major * 1e6 + minor * 1e3 + patch
        out["library"] = "%d_%d_%d" % (v // 1000000, (v // 1000) % 1000, v % 1000)
    except Exception:                                             # noqa: BLE001 -- reporting only
        return out
    if path and _os.path.isdir(path):
        names = [n for n in _os.listdir(path) if n.endswith(".ufdb.txt")]
        if names:
            parts = names[0].split(".")
            rec = parts[2] if len(parts) > 2 else ""
            out["recorded_for"] = "_".join(rec.split("_")[:3])
            out["version_match"] = out["recorded_for"] == out["library"]
    return out


use_tuned_miopen_db()

from .msda import (  # noqa: E402,F401
    MSDeformAttnFunction,
    ms_deform_attn_backward,
    ms_deform_attn_forward,
)

__all__ = ["MSDeformAttnFunction", "ms_deform_attn_forward", "ms_deform_attn_backward"]
