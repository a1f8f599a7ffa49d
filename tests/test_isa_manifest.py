"""Every kernel's device code against tests/golden/isa_manifest.json: the hash of its instruction stream (tools/isa_audit.py:
kernarg offsets and label numbers normalised) and which hardware run that exact stream has behind it -- round 2's GPU suite
under the driver, a builder-run GPU session of round 3, or none.  The GPU pool was closed for most of rounds 3-4, so what
"validated on hardware" means is a property of the BYTES the compiler produces, not of the source history: any change of a
kernel file that alters a validated kernel's code fails here, on the CPU, and has to be acknowledged by regenerating the
manifest (`python tools/isa_audit.py --manifest tests/golden/isa_manifest.json r02_driver_gputest_green=dc4be55
r03_builder_gpu_runs=294ccf0`), which then records the kernel as never run."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HIPCC = "/opt/rocm/bin/hipcc"

pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc (cross-compiles gfx950 without a GPU)")


def test_device_code_equals_the_manifest():
    from tools import isa_audit
    with open(os.path.join(ROOT, "tests", "golden", "isa_manifest.json")) as f:
        want = json.load(f)["kernels"]
    cur = isa_audit.hashes_of("WORKTREE")
    names = isa_audit.demangle(sorted({k for _, k in cur}))
    got = {f"{f}::{names[k]}": h for (f, k), (n, h) in cur.items()}
    assert set(got) == set(want), (sorted(set(got) - set(want))[:5], sorted(set(want) - set(got))[:5])
    changed = [k for k in got if got[k] != want[k]["hash"]]
    assert not changed, f"device code changed (hardware status lost): {changed[:5]}"
    # what the product's default train step runs must have run on hardware, with two known exceptions
    never = sorted(k for k, v in want.items() if v["hardware"] == "never_run")
    # (cell_forward_kernel: opt-in experiment; step_scaled_kernel: the optimiser's data-parallel form, grad_scale != 1 -- the
    #  one-GPU step runs round 2's step_kernel; layernorm_wide.hip / window_attention.hip / msda_rows.hip (round 5): behind norm.fused_wide_layer_norm /
    #  swin.fused_window_attention / deform_attn.sample_then_project, OFF until
    #  routes.validate has compared the Swin step with it against the plain ops on a GPU -- the R50 configurations never reach it;
    #  cell_records_backward_kernel (round 5): behind msda.records_route, OFF)
    assert all("cell_forward_kernel" in k or "cell_records_backward_kernel" in k or "records_unbin_kernel" in k or "step_scaled_kernel" in k or k.startswith(("layernorm_wide.hip::", "window_attention.hip::", "msda_rows.hip::"))
               for k in never), never
