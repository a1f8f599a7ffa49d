"""Static check of the device assembly (tools/isa_hazard_check.py): no instruction touches the destination of an LDS read that
no `s_waitcnt lgkmcnt` has retired.  The kernels read LDS through inline assembly in places, where the wait in front of the
first use is written by hand; the host model (tools/emu/) cannot see that class of error, this can.  Unit tests of the checker's
rules on synthetic assembly, then the real thing: hipcc -S of the kernel files (cross-compiles without a GPU)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_hazard_check as H  # noqa: E402

HIPCC = "/opt/rocm/bin/hipcc"


def hazards(text):
    return H.check("k", text.strip("\n").split("\n"))


def test_read_consumed_without_wait_is_reported_and_with_wait_is_not():
    bad = """
\tds_read_b64_tr_b16 v[10:11], v4
\tv_mfma_f32_16x16x32_bf16 v[0:3], v[10:13], v[20:23], v[0:3]
\ts_endpgm
"""
    good = bad.replace("\tv_mfma", "\ts_waitcnt lgkmcnt(0)\n\tv_mfma")
    assert len(hazards(bad)) == 1 and not hazards(good)


def test_counted_wait_retires_in_issue_order_only():
    t = """
\tds_read_b128 v[0:3], v9
\tds_read_b128 v[4:7], v9 offset:16
\ts_waitcnt lgkmcnt(1)
\tv_add_f32_e32 v20, v0, v1
\tv_add_f32_e32 v21, v4, v5
\ts_endpgm
"""
    found = hazards(t)
    assert len(found) == 1 and list(found.values())[0][1] == [("v", 4), ("v", 5)]      # the second read is still in flight
    # LDS stores count too: with a store issued after the reads, "at most one outstanding" is the store -- both reads are
    # back; "at most two" may still include the second read
    t2 = t.replace("\ts_waitcnt lgkmcnt(1)", "\tds_write_b32 v9, v30\n\ts_waitcnt lgkmcnt(1)")
    assert not hazards(t2)
    t3 = t.replace("\ts_waitcnt lgkmcnt(1)", "\tds_write_b32 v9, v30\n\ts_waitcnt lgkmcnt(2)")
    assert len(hazards(t3)) == 1 and list(hazards(t3).values())[0][1] == [("v", 4), ("v", 5)]


def test_scalar_load_in_flight_makes_a_counted_wait_retire_nothing():
    t = """
\tds_read_b32 v0, v9
\ts_load_dword s4, s[0:1], 0x0
\ts_waitcnt lgkmcnt(1)
\tv_mov_b32_e32 v1, v0
\ts_endpgm
"""
    assert len(hazards(t)) == 1                     # (scalar loads return out of order: the one outstanding may be the read)
    assert not hazards(t.replace("lgkmcnt(1)", "lgkmcnt(0)"))


def test_loop_carried_read_is_seen_through_the_back_edge():
    t = """
\ts_waitcnt lgkmcnt(0)
.LBB0_1:
\tv_add_f32_e32 v5, v0, v5
\tds_read_b32 v0, v9
\ts_cbranch_scc1 .LBB0_1
\ts_endpgm
"""
    assert len(hazards(t)) == 1
    assert not hazards(t.replace(".LBB0_1:\n", ".LBB0_1:\n\ts_waitcnt lgkmcnt(0)\n"))


# every kernel file with hand-written LDS reads or waits, and the big LDS users; ":ablation" = with the experiment arms
FILES = ["token_gemm", "expand_gemm", "msda_patch", "msda_patch:ablation", "msda_window", "alif_attention", "msda_quad", "msda_dest",
         "window_attention", "msda_rows", "msda_sparse"]        # (round 5 / the decoders' pass: compiler-scheduled LDS reads only)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
@pytest.mark.parametrize("name", FILES)
def test_kernel_files_have_no_unretired_lds_read(name, tmp_path):
    stem, _, flavour = name.partition(":")
    out = str(tmp_path / (stem + ".s"))
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "--cuda-device-only", "-S",
           "-I" + os.path.join(ROOT, "rlipv2_amd", "csrc"), "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "rlipv2_amd", "csrc", stem + ".hip"), "-o", out]
    if flavour == "ablation":
        cmd.insert(1, "-DMSDA_ABLATION")
    subprocess.run(cmd, check=True, capture_output=True, timeout=900)
    kernels = H.kernels(out)
    assert kernels
    reads = 0
    for kname, body in kernels:
        assert not H.check(kname, body), kname
        reads += sum(1 for l in body if H.RET.match(l.strip()))
    assert reads > 0                                                # (the files named here all read LDS)
