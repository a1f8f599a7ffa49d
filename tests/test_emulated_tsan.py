"""ThreadSanitizer on the lane-level model (tools/emu/, EMU_TSAN=1): every lane is a host thread, and the only happens-before edges
between lanes are the model's rendezvous (DPP, shuffles, MFMA, ballots), wave fences and __syncthreads().  A pair of LDS / global
accesses of two lanes that no such edge orders -- a missing barrier in a kernel -- is therefore a reported data race, even when the
numbers of a test happen to come out right.  Here: the positive control (a kernel without its barrier IS reported) and one call
through the newest, never-run kernels (the records route).  The wider runs are recorded in profiles/r05_emulated_checks.txt."""
import ctypes  # noqa: F401
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def _runtime():
    if not os.path.exists(CLANG):
        return None
    p = subprocess.run([CLANG, "-print-file-name=libclang_rt.tsan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


pytestmark = pytest.mark.skipif(_runtime() is None, reason="needs the ROCm clang++ with its ThreadSanitizer runtime")


def _under_tsan(args, timeout):
    env = dict(os.environ, LD_PRELOAD=_runtime(), TSAN_OPTIONS="halt_on_error=0:exitcode=0:report_signal_unsafe=0")
    r = subprocess.run([sys.executable, *args], capture_output=True, text=True, env=env, timeout=timeout, cwd=ROOT)
    return r.returncode, r.stdout, r.stderr.count("WARNING: ThreadSanitizer"), r.stderr


def test_a_kernel_without_its_barrier_is_reported(tmp_path):
    so = str(tmp_path / "libcontrol.so")
    subprocess.run([CLANG, "-x", "c++", "-std=c++20", "-O1", "-fPIC", "-pthread", "-DMSDA_EMU", "-I" + os.path.join(ROOT, "tools", "emu", "stub"),
                    "-I" + os.path.join(ROOT, "rlipv2_amd", "csrc"), "-I" + os.path.join(ROOT, "include"), "-g", "-fsanitize=thread",
                    "-shared-libsan", "-shared", os.path.join(ROOT, "tests", "scripts", "tsan_control_kernel.cpp"), "-o", so],
                   check=True, capture_output=True, timeout=300)
    prog = ("import ctypes, sys, numpy as np\nL = ctypes.CDLL(sys.argv[1])\nout = np.zeros(128, dtype=np.int32)\n"
            "L.run(ctypes.c_void_p(out.ctypes.data), int(sys.argv[2]))\nprint('sum', int(out.sum()))\n")
    rc, out, races, err = _under_tsan(["-c", prog, so, "1"], 120)
    assert rc == 0 and "sum 24384" in out and races == 0, err[-600:]
    rc, out, races, err = _under_tsan(["-c", prog, so, "0"], 120)
    assert rc == 0 and races >= 1, "the missing barrier went unreported: the check below would mean nothing"


def test_records_route_has_no_unordered_lane_pair(tmp_path):
    so = str(tmp_path / "libmsda_emu_tsan.so")
    subprocess.run([os.path.join(ROOT, "tools", "emu", "build_lib.sh"), so], check=True, capture_output=True, timeout=900,
                   env=dict(os.environ, EMU_TSAN="1"))
    rc, out, races, err = _under_tsan([os.path.join(ROOT, "tests", "scripts", "tsan_records_driver.py"), so], 600)
    assert rc == 0 and "forward 0" in out and "backward 0" in out and "finite True True" in out, (out, err[-600:])
    assert races == 0, err[-3000:]
