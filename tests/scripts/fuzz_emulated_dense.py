"""Random-shape fuzz of the dense MFMA / normalisation kernels on the lane-level workgroup model (tools/emu/): the test bodies of
tests/test_dense_emulated.py called with shapes drawn at random -- token counts around every boundary the kernels have (fewer than
one 32-token step, one step, ragged last row block, several chunks + tail rows), every supported width.
TEST INFRASTRUCTURE (uses the float32 PyTorch formulas of the tests as the checker).
usage: python tests/scripts/fuzz_emulated_dense.py [seed] [seconds]      (EMU_SANITIZE=1 + LD_PRELOAD of the ASan runtime: sanitizer build)
"""
import ctypes
import os
import subprocess
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import test_dense_emulated as D  # noqa: E402


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    budget = float(sys.argv[2]) if len(sys.argv) > 2 else 600.0
    so = os.path.join(tempfile.mkdtemp(prefix="emu_dense_fuzz"), "libdense_emu.so")
    subprocess.run([os.path.join(ROOT, "tools", "emu", "build_dense_lib.sh"), so], check=True, capture_output=True, timeout=900)
    lib = ctypes.CDLL(so)
    rng = np.random.default_rng(seed)
    t_end = time.time() + budget
    edge_t = [1, 2, 31, 32, 33, 63, 64, 65, 255, 256, 257, 287, 288, 511, 513]
    n = fails = 0
    while time.time() < t_end:
        kind = int(rng.integers(0, 3))
        T = int(rng.choice(edge_t)) if rng.random() < 0.5 else int(rng.integers(1, 1400))
        try:
            if kind == 0:
                M, K, f32 = int(rng.choice([128, 256, 384])), int(rng.choice([128, 256, 384])), bool(rng.integers(0, 2))
                desc = f"wgrad T={T} M={M} K={K} f32={f32}"
                D.test_mfma_weight_gradient_against_torch.__wrapped__(lib, T, M, K, f32) if hasattr(
                    D.test_mfma_weight_gradient_against_torch, "__wrapped__") else D.test_mfma_weight_gradient_against_torch(lib, T, M, K, f32)
            elif kind == 1:
                N = 64 * int(rng.integers(1, 6))
                mask, bias, relu = bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
                desc = f"expand T={T} N={N} mask={mask} bias={bias} relu={relu}"
                D.test_mfma_expand_gemm_against_torch(lib, T, N, mask, bias, relu)
            else:
                with_b = bool(rng.integers(0, 2))
                desc = f"add_layernorm rows={T} with_b={with_b}"
                D.test_add_layernorm_forward_and_backward_against_torch(lib, T, with_b)
            print("ok  ", f"#{n}", desc, flush=True)
        except Exception as e:  # noqa: BLE001
            fails += 1
            print("FAIL", f"#{n}", desc, repr(e)[:300], flush=True)
        n += 1
    print(f"seed {seed}: {n} problems, {fails} failures", flush=True)
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
