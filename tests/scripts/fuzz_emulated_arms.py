"""Random encoder problems through tools/emu/backward_emu (msda_patch.hip on the lane-level workgroup model): the experiment
arms of round 3 against the product kernels, bit for bit, B0 signature and fused geometry, with and without the out-of-reach
flag raised.  Minutes of host time; not part of the suite.
usage: build /tmp/backward_emu as tests/test_backward_emulated.py does, then python tests/scripts/fuzz_emulated_arms.py <seed> <seconds>"""
import os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from test_cell_forward_emulated import bf16_bits, bf16_val
rng = np.random.default_rng(int(sys.argv[1]))
t_end = time.time() + float(sys.argv[2])
def run(env, fused):
    e = {k: v for k, v in os.environ.items() if not k.startswith("RLIPV2_")}; e.update(env); e["EMU_FUSED"] = "1" if fused else "0"
    subprocess.run(['/tmp/backward_emu', '/tmp/af_problem.bin', '/tmp/af_out.bin'], check=True, env=e, timeout=900)
    return np.fromfile('/tmp/af_out.bin', dtype=np.uint8)
n = 0
while time.time() < t_end:
    n += 1
    H0, W0 = int(rng.integers(5, 60)), int(rng.integers(5, 60))
    pyr = [(H0, W0)]
    for l in range(3):
        h, w = pyr[-1]; pyr.append((max(1, (h + 1) // 2), max(1, (w + 1) // 2)))
    pyr = np.asarray(pyr, dtype=np.int64)
    starts = np.concatenate(([0], np.cumsum(pyr[:, 0] * pyr[:, 1])[:-1])).astype(np.int64)
    S = int((pyr[:, 0] * pyr[:, 1]).sum()); M = int(rng.choice([1, 2]))
    spread = float(rng.choice([0.5, 1.5, 3.0, 6.0]))
    ref = []
    for H, W in pyr:
        ys, xs = np.meshgrid((np.arange(H) + 0.5) / H, (np.arange(W) + 0.5) / W, indexing='ij'); ref.append(np.stack([xs.ravel(), ys.ravel()], -1))
    ref = np.concatenate(ref, 0)
    off = rng.standard_normal((1, S, M, 4, 4, 2)) * spread
    loc = (ref[None, :, None, None, None, :] + off / np.stack([pyr[:, 1], pyr[:, 0]], -1)[None, None, None, :, None, :]).astype(np.float32)
    aw = rng.random((1, S, M, 4, 4)); aw = (aw / aw.sum((-1, -2), keepdims=True)).astype(np.float32)
    value = rng.standard_normal((1, S, M, 32)); go = rng.standard_normal((1, S, M * 32))
    with open('/tmp/af_problem.bin', 'wb') as f:
        f.write(np.asarray([1, S, M, S] + [int(v) for hw in pyr for v in hw], dtype=np.int32).tobytes())
        f.write(bf16_bits(value).tobytes()); f.write(starts.tobytes()); f.write(loc.tobytes()); f.write(aw.tobytes()); f.write(bf16_bits(go).tobytes())
    fused = bool(rng.integers(0, 2))
    try:
        base = run({}, fused)
        far = int(base[-4:].view(np.int32)[0])
        res = []
        for env in ({"RLIPV2_CELL_SHARED": "3", "RLIPV2_PATCH_MULTI": "1"}, {"RLIPV2_CELL_SHARED": "2", "RLIPV2_PATCH_REPS": "3"}):
            got = run(env, fused)
            res.append(bool(np.array_equal(got, base)))
        print('ok  ' if all(res) else 'FAIL', f"#{n} pyr={pyr.tolist()} M={M} spread={spread} fused={fused} far={far} same={res}", flush=True)
    except Exception as e:
        print('ERR ', f"#{n} pyr={pyr.tolist()} M={M} spread={spread} fused={fused}", repr(e)[:200], flush=True)
