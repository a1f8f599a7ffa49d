"""Per-phase cycle stamps of the MSDA grad_value scatter kernel (timeline build: hipcc -DMSDA_K2_TIMELINE on
csrc/msda_window.hip, library path in RLIPV2_LIB_PATH)."""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlipv2_amd import msda, _lib
from tools.msda_inputs import make_inputs
mode = sys.argv[1] if len(sys.argv) > 1 else "model"
inp = make_inputs(4, Lq=None, mode=mode, dtype=torch.bfloat16)
a = (inp["value"], inp["shapes"], inp["starts"], inp["loc"], inp["aw"])
msda.set_variant("window")
for _ in range(3):
    msda.ms_deform_attn_backward(*a, inp["grad_out"], 64)
torch.cuda.synchronize()
buf = np.zeros(3 * 32 * 10, dtype=np.uint64)
st = _lib.lib().msda_debug_k2_timeline(buf.ctypes.data_as(ctypes.c_void_p))
ts = buf.reshape(3, 32, 10).astype(np.int64)
names = {1: "stage+geometry", 2: "histogram", 3: "scan a", 4: "scan b", 5: "scatter", 6: "pad", 8: "walk", 9: "end barrier"}
order = [1, 2, 3, 4, 5, 6, 8, 9]
print(f"status {st}; mode {mode}; cycles per phase, mean over the workgroup's first items (item = image, head, 16x16 tile, level)")
for p, blk in enumerate((0, 100, 301)):
    n = int((ts[p, :, 0] > 0).sum())
    if n < 2:
        continue
    d = {k: [] for k in order}
    for it in range(n):
        last = ts[p, it, 0]
        for k in order:
            d[k].append(ts[p, it, k] - last); last = ts[p, it, k]
    tot = (ts[p, n - 1, 9] - ts[p, 0, 0]) / n
    print(f"workgroup {blk}: {n} items, {tot:.0f} clk per item: " + ", ".join(f"{names[k]} {np.mean(d[k]):.0f}" for k in order))
    lv = [int(np.mean([sum(d[k][it] for k in order) for it in range(l, n, 4)])) for l in range(4)]
    print(f"    by sampled level (item & 3): {lv};  walk by level: {[int(np.mean(d[8][l::4])) for l in range(4)]}")
