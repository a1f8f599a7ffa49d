"""Per-phase cycle stamps of one workgroup of expand_gemm (timeline build: hipcc -DXDBG=64, see tools/expand_ablate.sh)."""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlipv2_amd import linear, _lib
T, N = 4 * 22223, 2048
dy = torch.randn(T, 256, device="cuda", dtype=torch.bfloat16)
w2t = (torch.randn(N, 256, device="cuda") / 16).to(torch.bfloat16)
h = torch.randn(T, N, device="cuda").relu().to(torch.bfloat16)
b1 = torch.randn(N, device="cuda", dtype=torch.bfloat16)
names = ["barrier1", "mfma0", "stage0", "mfma1", "stage1", "barrier2", "dma+mask issue", "(7)", "readback+stores", "vmcnt0"]
for label, fn in (("dgrad+mask", lambda: linear.expand_gemm(dy, w2t, mask=h)),
                  ("fwd bias+relu", lambda: linear.expand_gemm(dy, w2t, bias=b1, relu=True))):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    buf = np.zeros(4 * 16 * 10, dtype=np.uint64)
    st = _lib.lib().linear_expand_debug_read(buf.ctypes.data_as(ctypes.c_void_p))
    ts = buf.reshape(4, 16, 10).astype(np.int64)
    print(f"== {label} (status {st}); rows: probe (block 0 / 301) x (wave 0 / 3); columns: cycles spent reaching each point")
    for p in range(4):
        steps = ts[p, :8]
        d = np.zeros((8, 10), dtype=np.int64)
        for s in range(8):
            prev = steps[s - 1, 9] if s else steps[0, 0]
            order = [0, 1, 2, 3, 4, 5, 6, 8, 9]
            last = prev
            for k in order:
                d[s, k] = steps[s, k] - last
                last = steps[s, k]
        print(f"probe {p}: step total {int((steps[7, 9] - steps[0, 0]) / 8)} clk;  mean per phase: " +
              ", ".join(f"{names[k]} {int(d[1:, k].mean())}" for k in [0, 1, 2, 3, 4, 5, 6, 8, 9]))
