"""TunableOp on the input-gradient GEMMs of the token-major Linears, each shape in isolation (a whole-step tuning
run hit a memory fault in one candidate solution during the backward pass; this keeps the candidates per call few)."""
import os, sys, torch
import torch.cuda.tunable as tn
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "tunableop_dgrad.csv")
tn.enable(True); tn.tuning_enable(True)
tn.set_max_tuning_duration(30); tn.set_max_tuning_iterations(20)
tn.set_filename(out)
T = 4 * 22223
shapes = [(T, 2048, 256), (T, 256, 2048), (T, 256, 256), (T, 384, 256)]
if len(sys.argv) > 1:
    shapes = [shapes[int(sys.argv[1])]]
for t, n, k in shapes:                     # dx[t, k] = dy[t, n] @ w[n, k]
    dy = torch.randn(t, n, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(n, k, device="cuda", dtype=torch.bfloat16)
    dx = dy @ w
    torch.cuda.synchronize()
    print("tuned", t, n, k, flush=True)
