"""One small encoder call through the records route (forward with EMIT = 3, records backward + patch pass) on the host-model
library given as argv[1] -- run by tests/test_emulated_tsan.py under ThreadSanitizer (python with LD_PRELOAD of the TSAN runtime,
the library built with EMU_TSAN=1): a lane pair that no rendezvous / fence / barrier orders is printed as a data race."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
from test_cell_forward_emulated import make_problem, bf16_bits, bf16_val
L = ctypes.CDLL(sys.argv[1])
vp, i, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
d=[i]*7
L.msda_records_bytes.argtypes = [i, vp, *d]; L.msda_records_bytes.restype = sz
L.msda_records_forward.argtypes = [i, vp, vp, vp, vp, vp, vp, i, vp, vp, *d, vp, vp, sz, vp]
L.msda_records_backward.argtypes = [i, i, vp, vp, vp, vp, vp, vp, vp, i, vp, *d, vp, vp, vp, vp, vp, sz, vp, sz, vp]
L.msda_backward_workspace_bytes.argtypes = [i, vp, *d]; L.msda_backward_workspace_bytes.restype = sz
p = lambda a: a.ctypes.data if a is not None else None
BF16=2
M=1
pyr, starts, S, value, loc, aw = make_problem([(20, 27), (10, 14), (5, 7), (3, 4)], M, (1.5, 1.5, 1.0, 0.7), seed=7)
rng = np.random.default_rng(3)
vb = np.ascontiguousarray(bf16_bits(value)); gob = np.ascontiguousarray(bf16_bits(rng.standard_normal((1, S, M*32))))
sh, st = np.ascontiguousarray(pyr, dtype=np.int64), np.ascontiguousarray(starts, dtype=np.int64)
dims = (1, S, M, 32, 4, S, 4)
rec_bytes = L.msda_records_bytes(BF16, p(sh), *dims); ws_bytes = L.msda_backward_workspace_bytes(BF16, p(sh), *dims)
records = np.zeros(rec_bytes, dtype=np.uint8); out = np.zeros((1, S, M*32), dtype=np.uint16)
# module operands, no saved locations (EMIT = 3)
qproj = rng.standard_normal((1, S, M*48)); qproj[..., :M*32] *= 2.0
qb = np.ascontiguousarray(bf16_bits(qproj))
refp = np.concatenate([np.stack([g.ravel() for g in np.meshgrid((np.arange(W)+0.5)/W, (np.arange(H)+0.5)/H)], -1) for H, W in pyr], 0)
ref = np.ascontiguousarray(np.broadcast_to(refp[None,:,None,:], (1,S,4,2)), dtype=np.float32)
print("forward", L.msda_records_forward(BF16, p(vb), p(sh), p(st), p(sh), p(qb), p(ref), 2, None, None, *dims, p(out), p(records), rec_bytes, None), flush=True)
gv = np.zeros(vb.shape, dtype=np.uint16); gq = np.zeros(qb.shape, dtype=np.uint16); ws = np.zeros(ws_bytes+64, dtype=np.uint8)
print("backward", L.msda_records_backward(0x200|0x400, BF16, p(vb), p(sh), p(st), p(sh), None, None, p(ref), 2, p(gob), *dims, p(gv), None, None, p(gq), p(records), rec_bytes, p(ws), ws_bytes, None), flush=True)
print("finite", np.isfinite(bf16_val(gv)).all(), np.isfinite(bf16_val(gq)).all())
