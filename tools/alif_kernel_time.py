"""Stand-alone timing of the ALIF forward kernel (HIP events around the C-ABI call)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlipv2_amd import _lib
L = _lib.lib()
DEV = "cuda:0"
for B in (4, 32):
    H, Tv, Tl, hd = 8, 273, 64, 256
    E = H * hd
    Tvp = L.alif_attention_padded_tv(Tv)
    q = (torch.randn(B, Tv, E, device=DEV) * 0.08).bfloat16(); k = torch.randn(B, Tl, E, device=DEV).bfloat16()
    vlt = torch.randn(B, E, 64, device=DEV).bfloat16(); vvt = torch.randn(B, E, Tvp, device=DEV).bfloat16()
    ov = torch.empty_like(q); ol = torch.empty_like(k)
    pv = torch.empty(B, H, Tv, Tl, device=DEV, dtype=torch.bfloat16); pl = torch.empty(B, H, Tl, Tv, device=DEV, dtype=torch.bfloat16)
    keep = (torch.rand(2, B, H, Tv * Tl, device=DEV) >= 0.1)
    st = torch.cuda.current_stream().cuda_stream
    for drop in (False, True):
        kv = keep[0].data_ptr() if drop else None; kl = keep[1].data_ptr() if drop else None
        call = lambda: L.alif_attention_forward_bf16(q.data_ptr(), k.data_ptr(), vlt.data_ptr(), vvt.data_ptr(), kv, kl, 1.0 / 0.9 if drop else 1.0, B, H, Tv, Tl, ov.data_ptr(), ol.data_ptr(), pv.data_ptr(), pl.data_ptr(), st)
        for _ in range(5): assert call() == 0
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50): call()
        b.record(); torch.cuda.synchronize()
        print(f"B={B} drop={drop}: {a.elapsed_time(b) / 50 * 1e3:7.1f} us per launch ({B * H} workgroups)")
