"""Random problems through the MSDA library built for the lane-level workgroup model (tools/emu/build_lib.sh), checked against
the oracle: ragged and degenerate pyramids (levels that do not halve, 1-pixel levels), 1-3 images, 1-8 heads, encoder calls
(Lq == S; bfloat16 ones take the matrix-core backward, some also the experimental "cell" forward) and decoder calls, offsets
from half a pixel to far outside the image.  Not part of the test suite (minutes of host time): a way to look for latent
kernel bugs without a GPU.     usage: tools/emu/build_lib.sh /tmp/libmsda_emu.so && python tests/scripts/fuzz_emulated_library.py <seed> <seconds>
(lives under tests/: it uses the oracle as the checker)"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from oracle import msda_oracle as O
import test_msda_emulated_library as T
from conftest import kink_samples
lib = T.EmuLib('/tmp/libmsda_emu.so')
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv)>1 else 0)
t_end = time.time() + float(sys.argv[2] if len(sys.argv)>2 else 300)
n=0
while time.time() < t_end:
    n+=1
    H0, W0 = int(rng.integers(1, 41)), int(rng.integers(1, 41))
    pyr=[(H0,W0)]
    for l in range(3):
        h,w = pyr[-1]; pyr.append((max(1,(h+1)//2 if rng.random()<0.7 else int(rng.integers(1,h+1))), max(1,(w+1)//2 if rng.random()<0.7 else int(rng.integers(1,w+1)))))
    pyr=np.asarray(pyr,dtype=np.int64)
    starts=np.concatenate(([0],np.cumsum(pyr[:,0]*pyr[:,1])[:-1])).astype(np.int64)
    S=int((pyr[:,0]*pyr[:,1]).sum())
    N=int(rng.integers(1,4)); M=int(rng.choice([1,2,3,8]))
    enc = rng.random()<0.6
    Lq = S if enc else int(rng.integers(1,60))
    spread = float(rng.choice([0.5, 2.0, 8.0, 100.0]))
    if enc:
        ref=[]
        for H,W in pyr:
            ys,xs=np.meshgrid((np.arange(H)+0.5)/H,(np.arange(W)+0.5)/W,indexing='ij'); ref.append(np.stack([xs.ravel(),ys.ravel()],-1))
        ref=np.concatenate(ref,0)[None].repeat(N,0)
    else:
        ref=rng.uniform(-0.1,1.1,size=(N,Lq,2))
    off=rng.standard_normal((N,Lq,M,4,4,2))*spread
    loc=(ref[:,:,None,None,None,:]+off/np.stack([pyr[:,1],pyr[:,0]],-1)[None,None,None,:,None,:]).astype(np.float32)
    aw=rng.random((N,Lq,M,4,4)); aw=(aw/aw.sum((-1,-2),keepdims=True)).astype(np.float32)
    dt = T.BF16 if rng.random()<0.6 else T.F32
    value=rng.standard_normal((N,S,M,32)).astype(np.float32); go=rng.standard_normal((N,Lq,M*32)).astype(np.float32)
    if dt==T.BF16:
        value=T.bf16_val(T.bf16_bits(value)).astype(np.float32); go=T.bf16_val(T.bf16_bits(go)).astype(np.float32)
    g=dict(value=value,loc=loc,aw=aw,grad_out=go,shapes=pyr,starts=starts)
    a=(value.astype(np.float64),pyr,starts,loc.astype(np.float64),aw.astype(np.float64))
    ref_out=O.forward(*a); rgv,rgl,rga=O.backward(*a,go.astype(np.float64))
    fwd = "cell" if (enc and dt==T.BF16 and rng.random()<0.5) else "quad"
    desc=f"#{n} pyr={pyr.tolist()} N={N} M={M} Lq={Lq} enc={enc} spread={spread} dt={dt} fwd={fwd}"
    try:
        out,gv,gl,ga=lib.run(fwd,"dest",dt,g)
        tol = 2.0**-7 if dt==T.BF16 else 1e-4
        at = (1e-3 if dt==T.BF16 else 1e-5)
        np.testing.assert_allclose(out,ref_out,rtol=tol,atol=at*max(1.0,float(np.abs(ref_out).max())))
        np.testing.assert_allclose(gv,rgv,rtol=tol,atol=at*max(1.0,float(np.abs(rgv).max())))
        T.close32(ga,rga)
        keep=~kink_samples(g)
        T.close32(gl[keep],rgl[keep])
        print('ok  ',desc,flush=True)
    except Exception as e:
        print('FAIL',desc,str(e)[:300].replace('\n',' | '),flush=True)
