"""Generate op-level golden vectors for multi-scale deformable attention.

Runs ONLY in the build container, where the reference checkout is mounted at
/root/reference.  It imports the reference's own pure-PyTorch implementation
(`ms_deform_attn_core_pytorch`, models/ops/functions/ms_deform_attn_func.py:45-65)
without touching the rest of the package (a dummy `MultiScaleDeformableAttention`
module satisfies the file's top-level import), runs it forward and differentiates it
with autograd, and stores inputs + outputs + the three gradients as .npz files next to
this script.  The .npz files are data (inputs and expected outputs); nothing of the
reference's source travels.

Cases
-----
* testpy_*   : the recipe of the reference's only test, models/ops/test.py:25-40
               (N1 M2 D2 Lq2 L2 P2, shapes (6,4),(3,2), torch.manual_seed(3)), and the same
               recipe with D in {30, 32, 64, 71} (test.py:89-90, the values small enough to
               store).
* pyr_*      : the same recipe on the pyramid (20,27),(10,14),(5,7),(3,4) of SURVEY 8c
* model_*    : M8 D32 L4 P4 on a 4-level pyramid, Lq = S ("encoder") and Lq = 50
               ("decoder"), with sampling locations that leave [0,1], sit exactly on pixel
               centres / borders, and hit the -1 / H boundary cases of
               ms_deform_im2col_cuda.cuh:285-288.

Every case stores float32-representable inputs; `*_f64` outputs come from running the
reference in float64 on those inputs, `*_f32` outputs from running it in float32.

usage:  python tests/golden/make_msda_golden.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def load_reference_core():
    sys.modules.setdefault("MultiScaleDeformableAttention", types.ModuleType("MultiScaleDeformableAttention"))
    spec = importlib.util.spec_from_file_location(
        "_ref_msda_func", os.path.join(REF, "models/ops/functions/ms_deform_attn_func.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.ms_deform_attn_core_pytorch


def run_reference(core, value, shapes, loc, aw, grad_out, dtype):
    v = torch.from_numpy(value).to(dtype).requires_grad_(True)
    l = torch.from_numpy(loc).to(dtype).requires_grad_(True)
    a = torch.from_numpy(aw).to(dtype).requires_grad_(True)
    out = core(v, torch.from_numpy(shapes), l, a)
    out.backward(torch.from_numpy(grad_out).to(dtype))
    return (out.detach().numpy(), v.grad.numpy(), l.grad.numpy(), a.grad.numpy())


def starts_of(shapes):
    hw = shapes[:, 0] * shapes[:, 1]
    return np.concatenate([[0], np.cumsum(hw)[:-1]]).astype(np.int64)


def save_case(core, name, value, shapes, loc, aw, grad_out):
    shapes = np.asarray(shapes, dtype=np.int64)
    rec = dict(value=value, shapes=shapes, starts=starts_of(shapes), loc=loc, aw=aw, grad_out=grad_out)
    for tag, dt in (("f64", torch.float64), ("f32", torch.float32)):
        out, gv, gl, ga = run_reference(core, value, shapes, loc, aw, grad_out, dt)
        rec.update({f"out_{tag}": out, f"g_value_{tag}": gv, f"g_loc_{tag}": gl, f"g_aw_{tag}": ga})
    path = os.path.join(HERE, f"msda_{name}.npz")
    np.savez_compressed(path, **rec)
    print(f"{name:24s} value{value.shape} loc{loc.shape} -> {os.path.getsize(path)/1024:.0f} KiB")


def testpy_case(core, D, tag):
    # models/ops/test.py:25-40
    N, M = 1, 2
    Lq, L, P = 2, 2, 2
    shapes = np.array([(6, 4), (3, 2)], dtype=np.int64)
    S = int((shapes[:, 0] * shapes[:, 1]).sum())
    torch.manual_seed(3)
    value = (torch.rand(N, S, M, D) * 0.01).numpy()
    loc = torch.rand(N, Lq, M, L, P, 2).numpy()
    aw = torch.rand(N, Lq, M, L, P) + 1e-5
    aw = (aw / aw.sum(-1, keepdim=True).sum(-2, keepdim=True)).numpy()
    grad_out = torch.randn(N, Lq, M * D).numpy()
    save_case(core, tag, value, shapes, loc, aw, grad_out)


def model_case(core, Lq_mode, tag, seed, shapes=((10, 14), (5, 7), (3, 4), (2, 2)), N=2):
    rng = np.random.default_rng(seed)
    M, D, L, P = 8, 32, 4, 4
    shapes = np.array(shapes, dtype=np.int64)
    S = int((shapes[:, 0] * shapes[:, 1]).sum())
    Lq = S if Lq_mode == "enc" else 50
    value = rng.standard_normal((N, S, M, D)).astype(np.float32)
    # reference points: pixel centres of the pyramid (encoder) or random (decoder)
    if Lq_mode == "enc":
        ref = []
        for (H, W) in shapes:
            ys, xs = np.meshgrid((np.arange(H) + 0.5) / H, (np.arange(W) + 0.5) / W, indexing="ij")
            ref.append(np.stack([xs.ravel(), ys.ravel()], -1))
        ref = np.concatenate(ref, 0)[None].repeat(N, 0)                     # [N,S,2]
    else:
        ref = rng.uniform(-0.1, 1.1, size=(N, Lq, 2))
    off = rng.standard_normal((N, Lq, M, L, P, 2)) * 2.0                      # pixels
    norm = np.stack([shapes[:, 1], shapes[:, 0]], -1).astype(np.float64)      # (W, H)
    loc = ref[:, :, None, None, None, :] + off / norm[None, None, None, :, None, :]
    # boundary cases of ms_deform_im2col_cuda.cuh:285-288, planted on known samples
    for l, (H, W) in enumerate(shapes):
        loc[0, 0, 0, l, 0] = (0.5 / W, 0.5 / H)              # exactly on pixel (0,0) centre
        loc[0, 0, 1, l, 1] = (-0.5 / W, 0.25)                # w_im == -1  -> excluded
        loc[0, 0, 2, l, 2] = (0.25, (H + 0.5) / H)           # h_im == H   -> excluded
        loc[0, 1, 0, l, 0] = (0.0, 0.0)                      # h_im = w_im = -0.5 (corner)
        loc[0, 1, 1, l, 1] = (1.0, 1.0)                      # opposite corner
        loc[0, 1, 2, l, 2] = ((W - 0.5) / W, (H - 0.5) / H)  # last pixel centre
        loc[0, 1, 3, l, 3] = (3.0, -2.0)                     # far outside
        loc[N - 1, 2, 4, l, 0] = (1.0 + 0.49 / W, 0.5)       # w_im just below W
        loc[N - 1, 2, 5, l, 1] = (0.5, -0.49 / H)            # h_im just above -1
    loc = loc.astype(np.float32)
    logits = rng.standard_normal((N, Lq, M, L * P))
    aw = np.exp(logits - logits.max(-1, keepdims=True))
    aw = (aw / aw.sum(-1, keepdims=True)).reshape(N, Lq, M, L, P).astype(np.float32)
    grad_out = rng.standard_normal((N, Lq, M * D)).astype(np.float32)
    save_case(core, tag, value, shapes, loc, aw, grad_out)


def main():
    if not os.path.isdir(REF):
        raise SystemExit("reference checkout not mounted; goldens can only be regenerated in the build container")
    core = load_reference_core()
    testpy_case(core, 2, "testpy_d2")
    for D in (30, 32, 64, 71):
        testpy_case(core, D, f"testpy_d{D}")
    model_case(core, "enc", "model_enc", seed=11)
    model_case(core, "dec", "model_dec", seed=12)
    # the pyramid SURVEY 8c names -- (20,27),(10,14),(5,7),(3,4): 2 x 2 cells of 16 x 16 level-0 pixels, a ragged last
    # cell row / column on every level; the encoder case with one image to keep the file small
    pyr = ((20, 27), (10, 14), (5, 7), (3, 4))
    model_case(core, "enc", "pyr_enc", seed=13, shapes=pyr, N=1)
    model_case(core, "dec", "pyr_dec", seed=14, shapes=pyr, N=2)


if __name__ == "__main__":
    main()
