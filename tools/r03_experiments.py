"""The default-off experiments of round 3, one A/B table (ablation build: the switches are read there only):
    RLIPV2_CELL_SHARED = 1 | 2 | 3  cell_backward_kernel: sample geometry once per quad | + operand swap of the dot MFMAs |
                                    + level starts not through a dependent vector load, window copies issued up front
    RLIPV2_PATCH_MULTI = 1          patch_dest_kernel: the experimental instantiation (mask-word prefetch not in a branch)
    RLIPV2_PATCH_REPS  = 2..8       that + several patches per wave on the fine levels
Every arm must reproduce the default's three gradients bit for bit (B0 signature and fused geometry); the whole backward
is timed with HIP events.
    make -C rlipv2_amd/csrc ablation && RLIPV2_LIB_PATH=$PWD/tools/_build/librlipv2_msda_ablation.so python tools/r03_experiments.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlipv2_amd import msda  # noqa: E402
from tools.msda_inputs import PYRAMID_800x1333, make_inputs  # noqa: E402
from tools.patch_check import timed  # noqa: E402

ARMS = [("default", {}), ("cell 1 shared", {"RLIPV2_CELL_SHARED": "1"}), ("cell 2 +swap", {"RLIPV2_CELL_SHARED": "2"}),
        ("cell 3 +loads", {"RLIPV2_CELL_SHARED": "3"}), ("cell 4 no swap", {"RLIPV2_CELL_SHARED": "4"}),
        ("patch multi1", {"RLIPV2_PATCH_MULTI": "1"}),
        ("patch reps2", {"RLIPV2_PATCH_REPS": "2"}), ("patch reps3", {"RLIPV2_PATCH_REPS": "3"}),
        ("patch reps4", {"RLIPV2_PATCH_REPS": "4"}), ("patch reps8", {"RLIPV2_PATCH_REPS": "8"}),
        ("cell 3, multi1", {"RLIPV2_CELL_SHARED": "3", "RLIPV2_PATCH_MULTI": "1"}),
        ("cell 3, reps2", {"RLIPV2_CELL_SHARED": "3", "RLIPV2_PATCH_REPS": "2"}),
        ("cell 3, reps4", {"RLIPV2_CELL_SHARED": "3", "RLIPV2_PATCH_REPS": "4"})]
KEYS = ("RLIPV2_CELL_SHARED", "RLIPV2_PATCH_REPS", "RLIPV2_PATCH_MULTI")


def set_arm(env):
    for k in KEYS:
        os.environ.pop(k, None)
    os.environ.update(env)


def fused_problem(N, inp):
    value, shapes, starts = inp["value"], inp["shapes"], inp["starts"]
    S, M, L, P = value.shape[1], 8, 4, 4
    g = torch.Generator(device="cuda").manual_seed(1)
    ref = []
    for (H, W) in PYRAMID_800x1333:
        ys, xs = torch.meshgrid((torch.arange(H, device="cuda") + 0.5) / H, (torch.arange(W, device="cuda") + 0.5) / W,
                                indexing="ij")
        ref.append(torch.stack([xs.reshape(-1), ys.reshape(-1)], -1))
    ref = torch.cat(ref, 0)[None, :, None, :].expand(N, S, L, 2).contiguous()
    qproj = torch.randn(N, S, M * L * P * 3, device="cuda", generator=g)
    qproj[..., :M * L * P * 2] *= 2.5
    return qproj.bfloat16(), ref


def main():
    # (the static switches of the launchers are read once per process: one process per arm)
    if len(sys.argv) > 1:
        arm = int(sys.argv[1])
        name, env = ARMS[arm]
        set_arm(env)
        out = {}
        for N, mode in ((4, "model"), (4, "init"), (1, "model")):
            inp = make_inputs(N, mode=mode, dtype=torch.bfloat16, seed=3)
            a = (inp["value"], inp["shapes"], inp["starts"], inp["loc"], inp["aw"], inp["grad_out"])
            res = [t.float().cpu() for t in msda.ms_deform_attn_backward(*a, 64)]
            t = timed(lambda: msda.ms_deform_attn_backward(*a, 64), iters=20)
            out[f"b0 N={N} {mode}"] = (res, t)
            if mode == "model":
                msda.attach_host_shapes(inp["shapes"], PYRAMID_800x1333)
                qproj, ref = fused_problem(N, inp)
                _, loc, aw = msda.ms_deform_attn_fused_forward(inp["value"], inp["shapes"], inp["starts"], qproj, ref, True)
                hs = msda.host_shapes(inp["shapes"])
                f = lambda: msda.ms_deform_attn_fused_backward(inp["value"], inp["shapes"], inp["starts"], loc, aw, ref,  # noqa: E731
                                                               inp["grad_out"], hs)
                res = [t.float().cpu() for t in f() if torch.is_tensor(t)]
                out[f"fused N={N} {mode}"] = (res, timed(f, iters=20))
        torch.save(out, f"/tmp/r03_arm_{arm}.pt")
        return
    import subprocess
    base = None
    for arm, (name, env) in enumerate(ARMS):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), str(arm)], capture_output=True, text=True, timeout=600)
        if r.returncode != 0:
            print(f"{name:16s} FAILED rc={r.returncode}: {r.stderr[-400:]}", flush=True)
            continue
        out = torch.load(f"/tmp/r03_arm_{arm}.pt")
        if base is None:
            base = out
        for case, (res, t) in out.items():
            same = all(torch.equal(x, y) for x, y in zip(res, base[case][0]))
            finite = all(bool(torch.isfinite(x).all()) for x in res)
            worst = max(float((x - y).abs().max() / y.abs().max().clamp_min(1e-30)) for x, y in zip(res, base[case][0]))
            print(f"{name:16s} {case:18s} {t:8.1f} us (default {base[case][1]:8.1f})  equal bits {same}  finite {finite}  "
                  f"max rel diff {worst:.2e}", flush=True)


if __name__ == "__main__":
    main()
