"""Triage of the device code that has NEVER executed on hardware (tests/golden/isa_manifest.json: `never_run`) -- and of the
ablation-build arms, which are not in the manifest at all -- one FAMILY at a time: every family is one or more child processes in
a process group of their own with a timeout, followed by a health check of the device, and ends in ONE verdict line

    TRIAGE family=<name> verdict=<PASS|FAIL|TIMEOUT|SKIPPED|DEVICE-LOST> wall_s=<s> never_run_kernels=<n> default_after_promotion=<yes|no> row=<SURVEY 8 row>

A kernel that hangs or faults takes its child with it, never the families after it (unless the device itself is gone: the
triage then stops and says so).  Order = (would be on the product's default path once promoted) -> (SURVEY.md 8 a-e) -> (8 f).
The whole table is budgeted at <= 20 GPU-minutes (sum of the family timeouts; asserted in tests/test_bench_host.py).

    python tools/gpu_triage_r06.py                       # on the GPU box: parity tests of every family + the A/B timings
    python tools/gpu_triage_r06.py --dry-run             # no GPU: the SAME families through their tests on the lane-level model
                                                         # (tools/emu/) -- what profiles/r06_triage_dry_run.txt records
    python tools/gpu_triage_r06.py --only records_route,cell_forward
Results: gpurun_out/r06/triage.json (+ .txt: the table), one line per family on stdout as it finishes.
"""
import argparse
import json
import os
import re
import signal
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PY = sys.executable
# (--runxfail + the child marker: a family's child IS the isolation, so its `first_contact` tests run in-process and count)
PYTEST = [PY, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", "--runxfail"]
X = [PY, os.path.join("tools", "experiments_r05.py")]

# name, SURVEY 8 row, on the default path once promoted?, regexes over the manifest's never_run kernel names, GPU commands (each
# (argv, extra environment)), lane-model commands of the dry run, timeout of the whole family on the GPU (seconds)
FAMILIES = [
    dict(name="records_route", row="a2", default_after_promotion=True,
         what="encoder MSDA pair: record-emitting forward + geometry-free backward (msda_cell_forward.inc EMIT, msda_cell_records.inc)",
         kernels=[r"cell_forward_kernel<\d, [23]>", r"cell_records_backward_kernel", r"records_unbin_kernel"],
         gpu=[(PYTEST + ["-m", "gpu", "tests/test_zzz_records_gpu.py"], {}), (X + ["--records"], {})],
         emu=[(PYTEST + ["tests/test_records_emulated.py"], {})], timeout=210),
    dict(name="cell_forward", row="a1", default_after_promotion=True,
         what="encoder MSDA forward from LDS windows on the matrix cores (cell_forward_kernel<., 0>, explicit variant `cell`)",
         kernels=[r"cell_forward_kernel<\d, 0>"],
         gpu=[(PYTEST + ["-m", "gpu", "tests/test_msda_cell_forward_gpu.py"], {}), (X + ["--fwd"], {})],
         emu=[(PYTEST + ["tests/test_cell_forward_emulated.py"], {})], timeout=120),
    dict(name="backward_arms", row="a2", default_after_promotion=True,
         what="ablation build: cell_backward_kernel modes 2-4, patch_dest_multi_kernel (MULTI, REPS, CELLG + grad_out_cells_kernel); "
              "uniform locations: far-return + gated K1, queue-fed fallback launches",
         kernels=[],                                                 # (ablation-only instantiations: not in the product manifest)
         gpu=[(X + ["--arms"], {}), (X + ["--uniform-arms"], {})],
         emu=[(PYTEST + ["tests/test_backward_emulated.py"], {}),
              (PYTEST + ["tests/test_msda_emulated_library.py", "-k", "far_return"], {})], timeout=270),
    dict(name="data_parallel_default_path", row="e", default_after_promotion=True,
         what="what every N > 1 run executes by default and no GPU has run: fused AdamW with grad_scale != 1 (fused_adamw.hip "
              "step_scaled_kernel), the auto gradient schedule with real captured steps on a 1-rank RCCL group",
         kernels=[r"fused_adamw\.hip::.*step_scaled_kernel"],
         gpu=[(PYTEST + ["-m", "gpu", "tests/test_optim_gpu.py", "tests/test_linear_gpu.py", "-k", "gradient_scale or grad_scale"], {}),
              (PYTEST + ["-m", "gpu", "tests/test_zz_round6_gpu.py", "-k", "auto_gradient_schedule"], {})],
         emu=[(PYTEST + ["tests/test_dense_emulated.py", "-k", "fused_adamw"], {}),
              (PYTEST + ["tests/test_dp_cpu.py", "-k", "auto_gradient_schedule"], {})], timeout=240),
    dict(name="decoder_sample_then_project", row="a5/a11", default_after_promotion=False,
         what="decoder cross-attention, sample the unprojected memory then project (msda_rows.hip scatter + dots kernels)",
         kernels=[r"msda_rows\.hip::"],
         gpu=[(PYTEST + ["-m", "gpu", "tests/test_zz_round5_gpu.py", "-k", "rows_backward or sample_then_project"], {}), (X + ["--stp"], {})],
         emu=[(PYTEST + ["tests/test_msda_emulated_library.py", "-k", "rows_backward or sample_then_project"], {})], timeout=120),
    dict(name="swin_wide_layer_norm", row="f3", default_after_promotion=False,
         what="Swin blocks: residual add + LayerNorm at the Swin widths (layernorm_wide.hip, 40 instantiations)",
         kernels=[r"layernorm_wide\.hip::"],
         gpu=[(PYTEST + ["-m", "gpu", "tests/test_zz_round5_gpu.py", "-k", "wide_layer_norm or swin_step"], {})],
         emu=[(PYTEST + ["tests/test_dense_emulated.py", "-k", "wide_layernorm"], {})], timeout=120),
    dict(name="swin_window_attention", row="f3", default_after_promotion=False,
         what="Swin WindowAttention as one kernel per direction (window_attention.hip) + the stage-0 A/B",
         kernels=[r"window_attention\.hip::"],
         gpu=[(PYTEST + ["-m", "gpu", "tests/test_zz_round5_gpu.py", "-k", "window_attention"], {}), (X + ["--swin"], {})],
         emu=[(PYTEST + ["tests/test_dense_emulated.py", "-k", "window_attention"], {}),
              (PYTEST + ["tests/test_swin_fused_emulated.py", "-k", "window_attention_module"], {})], timeout=120),
]
BUDGET_S = 20 * 60


def never_run_kernels():
    with open(os.path.join(ROOT, "tests", "golden", "isa_manifest.json")) as f:
        return sorted(k for k, v in json.load(f)["kernels"].items() if v["hardware"] == "never_run")


def assign(kernels=None):
    """{family name: [never_run kernel names]} + the kernels no family claims (must be empty: tests/test_bench_host.py)"""
    kernels = never_run_kernels() if kernels is None else kernels
    out, left = {f["name"]: [] for f in FAMILIES}, []
    for k in kernels:
        for f in FAMILIES:
            if any(re.search(p, k) for p in f["kernels"]):
                out[f["name"]].append(k)
                break
        else:
            left.append(k)
    return out, left


def run_group(argv, env, timeout, log):
    """one child in its own session (process group), killed as a group on timeout -> (rc | "timeout", seconds, tail of its output)"""
    t0 = time.time()
    with open(log, "ab") as lf:
        lf.write(("\n$ " + " ".join(argv) + "\n").encode())
        lf.flush()
        proc = subprocess.Popen(argv, stdout=lf, stderr=subprocess.STDOUT, env=env, cwd=ROOT, start_new_session=True)
        try:
            rc = proc.wait(timeout=timeout)
        except subprocess.TimeoutExpired:
            try:
                os.killpg(proc.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
            proc.wait()
            rc = "timeout"
    with open(log, "rb") as lf:
        lf.seek(max(0, os.path.getsize(log) - 1500))
        tail = lf.read().decode(errors="replace")
    return rc, round(time.time() - t0, 1), tail


def device_alive(env, log):
    rc, _, _ = run_group([PY, "-c", "import torch; x = torch.ones(1 << 20, device='cuda:0'); assert float((x * 2).sum()) == 2 ** 21"],
                         env, 90, log)
    return rc == 0


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--dry-run", action="store_true", help="no GPU: the families' tests on the lane-level model of tools/emu/")
    ap.add_argument("--only", default="", help="comma-separated family names")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r06"))
    ap.add_argument("--dry-timeout", type=int, default=1500, help="per family, dry run only (the model is ~1000x slower than the device)")
    args = ap.parse_args(argv)
    os.makedirs(args.out, exist_ok=True)
    only = {s for s in args.only.split(",") if s}
    claimed, unclaimed = assign()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["RLIPV2_TEST_FIRST_CONTACT_CHILD"] = "1"
    tag = "triage_dry_run" if args.dry_run else "triage"
    rows, lost = [], False
    t_start = time.time()
    for f in FAMILIES:
        if only and f["name"] not in only:
            continue
        log = os.path.join(args.out, f"{tag}_{f['name']}.log")
        open(log, "w").close()
        cmds = f["emu"] if args.dry_run else f["gpu"]
        deadline = time.time() + (args.dry_timeout if args.dry_run else f["timeout"])
        verdict, steps, t0 = "PASS", [], time.time()
        if lost:
            verdict = "SKIPPED"
        for cmd, extra in ([] if lost else cmds):
            left = deadline - time.time()
            if left < 5:
                verdict = "TIMEOUT"
                steps.append({"cmd": " ".join(cmd[1:]), "rc": "not started: the family's time was used up"})
                break
            rc, secs, tail = run_group(cmd, dict(env, **extra), left, log)
            steps.append({"cmd": " ".join(cmd[1:]), "rc": rc, "wall_s": secs, "tail": tail[-600:]})
            if rc == "timeout":
                verdict = "TIMEOUT"
                break
            if rc != 0:
                verdict = "FAIL"                                      # (keep going: the family's other children still inform)
        if not args.dry_run and not lost and not device_alive(env, log):
            verdict, lost = "DEVICE-LOST", True
        row = {"family": f["name"], "verdict": verdict, "wall_s": round(time.time() - t0, 1), "row": f["row"], "what": f["what"],
               "default_after_promotion": f["default_after_promotion"], "never_run_kernels": len(claimed[f["name"]]),
               "steps": steps}
        rows.append(row)
        print(f"TRIAGE family={f['name']} verdict={verdict} wall_s={row['wall_s']} never_run_kernels={row['never_run_kernels']} "
              f"default_after_promotion={'yes' if f['default_after_promotion'] else 'no'} row={f['row']}", flush=True)
    rep = {"mode": "dry run on the lane-level model (tools/emu/): logic only, no code generation, no timing" if args.dry_run
                   else "GPU", "wall_s": round(time.time() - t_start, 1), "unclaimed_never_run_kernels": unclaimed, "families": rows}
    with open(os.path.join(args.out, tag + ".json"), "w") as fp:
        json.dump(rep, fp, indent=1)
    with open(os.path.join(args.out, tag + ".txt"), "w") as fp:
        fp.write(f"{rep['mode']}; {rep['wall_s']} s\n")
        fp.write(f"{'family':30s} {'verdict':12s} {'wall s':>7s} {'never-run':>9s} {'default?':>8s} row   what\n")
        for r in rows:
            fp.write(f"{r['family']:30s} {r['verdict']:12s} {r['wall_s']:7.1f} {r['never_run_kernels']:9d} "
                     f"{'yes' if r['default_after_promotion'] else 'no':>8s} {r['row']:5s} {r['what']}\n")
            for s in r["steps"]:
                fp.write(f"    rc={s['rc']}  {s.get('wall_s', '')} s  {s['cmd']}\n")
        if unclaimed:
            fp.write("NOT COVERED BY ANY FAMILY: " + ", ".join(unclaimed) + "\n")
    return 0 if all(r["verdict"] == "PASS" for r in rows) else 1


if __name__ == "__main__":
    raise SystemExit(main())
