"""Host logic of bench.py and of the evidence tools that needs no GPU: the `parallelism` text for 1 / N ranks (round 4's
bench.py died in emit() for every N > 1), which recorded traffic table the roofline quotes, the kernel-name labels of the
counter summary, and the verdict logic of the host-route self-check."""
import json
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from rlipv2_amd import routes  # noqa: E402
from tools import pmc_final_summary  # noqa: E402


@pytest.mark.parametrize("world", [1, 2, 8])
def test_parallelism_text(world):
    flat = bench.parallelism_text(world, graphed=True, overlap=False, dp_group=world > 1)
    assert flat.startswith(f"dp{world} (one flat") and "HIP graphs" in flat
    over = bench.parallelism_text(world, graphed=True, overlap=True, dp_group=True)
    assert over.startswith(f"dp{world} (bf16 RCCL gradient all-reduce in buckets") and "HIP graphs" in over
    # the overlapped text needs a process group: a 1-rank run without one never claims it
    assert bench.parallelism_text(world, graphed=True, overlap=True, dp_group=False) == bench.parallelism_text(world, True, False, False)
    eager = bench.parallelism_text(world, graphed=False, overlap=True, dp_group=True)
    assert eager.startswith(f"dp{world} (DDP") and "eager" in eager


def test_emit_of_a_multi_rank_line_needs_nothing_from_the_step_function(monkeypatch, capsys):
    """emit() with world = 2 and the values run_train_step_bench returns: one JSON line, n_gpus = 2 (round 4 raised NameError
    here, reading a local of another function)."""
    import argparse
    args = argparse.Namespace(batch=4, steps=2, warmup=1, dtype="bf16", no_cpu_baseline=True)

    class Lib:
        @staticmethod
        def msda_pick_variant(*a):
            return 0

        @staticmethod
        def msda_variant_name(v):
            return b"quad"

    kern = {"enc_bwd_fused": {"ms": 1.0, "n": 2, "dims": (4, 22223, 8, 32, 4, 22223, 4), "code": 2, "bwd": True, "variant": "cell+patch",
                              "bytes": 409_600_000, "operand_bytes": 300_000_000}}
    dp = {"graphed": True, "overlap": False, "dp_group": True}
    bench.emit(args, 2, 0.08, kern, Lib, workload_text="train_step: test", cpu_calls=None,
               parallelism=bench.parallelism_text(2, dp["graphed"], dp["overlap"], dp["dp_group"]),
               host_routes={"residual_gradient_in_gemm": "on", "one_launch_box_head": "off (self-check failed: test)"})
    line = json.loads(capsys.readouterr().out.strip())
    assert line["n_gpus"] == 2 and line["value"] == pytest.approx(4 * 2 * 2 / 0.08)
    assert line["config"]["parallelism"].startswith("dp2 (one flat")
    assert line["config"]["host_routes"]["one_launch_box_head"].startswith("off")
    assert "cpu_baseline" not in line and line["roofline"]["frac"] == pytest.approx(409.6e6 / 0.5e-3 / 1e9 / 8000, rel=1e-3)


def test_traffic_table_is_the_newest_committed_one_and_keys_must_match(tmp_path, monkeypatch):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    os.makedirs(tmp_path / "profiles")
    assert bench.newest_traffic_file() is None
    row = {"fetch_bytes": 100.0, "write_bytes": 50.0, "launches": 5}
    for rnd, keys in (("r03", ["msda:cell_backward_kernel+geometry", "msda:patch_dest_kernel"]),
                      ("r05", ["msda:cell_backward_kernel+geometry", "msda:patch_dest_kernel", "b0:cell_backward_kernel"])):
        json.dump({k: row for k in keys}, open(tmp_path / "profiles" / f"{rnd}_final_traffic.json", "w"))
    assert os.path.basename(bench.newest_traffic_file()) == "r05_final_traffic.json"
    t = json.load(open(bench.newest_traffic_file()))
    assert bench.traffic_from_table(t, "enc_bwd_fused") == 300
    assert bench.traffic_from_table(t, "enc_bwd") is None            # b0:patch_dest_kernel missing: null, not a stale number
    assert bench.traffic_from_table(t, "dec300_fwd") is None


def test_counter_summary_labels():
    s = pmc_final_summary.short
    assert s("void rlipv2::cell_backward_kernel<0, 0>(rlipv2::PatchPlan, ...)") == "cell_backward_kernel"
    assert s("void rlipv2::cell_backward_kernel<2, 0>(...)") == "cell_backward_kernel+geometry"
    assert s("void rlipv2::cell_backward_kernel<4, 2>(...)") == "cell_backward_kernel+geometry"
    assert s("void rlipv2::patch_dest_kernel<__hip_bfloat16, 2>(...)") == "patch_dest_kernel"
    assert s("void rlipv2::quad_backward_shared_kernel<__hip_bfloat16, 2>(...)") == "quad_backward_shared_kernel+geometry"
    assert s("void rlipv2::quad_forward_fused_kernel<...>(...)") == "quad_forward_fused_kernel"
    assert s("Cijk_Alik_Bljk_BBS_BH_Bias_HA_S_SAV_UserArgs_MT128x128x64") is None


def test_route_verdicts():
    """routes.compare: the tolerances the self-check applies (loss 1e-3, whole gradient max(2 %, 4 x step noise), every
    parameter 5 % of its norm + 1e-3 of the largest)"""
    g = torch.Generator().manual_seed(0)
    ref = [torch.randn(300, generator=g), torch.randn(40, 7, generator=g) * 1e-3, torch.randn(5, generator=g)]
    same = [t + 1e-3 * torch.randn(t.shape, generator=g) * t.abs().mean() for t in ref]
    assert routes.compare(1.0, same, 1.0, ref) is None
    assert "loss" in routes.compare(1.01, same, 1.0, ref)
    assert routes.compare(float("nan"), same, 1.0, ref) == "non-finite loss"
    lost = [ref[0] * 0.5, ref[1], ref[2]]                             # half of one contribution dropped
    assert "whole gradient" in routes.compare(1.0, lost, 1.0, ref)
    small = [ref[0], ref[1], ref[2] * 0.0]                             # a small parameter's gradient lost: whole gradient fine?
    why = routes.compare(1.0, small, 1.0, ref)
    assert why is not None
    nan = [ref[0], ref[1] * float("nan"), ref[2]]
    assert routes.compare(1.0, nan, 1.0, ref) == "non-finite gradient"
    # a noisy step (MIOpen's non-deterministic convolution) widens the whole-gradient bar, nothing else
    noisy = [t + 0.03 * torch.randn(t.shape, generator=g) * t.abs().mean() for t in ref]
    assert routes.compare(1.0, noisy, 1.0, ref, noise=0.0) is not None
    assert routes.compare(1.0, noisy, 1.0, ref, noise=0.02) is None


def test_routes_are_off_in_the_package_and_stay_off_without_a_gpu():
    from rlipv2_amd import train
    assert routes.state() == {"residual_gradient_in_gemm": False, "one_launch_box_head": False, "fused_wide_layer_norm": False,
                              "fused_window_attention": False}
    samples, text, targets = train.synthetic_batch(1, 32, 32, n_obj=3, n_verb=2, triplets=1, device="cpu")
    verdict = routes.validate(None, None, (samples, text, targets))
    assert all(v.startswith("off (not applicable") for v in verdict.values()) and not any(routes.state().values())


def test_set_flag_flips_existing_switches_only():
    from rlipv2_amd import decoder, norm
    try:
        assert bench.apply_overrides(["decoder.fused_glue=0", "norm.MIN_ROWS=512"]) == {"decoder.fused_glue": False, "norm.MIN_ROWS": 512}
        assert decoder.fused_glue is False and norm.MIN_ROWS == 512
    finally:
        decoder.fused_glue, norm.MIN_ROWS = True, 256
    for bad in ("decoder.no_such_switch=1", "decoder.box_head=0", "decoder.fused_glue"):
        with pytest.raises(SystemExit):
            bench.apply_overrides([bad])


def test_package_reads_no_code_path_switch_from_the_environment():
    """~15 environment-variable A/B switches were read inside the product package.  What is left is
    configuration of the libraries underneath (library path, MIOpen / hipBLASLt tuning tables)."""
    import re
    allowed = {"RLIPV2_LIB_PATH", "RLIPV2_CPU_LIB_PATH", "RLIPV2_TUNED_MIOPEN", "MIOPEN_USER_DB_PATH", "XDG_CACHE_HOME", "RLIPV2_TUNED_GEMM_TABLE",
               "RLIPV2_TUNED_GEMMS"}
    pkg = os.path.join(ROOT, "rlipv2_amd")
    seen = set()
    for f in os.listdir(pkg):
        if f.endswith(".py"):
            seen |= set(re.findall(r"environ(?:\.get)?[\(\[]\s*\"([A-Z0-9_]+)\"", open(os.path.join(pkg, f)).read()))
    assert seen <= allowed, seen - allowed


def test_experiments_leg_only_in_the_full_default_run(monkeypatch):
    """bench.py appends the A/B table of the unmeasured kernel arms (tools/experiments_r05.py, child processes) to the default
    1-GPU evidence run only -- never under a profiler, with overrides, on another configuration, or on several ranks."""
    import argparse
    from tools import experiments_r05
    monkeypatch.setattr(experiments_r05, "main", lambda: {"ran": True})
    monkeypatch.setattr(bench.torch.cuda, "synchronize", lambda: None)
    monkeypatch.setattr(bench.torch.cuda, "empty_cache", lambda: None)
    base = dict(experiments=True, no_cpu_baseline=False, backbone="resnet50", dtype="bf16", batch=4, overrides=[], padded=False,
                var_targets=False)
    assert bench.run_experiments(argparse.Namespace(**base), 1) == {"ran": True}
    assert bench.run_experiments(argparse.Namespace(**base), 2) is None
    for k, v in (("experiments", False), ("no_cpu_baseline", True), ("batch", 8), ("overrides", ["a.b=1"]), ("backbone", "swin_large")):
        assert bench.run_experiments(argparse.Namespace(**dict(base, **{k: v})), 1) is None
    monkeypatch.setenv("LD_PRELOAD", "/opt/rocm/lib/librocprofiler-sdk-tool.so")
    assert bench.run_experiments(argparse.Namespace(**base), 1) is None
    monkeypatch.delenv("LD_PRELOAD")
    monkeypatch.setattr(experiments_r05, "main", lambda: 1 / 0)
    assert "ZeroDivisionError" in bench.run_experiments(argparse.Namespace(**base), 1)["error"]


def test_experiment_parent_survives_failing_and_hanging_arms(monkeypatch, tmp_path):
    from tools import experiments_r05 as X

    def fake(args, env, timeout):
        if args[0] == "--fwd":
            return {"model": {"quad_us": 150.0, "cell_us": 110.0}}
        if args[0] == "--swin":
            return {"window_attention_module_fwd_bwd": {"ops_us": 900.0, "fused_us": 300.0}}
        if args[0] == "--stp":
            return {"standard_us": 400.0, "sample_then_project_us": 380.0}
        if args[0] == "--records":
            return {"product": {"fwd_us": 190.0, "bwd_us": 540.0}, "records": {"fwd_us": 220.0, "bwd_us": 350.0, "equal_bits": False, "accepted": True, "finite": True},
                    "records_swap": {"fwd_us": 220.0, "bwd_us": 330.0, "equal_bits": False, "accepted": False, "finite": True}}
        k = int(args[1])
        if k == 3:
            return {"error": "timed out after 75 s (child killed)"}
        d = [[1, 2], [3, 4], [5, 6]] if k != 2 else [[1, 2], [3, 4], [5, 7]]
        return {"b0": {"digest": d, "finite": True, "us": 500.0 - k}, "fused": {"digest": d, "finite": True, "us": 540.0 - k}}
    steps = []
    monkeypatch.setattr(X, "run_child", fake)
    monkeypatch.setattr(X, "run_step_child", lambda flags, env, timeout: steps.append(flags) or {"ms_per_step": 35.0, "roofline_frac": 0.14})
    monkeypatch.setattr(X, "ABLATION_LIB", os.path.join(ROOT, "bench.py"))
    rep = X.main()
    # (the records route was bit-equal in the fake child: the whole step is measured with it, in a child of its own)
    assert steps == [["--set", "msda.records_route=1", "--set", "msda.records_swap=0"]] and rep["train_step_with_records_route"]["roofline_frac"] == 0.14
    arms = rep["encoder_backward_arms"]
    names = [n for n, _ in X.ARMS]
    assert arms[names[0]]["b0"]["equal_bits"] and arms[names[1]]["fused"]["equal_bits"]
    assert not arms[names[2]]["b0"]["equal_bits"] and "error" in arms[names[3]]
    assert rep["encoder_forward_cell"]["model"]["cell_us"] == 110.0 and "swin_routes" in rep
    assert rep["encoder_records_route"]["records"]["accepted"]
    assert "digest" not in json.dumps(rep)
    # a real child that produces nothing (here: no GPU) is an error entry, not an exception
    monkeypatch.undo()
    out = X.run_child(["--fwd"], dict(os.environ), 120)
    assert "error" in out


def test_promotion_report_reads_an_experiments_object(tmp_path, capsys):
    """tools/promote_r05.py on a synthetic bench line: bit-equal + faster -> PROMOTE with the place to edit, differing bits ->
    REJECT, slower -> KEEP OFF, a child that timed out -> SKIP"""
    from tools import promote_r05 as P
    rep = {"experiments": {
        "encoder_backward_arms": {
            "default": {"b0": {"us": 540.0, "equal_bits": True, "finite": True}, "fused": {"us": 545.0, "equal_bits": True, "finite": True}},
            "cell 3": {"b0": {"us": 470.0, "equal_bits": True, "finite": True}, "fused": {"us": 480.0, "equal_bits": True, "finite": True}},
            "cell 2": {"b0": {"us": 450.0, "equal_bits": False, "finite": True}, "fused": {"us": 455.0, "equal_bits": False, "finite": True}},
            "patch multi": {"error": "timed out after 45 s (child killed)"}},
        "encoder_records_route": {
            "product": {"fwd_us": 190.0, "bwd_us": 545.0},
            "cell_forward": {"fwd_us": 170.0, "equal_bits": True, "out_max_diff_rel_to_max": 0.004},
            "records": {"fwd_us": 230.0, "bwd_us": 350.0, "equal_bits": False, "accepted": True, "finite": True, "far_flag": 0, "records_MB": 409.3},
            "records_swap": {"fwd_us": 230.0, "bwd_us": 330.0, "equal_bits": False, "accepted": False, "finite": True, "far_flag": 0, "records_MB": 409.3}},
        "encoder_forward_cell": {"model": {"quad_us": 160.0, "cell_us": 165.0, "max_diff_rel_to_max": 0.004, "non_finite": 0}},
        "decoder_cross_attention_sample_then_project": {"standard_us": 400.0, "sample_then_project_us": 300.0, "rel_l2_out": 0.004,
                                                        "rel_l2_d_src": 0.006, "rel_l2_d_value_proj_weight": 0.005},
        "train_step_with_records_route": {"ms_per_step": 34.9, "roofline_frac": 0.146, "mean_launch_us": 350.0, "roofline_kernel": "msda_records"},
        "swin_routes": {"error": "not started: time budget used up"}},
        "ms_per_step": 36.1, "roofline": {"frac": 0.094, "mean_launch_us": 542.0}}
    path = tmp_path / "line.json"
    path.write_text("[bench] some log line\n" + json.dumps(rep) + "\n")
    rows = {name: (verdict, where) for verdict, name, _, where in P.decide(P.load(str(path)))}
    assert rows["cell 3"][0] == "PROMOTE" and "kCellMode" in rows["cell 3"][1]
    assert rows["cell 2"][0] == "REJECT" and rows["patch multi"][0] == "SKIP"
    assert rows["records route (records)"][0] == "PROMOTE" and "records_route = True" in rows["records route (records)"][1]
    assert rows["records route (records_swap)"][0] == "REJECT"
    assert rows["train step with the records route"][0] == "PROMOTE"
    assert rows["cell forward, B0 signature (model locations)"][0] == "KEEP OFF"
    assert rows["decoder cross-attention: sample, then project"][0] == "PROMOTE" and rows["Swin routes"][0] == "SKIP"
    sys.argv = ["promote_r05.py", str(path)]
    P.main()
    assert "PROMOTE   records route (records)" in capsys.readouterr().out


@pytest.mark.parametrize("script", ["gpu_first_r05.sh", "gpu_quick_r05.sh", "gpu_reopen_r05.sh", "gpu_final_r03.sh", "gpu_ab.sh"])
def test_gpu_session_scripts_parse_and_name_existing_files(script):
    """the scripts of the next GPU session cannot be run here; at least they must parse and every repository file they name
    (tools/*.py, tests/*.py, bench.py) must exist"""
    import re
    import subprocess
    path = os.path.join(ROOT, "tools", script)
    assert subprocess.run(["bash", "-n", path], capture_output=True).returncode == 0
    text = open(path).read()
    for rel in set(re.findall(r"\b((?:tools|tests)/[\w/]+\.(?:py|sh))\b", text)) | ({"bench.py"} if "bench.py" in text else set()):
        assert os.path.exists(os.path.join(ROOT, rel)), f"{script} names {rel}, which does not exist"
