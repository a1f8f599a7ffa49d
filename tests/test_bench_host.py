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
    # `--dp-schedule auto`: the line says which schedule ran and why (train.choose_dp_schedule's verdict)
    auto = bench.parallelism_text(world, True, True, True, schedule="auto", reason="captured collectives replay and its first step equals the flat schedule's to 0.001")
    assert auto.startswith(f"dp{world} (bf16 RCCL gradient all-reduce in buckets") and auto.endswith("; schedule auto: captured collectives replay and its first step equals the flat schedule's to 0.001")
    fell = bench.parallelism_text(world, True, False, True, schedule="auto", reason="the overlapped step differs from the flat one (loss 1 vs 2, gradient norm 3 vs 4)")
    assert fell.startswith(f"dp{world} (one flat") and "schedule auto: the overlapped step differs" in fell
    assert "schedule" not in bench.parallelism_text(world, True, False, False, schedule="auto")      # no process group: nothing to choose


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
    """~15 environment-variable A/B switches were read inside the product package.
    What is left is configuration of the libraries underneath (MIOpen / hipBLASLt tuning tables).  Another build of the C-ABI
    library is selected by an explicit `_lib.use_library(path)` call; RLIPV2_LIB_PATH is read by the tools package only."""
    import re
    allowed = {"RLIPV2_TUNED_MIOPEN", "MIOPEN_USER_DB_PATH", "XDG_CACHE_HOME", "RLIPV2_TUNED_GEMM_TABLE", "RLIPV2_TUNED_GEMMS"}
    pkg = os.path.join(ROOT, "rlipv2_amd")
    seen = set()
    for f in os.listdir(pkg):
        if f.endswith(".py"):
            seen |= set(re.findall(r"environ(?:\.get)?[\(\[]\s*\"([A-Z0-9_]+)\"", open(os.path.join(pkg, f)).read()))
    assert seen <= allowed, seen - allowed


def test_use_library_is_explicit_and_refuses_a_second_build(monkeypatch, tmp_path):
    from rlipv2_amd import _lib
    import tools
    monkeypatch.setattr(_lib, "LIB_PATH", _lib.LIB_PATH)
    monkeypatch.setattr(_lib, "CPU_LIB_PATH", _lib.CPU_LIB_PATH)
    monkeypatch.setattr(_lib, "_lib", None)
    _lib.use_library(str(tmp_path / "other.so"))
    assert _lib.LIB_PATH == str(tmp_path / "other.so")
    monkeypatch.setattr(_lib, "_lib", object())                       # "already loaded"
    with pytest.raises(RuntimeError, match="already loaded"):
        _lib.use_library(str(tmp_path / "third.so"))
    _lib.use_library(str(tmp_path / "other.so"))                      # the same path again is fine
    # the tools package is what reads the environment variable
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setenv("RLIPV2_LIB_PATH", str(tmp_path / "from_env.so"))
    tools.apply_library_overrides()
    assert _lib.LIB_PATH == str(tmp_path / "from_env.so")


def _ns(**kw):
    import argparse
    base = dict(experiments=True, no_cpu_baseline=False, backbone="resnet50", dtype="bf16", batch=4, overrides=[], padded=False,
                var_targets=False)
    return argparse.Namespace(**dict(base, **kw))


def test_experiments_leg_is_opt_in_and_only_next_to_the_full_default_run(monkeypatch):
    """the A/B table of the unmeasured kernel arms is opt-in (`--experiments`) and only runs next to the default 1-GPU evidence
    configuration -- never under a profiler, with overrides, on another configuration, or on several ranks"""
    assert bench.experiments_applicable(_ns(), 1)
    assert not bench.experiments_applicable(_ns(), 2)
    for k, v in (("experiments", False), ("no_cpu_baseline", True), ("batch", 8), ("overrides", ["a.b=1"]), ("backbone", "swin_large")):
        assert not bench.experiments_applicable(_ns(**{k: v}), 1)
    monkeypatch.setenv("LD_PRELOAD", "/opt/rocm/lib/librocprofiler-sdk-tool.so")
    assert not bench.experiments_applicable(_ns(), 1)
    monkeypatch.delenv("LD_PRELOAD")
    assert bench.experiments_leg(_ns(experiments=False), 1) is None
    # default command line: off
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'ap.add_argument("--experiments", action="store_true"' in src


def _leg(tmp_path, body, budget_s=20):
    script = tmp_path / "leg.py"
    script.write_text(body)
    out = tmp_path / "experiments_last.json"
    return bench.experiments_leg(_ns(), 1, out_path=str(out), cmd=[sys.executable, str(script)], budget_s=budget_s), out


def test_experiments_leg_records_the_object_and_survives_a_leg_that_raises_or_hangs(tmp_path, capfd):
    rep, out = _leg(tmp_path, "import json; print('noise'); print(json.dumps({'encoder_backward_arms': {}}))")
    assert rep == {"encoder_backward_arms": {}} and json.load(open(out)) == rep
    assert "EXPERIMENTS " + json.dumps(rep) in capfd.readouterr().err
    rep, out = _leg(tmp_path, "raise SystemExit(3)")
    assert "rc 3" in rep["error"] and json.load(open(out)) == rep
    # a leg that hangs past its budget -- together with a grandchild that ignores SIGTERM -- is killed as a process group
    (tmp_path / "grandchild.py").write_text(
        "import os, signal, time\n"
        "signal.signal(signal.SIGTERM, signal.SIG_IGN)\n"
        "open(%r, 'w').write(str(os.getpid()))\n"
        "time.sleep(600)\n" % str(tmp_path / "grandchild.pid"))
    body = ("import subprocess, sys, time\n"
            "subprocess.Popen([sys.executable, %r])\n"
            "time.sleep(600)\n" % str(tmp_path / "grandchild.py"))
    import time
    t0 = time.time()
    rep, out = _leg(tmp_path, body, budget_s=3)
    assert "timed out" in rep["error"] and time.time() - t0 < 30 and json.load(open(out)) == rep
    pid = int(open(tmp_path / "grandchild.pid").read())
    time.sleep(0.5)
    assert not os.path.exists(f"/proc/{pid}") or open(f"/proc/{pid}/stat").read().split()[2] == "Z"
    capfd.readouterr()


_EMIT_SNIPPET = """
import argparse, json, os, signal, sys
sys.path.insert(0, %r)
import bench
class Lib:
    msda_pick_variant = staticmethod(lambda *a: 0)
    msda_variant_name = staticmethod(lambda v: b"quad")
args = argparse.Namespace(batch=4, steps=2, warmup=1, dtype="bf16", no_cpu_baseline=True, experiments=True, backbone="resnet50",
                          overrides=[], padded=False, var_targets=False)
kern = {"enc_bwd_fused": {"ms": 1.0, "n": 2, "dims": (4, 22223, 8, 32, 4, 22223, 4), "code": 2, "bwd": True, "variant": "cell+patch",
                          "bytes": 409600000, "operand_bytes": 300000000}}
bench.emit(args, 1, 0.08, kern, Lib, workload_text="train_step: test", cpu_calls=None, parallelism="dp1")
%s
"""


@pytest.mark.parametrize("after", ["os.kill(os.getpid(), signal.SIGKILL)", "raise RuntimeError('experiments blew up')",
                                   "os.kill(os.getpid(), signal.SIGSEGV)"])
def test_the_stdout_line_is_out_before_anything_else_can_go_wrong(after):
    """whatever happens after emit() -- the process killed, an exception, a fault -- the driver's pipe
    already holds the complete line (printed and flushed before the opt-in experiments leg is even considered)"""
    import subprocess
    r = subprocess.run([sys.executable, "-c", _EMIT_SNIPPET % (ROOT, after)], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["ms_per_step"] == pytest.approx(40.0) and line["roofline"]["frac"] > 0 and "experiments" not in line


def test_main_prints_the_line_before_the_experiments_leg_in_source_order():
    """main(): emit(...) comes before experiments_leg(...), and emit() itself no longer knows the leg"""
    import inspect
    src = inspect.getsource(bench.main)
    assert 0 < src.index("emit(args, world, elapsed, kern, lib") < src.index("experiments_leg(args, world)")
    assert "experiments" not in inspect.signature(bench.emit).parameters
    code = "\n".join(ln.split("#")[0] for ln in inspect.getsource(bench.emit).splitlines())      # (comments may mention it)
    assert "experiments" not in code and code.rstrip().endswith("return line")


def _routes_child(tmp_path, body, timeout=20, backbone="resnet50"):
    script = tmp_path / "child.py"
    script.write_text(body)
    return bench.routes_verdicts_from_child([], backbone, cmd=[sys.executable, str(script)], timeout=timeout)


def test_route_self_check_child_verdicts_and_failures(tmp_path, capfd):
    """`--host-routes auto`: the self-check runs in a child before the parent touches the GPU.  A child that segfaults, aborts,
    hangs, exits non-zero or prints a malformed verdict leaves every route off; Swin routes are never exercised on R50."""
    r50 = ["residual_gradient_in_gemm", "one_launch_box_head"]
    assert bench.applicable_routes("resnet50") == r50
    assert bench.applicable_routes("swin_large") == list(routes.GPU_ONLY_ROUTES)
    good = {"residual_gradient_in_gemm": "on", "one_launch_box_head": "off (self-check failed: loss 1 vs 2)"}
    v = _routes_child(tmp_path, "import sys, json; print('[routes] step noise 0.001', file=sys.stderr); print('ROUTES ' + json.dumps(%r))" % good)
    assert v["residual_gradient_in_gemm"] == "on" and v["one_launch_box_head"].startswith("off (self-check failed")
    assert v["fused_wide_layer_norm"].startswith("off (not applicable") and v["fused_window_attention"].startswith("off (not applicable")
    assert list(v) == list(routes.GPU_ONLY_ROUTES)
    assert "[routes] step noise" in capfd.readouterr().err
    for body, word in (("import os, signal; print('ROUTES {}'); os.kill(os.getpid(), signal.SIGSEGV)", "SIGSEGV"),
                       ("import os; os.abort()", "SIGABRT"),
                       ("import time; time.sleep(600)", "timed out"),
                       ("import sys; print('boom', file=sys.stderr); sys.exit(7)", "exit code 7; boom"),
                       ("print('ROUTES {not json')", "no well-formed verdict"),
                       ("import json; print('ROUTES ' + json.dumps({'residual_gradient_in_gemm': 'on'}))", "no well-formed verdict"),
                       ("import json; print('ROUTES ' + json.dumps({'residual_gradient_in_gemm': 'on', 'one_launch_box_head': 'yes'}))", "no well-formed verdict"),
                       ("print('nothing')", "no well-formed verdict")):
        v = _routes_child(tmp_path, body, timeout=3)
        assert all(x.startswith("off") for x in v.values()), (body, v)
        assert word in v["residual_gradient_in_gemm"], (body, v)
    v = bench.routes_verdicts_from_child([], "resnet50", cmd=["/nonexistent/python"], timeout=3)
    assert all(x.startswith("off") for x in v.values()) and "not started" in v["one_launch_box_head"]


def test_route_verdicts_are_applied_exactly(monkeypatch):
    try:
        out = bench.apply_route_verdicts({"residual_gradient_in_gemm": "on", "one_launch_box_head": "off (self-check child: signal SIGSEGV)",
                                          "fused_wide_layer_norm": "off (not applicable: x)", "fused_window_attention": "ON"}, 1, "cpu")
        assert routes.state() == {"residual_gradient_in_gemm": True, "one_launch_box_head": False, "fused_wide_layer_norm": False,
                                  "fused_window_attention": False}
        assert out["residual_gradient_in_gemm"] == "on" and "SIGSEGV" in out["one_launch_box_head"]
        out = bench.apply_route_verdicts(None, 1, "cpu")
        assert not any(routes.state().values()) and all(v.startswith("off") for v in out.values())
    finally:
        routes.set_all(False)


def test_parent_runs_no_route_self_check_itself():
    """the bench process never calls routes.validate (only the --routes-child process does), and starts the child before its own
    first GPU-initialising call"""
    import inspect
    assert "routes.validate" not in inspect.getsource(bench.run_train_step_bench)
    assert "routes.validate" in inspect.getsource(bench.routes_child_main)
    src = inspect.getsource(bench.main)
    assert src.index("routes_verdicts_from_child(") < src.index("if not torch.cuda.is_available():") < src.index("torch.cuda.set_device(local_rank)")


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
    rep["experiments"]["encoder_backward_arms"]["patch cellg"] = {"b0": {"us": 470.0, "accepted": True, "finite": True},
                                                                   "fused": {"us": 500.0, "accepted": True, "finite": True, "equal_bits": True}}
    rep["experiments"]["uniform_location_arms"] = {
        "default": {"uniform": {"us": 1340.0, "accepted": True}, "model": {"us": 545.0, "accepted": True}},
        "far return": {"uniform": {"us": 800.0, "accepted": True}, "model": {"us": 546.0, "accepted": True}},
        "far return + queue-fed fallback": {"uniform": {"us": 790.0, "accepted": False}, "model": {"us": 540.0, "accepted": True}}}
    path = tmp_path / "line.json"
    path.write_text("[bench] some log line\n" + json.dumps(rep) + "\n")
    rows = {name: (verdict, where) for verdict, name, _, where in P.decide(P.load(str(path)))}
    assert rows["cell 3"][0] == "PROMOTE" and "kCellMode" in rows["cell 3"][1]
    assert rows["cell 2"][0] == "REJECT" and rows["patch multi"][0] == "SKIP"
    assert rows["patch cellg"][0] == "PROMOTE" and "grad_out_cells_kernel" in rows["patch cellg"][1]
    assert rows["uniform locations: far return"][0] == "PROMOTE" and rows["uniform locations: far return + queue-fed fallback"][0] == "REJECT"
    # bench.py --experiments leaves the object behind the word EXPERIMENTS on stderr: found as well
    log = tmp_path / "bench_stderr.txt"
    log.write_text("[bench] x\nEXPERIMENTS " + json.dumps(rep["experiments"]) + "\n")
    assert {n for _, n, _, _ in P.decide(P.load(str(log)))} >= {"cell 3", "patch cellg"}
    assert rows["records route (records)"][0] == "PROMOTE" and "records_route = True" in rows["records route (records)"][1]
    assert rows["records route (records_swap)"][0] == "REJECT"
    assert rows["train step with the records route"][0] == "PROMOTE"
    assert rows["cell forward, B0 signature (model locations)"][0] == "KEEP OFF"
    assert rows["decoder cross-attention: sample, then project"][0] == "PROMOTE" and rows["Swin routes"][0] == "SKIP"
    sys.argv = ["promote_r05.py", str(path)]
    P.main()
    assert "PROMOTE   records route (records)" in capsys.readouterr().out


@pytest.mark.parametrize("script", ["gpu_triage_r06.sh", "gpu_profiles_r06.sh", "gpu_ab.sh"])
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


def test_triage_families_cover_every_never_run_kernel_within_the_budget():
    """tools/gpu_triage_r06.py: every kernel the manifest records as never run on hardware belongs to exactly one family (one child
    process group + one verdict line each), the families' GPU timeouts add up to <= 20 minutes, the order is (default path after
    promotion) first, and every test file / tool a family names exists"""
    import re
    from tools import gpu_triage_r06 as T
    claimed, unclaimed = T.assign()
    assert unclaimed == [], unclaimed
    assert sum(len(v) for v in claimed.values()) == len(T.never_run_kernels()) > 0
    assert sum(f["timeout"] for f in T.FAMILIES) <= T.BUDGET_S == 1200
    flags = [f["default_after_promotion"] for f in T.FAMILIES]
    assert flags == sorted(flags, reverse=True)                      # promotable families first
    for f in T.FAMILIES:
        assert f["gpu"] and f["emu"], f["name"]
        for cmd, _ in f["gpu"] + f["emu"]:
            for tok in cmd:
                if re.fullmatch(r"(tests|tools)/[\w/]+\.py", tok):
                    assert os.path.exists(os.path.join(ROOT, tok)), (f["name"], tok)
        if f["kernels"]:
            assert claimed[f["name"]], f"{f['name']}: no never-run kernel matches {f['kernels']}"


def test_triage_survives_failing_hanging_and_crashing_children(tmp_path, monkeypatch, capsys):
    from tools import gpu_triage_r06 as T
    py = sys.executable
    fams = [dict(name="ok", row="a1", default_after_promotion=True, what="w", kernels=[], timeout=30,
                 gpu=[([py, "-c", "print('fine')"], {})], emu=[([py, "-c", "print('fine')"], {})]),
            dict(name="fails", row="a2", default_after_promotion=True, what="w", kernels=[], timeout=30,
                 gpu=[([py, "-c", "raise SystemExit(3)"], {}), ([py, "-c", "print('second child still runs')"], {})], emu=[]),
            dict(name="hangs", row="a2", default_after_promotion=False, what="w", kernels=[], timeout=2,
                 gpu=[([py, "-c", "import time; time.sleep(600)"], {})], emu=[]),
            dict(name="segfaults", row="f3", default_after_promotion=False, what="w", kernels=[], timeout=30,
                 gpu=[([py, "-c", "import os, signal; os.kill(os.getpid(), signal.SIGSEGV)"], {})], emu=[])]
    monkeypatch.setattr(T, "FAMILIES", fams)
    monkeypatch.setattr(T, "never_run_kernels", lambda: [])
    monkeypatch.setattr(T, "device_alive", lambda env, log: True)
    rc = T.main(["--out", str(tmp_path)])
    rep = json.load(open(tmp_path / "triage.json"))
    got = {r["family"]: r["verdict"] for r in rep["families"]}
    assert got == {"ok": "PASS", "fails": "FAIL", "hangs": "TIMEOUT", "segfaults": "FAIL"} and rc == 1
    assert len(rep["families"][1]["steps"]) == 2                      # a failing child does not stop its family
    lines = [ln for ln in capsys.readouterr().out.splitlines() if ln.startswith("TRIAGE ")]
    assert len(lines) == 4 and "family=hangs verdict=TIMEOUT" in lines[2]
    # a device that does not answer after a family: the rest is skipped, and said so
    monkeypatch.setattr(T, "device_alive", lambda env, log: "segfaults" not in log and "hangs" not in log)
    T.main(["--out", str(tmp_path)])
    got = [r["verdict"] for r in json.load(open(tmp_path / "triage.json"))["families"]]
    assert got == ["PASS", "FAIL", "DEVICE-LOST", "SKIPPED"]
    # the dry run uses the lane-model commands
    monkeypatch.setattr(T, "device_alive", lambda env, log: (_ for _ in ()).throw(AssertionError("no GPU in a dry run")))
    T.main(["--dry-run", "--only", "ok", "--out", str(tmp_path)])
    assert json.load(open(tmp_path / "triage_dry_run.json"))["families"][0]["verdict"] == "PASS"


def test_first_contact_tests_run_isolated_and_never_colour_the_suite():
    """tests/conftest.py: the GPU tests of a file that are marked `first_contact` (device code / host routes that have never executed
    on hardware) run together in ONE child pytest with a timeout -- a hang or a fault costs those tests, not the suite -- and are
    reported as XPASS / XFAIL: the suite goes on under -x and its exit code stays the product path's."""
    import subprocess

    def run(path, **env):
        return subprocess.run([sys.executable, "-m", "pytest", os.path.join("tests", path), "-q", "-x", "-rxX", "-p", "no:cacheprovider"],
                              capture_output=True, text=True, cwd=ROOT, timeout=600, env=dict(os.environ, **env))
    r = run("first_contact_probe.py")
    out = r.stdout
    assert r.returncode == 0, out[-3000:]
    assert "1 passed" in out and "2 xfailed" in out and "2 xpassed" in out, out[-1500:]
    assert "XPASS tests/first_contact_probe.py::test_probe_passes" in out and "XPASS tests/first_contact_probe.py::test_probe_parametrized[1]" in out
    assert "XFAIL tests/first_contact_probe.py::test_probe_fails" in out and "XFAIL tests/first_contact_probe.py::test_probe_parametrized[2]" in out
    # a child that hangs or dies: every test of the file is XFAIL with the reason, the parent session is green and goes on
    # (the test that had finished before the child went down keeps its result: read from the child's -v output)
    for mode in ("hang", "fault"):
        r = run("first_contact_probe_fatal.py", PROBE_MODE=mode)
        assert r.returncode == 0 and "2 xfailed" in r.stdout and "1 xpassed" in r.stdout, r.stdout[-2000:]
        assert "XPASS tests/first_contact_probe_fatal.py::test_fatal_a_passes" in r.stdout


def test_design_documents_stay_within_120_columns_and_the_index_names_existing_files():
    """DESIGN.md is an index over section files of at most 120 columns (tools/reflow_md.py), and every
    profiles/ / tests/ / tools/ file the index names exists"""
    import glob
    import re
    files = [os.path.join(ROOT, "DESIGN.md")] + sorted(glob.glob(os.path.join(ROOT, "docs", "design", "*.md")))
    assert len(files) == 9
    for f in files:
        worst = max(len(l) for l in open(f).read().split("\n"))
        assert worst <= 120, (f, worst)
    index = open(files[0]).read()
    for rel in set(re.findall(r"`((?:profiles|tests|tools|docs/design)/[\w./]+\.(?:md|txt|json|csv|py|sh))`", index)):
        assert os.path.exists(os.path.join(ROOT, rel)), rel


def test_the_real_route_child_end_to_end_on_the_cpu():
    """`bench.py --routes-child` as the parent starts it (same argv + the three child flags), on the CPU device (-1): the child
    builds the model and batch, runs routes.validate -- which answers "not applicable" off the GPU -- and prints its ROUTES line;
    the parent parses it.  The only part a GPU adds is the eager steps inside validate()."""
    v = bench.routes_verdicts_from_child(["--batch", "1", "--queries", "30", "--steps", "2"], "resnet50", rank=0, device_index=-1, timeout=600)
    assert list(v) == list(routes.GPU_ONLY_ROUTES)
    assert v["residual_gradient_in_gemm"].startswith("off (not applicable: the route exists on the GPU only")
    assert v["fused_window_attention"].startswith("off (not applicable: no such block in a resnet50 step")
