"""The driver's contract for bench.py (one JSON line on stdout with the agreed keys), checked by running the
script the way the driver does (1 GPU, few steps).  GPU test: bench.py has no CPU path."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_keys():
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                           "--host-routes", "off"],  # (the self-check child of `auto` is exercised by the 2-rank test: ~1 min saved here)
                          capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [l for l in proc.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, proc.stdout[-2000:]
    d = json.loads(lines[0])
    baseline = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"] == baseline["metric"] and d["unit"] == "images/s"
    assert abs(d["per_gpu"] - d["value"] / d["n_gpus"]) < 1e-2
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "bf16" and d["data"] == "synthetic"
    assert d["value"] > 0 and d["ms_per_step"] > 0
    assert abs(d["value"] - 4 / (d["ms_per_step"] / 1e3)) / d["value"] < 1e-3          # whole-job images / s
    assert isinstance(d["config"]["workload"], str) and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert r["peak"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["traffic"] is None or r["traffic"] > 0
    c = d["cpu_baseline"]
    assert c["value"] > 0 and c["cores"] >= 1 and c["kind"] in ("reference", "port") and isinstance(c["sample"], str)
