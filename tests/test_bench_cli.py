"""bench.py's launch contract (no GPU needed): `--gpus N` must never silently run one rank."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, *args):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, env=env,
                          timeout=300)


def test_gpus_n_without_a_launcher_starts_its_own_ranks_or_refuses():
    """No RANK in the environment: bench.py launches torch.distributed.run as a child process before touching the GPU
    (reference: torch.distributed.launch --nproc_per_node, scripts/RLIP_ParSeDA/*.sh).  On a box with fewer GPUs than
    asked for it must refuse with a non-zero exit code, not fall back to one rank."""
    r = _run({}, "--gpus", "2", "--steps", "1", "--warmup", "0")
    assert r.returncode != 0
    assert "--gpus 2" in r.stderr and "GPU" in r.stderr
    assert '"n_gpus"' not in r.stdout


def test_world_size_must_equal_gpus():
    r = _run({"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"}, "--gpus", "8", "--steps", "1")
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr
    r = _run({"RANK": "0", "WORLD_SIZE": "4", "LOCAL_RANK": "0"}, "--gpus", "2", "--steps", "1")
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr


def test_self_launch_relays_the_launcher_exit_code():
    """RLIPV2_SINGLE_DEVICE=1 lets the self-launch proceed on a box without enough GPUs; here (no GPU at all) the ranks
    stop with bench.py's own "needs a GPU" message and the launcher's failure code comes back."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("CPU-only check of the relay")
    r = _run({"RLIPV2_SINGLE_DEVICE": "1"}, "--gpus", "2", "--steps", "1", "--warmup", "0")
    assert r.returncode != 0
    assert "torch.distributed.run" in r.stderr and "needs a GPU" in r.stderr
