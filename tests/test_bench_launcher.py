"""bench.py's launcher contract on a box without GPUs: `--gpus N` must start N ranks or fail loudly (never run one
rank and report it as N), and a launcher-provided WORLD_SIZE that disagrees with --gpus is an error."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, capture_output=True,
                          text=True, timeout=300)


def test_gpus_n_without_enough_devices_fails_loudly():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("box has GPUs")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert "--gpus 2" in r.stderr and "GPU" in r.stderr
    assert r.stdout.strip() == ""  # no JSON line claiming n_gpus


def test_world_size_mismatch_is_an_error():
    r = _run(["--gpus", "4"], env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0
    assert "WORLD_SIZE=2" in (r.stderr + r.stdout)
