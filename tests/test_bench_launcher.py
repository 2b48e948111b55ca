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


def _load_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_headline_is_the_last_line_and_small():
    """Round 5's single 23.5 KB line was cut by the driver's stdout tail (`parsed: null`).  Feed the line builder that
    very result set (profiles/r05_bench_line.json, prose and all, plus a failing secondary): the last line must be the
    headline, parse, stay under 4 KB and carry `roofline` + `cpu_baseline`; every secondary gets its own short line."""
    import json
    bench = _load_bench()
    with open(os.path.join(ROOT, "profiles", "r05_bench_line.json")) as fh:
        full = json.load(fh)
    secondary = full.pop("secondary") + [{"workload": "c3", "error": "RuntimeError: " + "x" * 5000}]
    full["config"]["step"] = "prose " * 400  # whatever a workload writes there must not reach the headline
    full["detail"] = "gpurun_out/bench_detail.json"
    lines = bench.output_lines(full, secondary)
    assert len(lines) == len(secondary) + 1
    head = lines[-1]
    assert "\n" not in head and len(head) < 4096 == bench.HEADLINE_LIMIT
    parsed = json.loads(head)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in parsed, key
    assert "secondary" not in parsed
    assert parsed["value"] == full["value"] and parsed["ms_per_step"] == full["ms_per_step"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in parsed["roofline"], key
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in parsed["cpu_baseline"], key
    assert parsed["config"]["workload"].startswith("DiffPool") and "step" not in parsed["config"]
    for text, sec in zip(lines[:-1], secondary):
        one = json.loads(text)
        assert "\n" not in text and len(text) < 2048, (sec["workload"], len(text))
        assert one["secondary"] == sec["workload"]
        if "error" not in sec:
            assert one["roofline"]["frac"] == sec["roofline"]["frac"] and one["ms_per_step"] == sec["ms_per_step"]
