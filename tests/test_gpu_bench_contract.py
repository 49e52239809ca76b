"""GPU: bench.py prints ONE JSON line with the contract's keys (small shapes so that it takes seconds)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_default_mode_contract():
    d = _run(["--batch", "4096", "--seq", "20", "--steps", "2", "--warmup", "1", "--cpu-seconds", "1"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    ro = d["roofline"]
    assert ro["bound"] in ("hbm", "mfma") and ro["unit"] in ("GB/s", "TFLOP/s")
    assert abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-9 and "traffic" in ro
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    assert cb["single_thread_value"] > 0 and cb["gru_half_torch_cpu"]["value"] > 0
    assert abs(d["value"] - 4096 * 20 * 2 / (d["ms_per_step"] * 2e-3)) / d["value"] < 1e-6


def test_other_modes_print_one_line():
    assert _run(["--mode", "kf", "--batch", "4096", "--seq", "20", "--steps", "2", "--warmup", "1", "--cpu-seconds", "0"])["value"] > 0
    d = _run(["--mode", "train", "--steps", "2", "--warmup", "1"])
    assert d["config"]["batch_per_gpu"] == 8192 and d["grad_bucket_bytes"] == 422424 * 4
    m = _run(["--mode", "mpc", "--batch", "512", "--seq", "10", "--steps", "1", "--warmup", "1"])
    assert m["value"] > 0 and m["status_nonzero_trajectories"] == 0 and m["kernels"]["mpc"][1] == 10
