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
    assert abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-9 and "traffic" in ro and "traffic_source" in ro
    assert "HIP events" in ro["frac_source"]                   # (the default SHAPE also carries frac_rocprof from the tracked summary)
    assert d["status_nonzero_trajectories"] == 0 and "trunc_edge_trajectories" in d      # failures (bits 0-3) and the informational bit apart
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    assert cb["single_thread_value"] > 0 and cb["gru_half_torch_cpu"]["value"] > 0
    assert abs(d["value"] - 4096 * 20 * 2 / (d["ms_per_step"] * 2e-3)) / d["value"] < 1e-6
    # the metric's second half: state l-inf of the timed batch vs the CPU reference, outside the timed region
    pa = d["parity"]
    # ... over EVERY (trajectory, timestep) of the timed batch, the oracle run doubling as the all-cores baseline
    assert pa["trajectories"] == 4096 and pa["whole_tensor"] is True and pa["timesteps_each"] == 20
    assert pa["state_linf"] < 1e-4 and pa["gru_linf"] < 1e-4 and pa["ok"] is True
    assert "TIMED batch" in cb["sample"]
    # the second Q / R set of SURVEY 8(d) rides along (Q_R.pkl values with R[0:3] = 1e-4)
    sn = d["second_noise_set"]
    assert "Q_R.pkl" in sn["noise"] and sn["parity"]["ok"] is True and sn["parity"]["trajectories"] == 4096 and sn["value"] > 0
    assert "settings.py" in d["config"]["noise"]
    assert cb["reference_python_steps_per_s"] == 3.05e3
    assert d["rccl_world_size"] == 1 and d["rank_devices"][0]["device"] == 0
    # B = 4096 takes the single fused kernel's smallest tile (16 trajectories per CU, four wavefronts per tile: round 6; the two-kernel
    # path until round 5): the line names the kernel that actually ran, and its MFMA instruction
    assert d["kernels"]["fused"]["kernel"] == "fused_kf_gru_kernel_v3<4>" and "kf" not in d["kernels"]
    assert ro["kernel"].startswith("fused_kf_gru_kernel_v3<4>") and "v_mfma_f32_16x16x4_f32" in ro["kernel"] and ro["bound"] == "mfma"


def _has_roofline_and_baseline(d):
    ro, cb = d["roofline"], d["cpu_baseline"]
    assert ro["bound"] in ("hbm", "mfma") and ro["unit"] in ("GB/s", "TFLOP/s") and "traffic" in ro and "traffic_source" in ro
    assert abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-9 and ro["achieved"] > 0
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb and "unit" in cb


def test_other_modes_print_one_line_with_roofline_and_cpu_baseline():
    k = _run(["--mode", "kf", "--batch", "4096", "--seq", "50", "--steps", "2", "--warmup", "1", "--cpu-seconds", "1"])
    _has_roofline_and_baseline(k)
    assert k["roofline"]["kernel"] == "kf_run_rows2_kernel" and "instruction issue" in k["roofline"]["limiter"]     # the kernel that RAN
    assert k["parity"]["state_linf"] < 1e-4 and "gru_linf" not in k["parity"]
    d = _run(["--mode", "train", "--steps", "2", "--warmup", "1", "--cpu-seconds", "2"])
    _has_roofline_and_baseline(d)
    assert d["config"]["batch_per_gpu"] == 8192 and d["grad_bucket_bytes"] == 422424 * 4
    assert d["roofline"]["bound"] == "mfma" and set(d["kernels"]) >= {"gru_layer", "train_sweep", "train_dw"}
    m = _run(["--mode", "mpc", "--batch", "512", "--seq", "10", "--steps", "1", "--warmup", "1", "--cpu-seconds", "2"])
    _has_roofline_and_baseline(m)
    assert m["value"] > 0 and m["status_nonzero_trajectories"] == 0 and m["kernels"]["mpc"]["launches_per_step"] == 1
    assert m["kernels"]["mpc"]["kernel"] == "kf_mpc_persistent_kernel"        # one launch for all T steps, no per-step launches
    # the QP line's bound: algorithmic fp64 flops of the iterations taken against the fp64 vector peak
    assert m["roofline"]["unit"] == "TFLOP/s" and m["roofline"]["peak"] == 78.6 and "fp64" in m["roofline"]["pipe"] and m["roofline"]["algorithmic_flops_per_pass"] > 0
    assert "trunc_edge_trajectories" in m
    f = _run(["--mode", "full", "--steps", "2", "--warmup", "1", "--cpu-seconds", "1"])
    _has_roofline_and_baseline(f)
    assert f["config"]["frames"] == 1024 and f["unit"] == "frames/s"
    w = _run(["--mode", "windows", "--batch", "2048", "--steps", "2", "--warmup", "1", "--cpu-seconds", "1"])
    _has_roofline_and_baseline(w)
    assert w["unit"] == "windows/s" and w["config"]["gru_timesteps_per_output"] == 10
    # the row-stream form: no window tensor, layer 0's input projection once per row; the roofline counts ITS flops
    assert "os_gru_forward_windows" in w["config"]["form"] and w["config"]["rows_per_gpu"] == 2048 + 9
    assert w["roofline"]["algorithmic_flops_per_pass"] < w["roofline"]["flops_of_the_materialised_form"]
    wm = _run(["--mode", "windows", "--materialise", "--batch", "2048", "--steps", "2", "--warmup", "1", "--cpu-seconds", "0"])
    assert "materialised" in wm["config"]["form"] and wm["roofline"]["algorithmic_flops_per_pass"] == wm["roofline"]["flops_of_the_materialised_form"]


def test_scaling_strong_runs_the_global_batch_and_reports_it():
    """--scaling strong at N = 1: --batch is the global batch, the line says "strong", carries the global checksum, and `value` counts the
    global batch once (the N-GPU form shards the same samples: tests/test_bench_cpu.py checks the shards, test_gpu_world_shared.py runs two)."""
    d = _run(["--scaling", "strong", "--batch", "8192", "--seq", "50", "--steps", "3", "--warmup", "1", "--cpu-seconds", "0",
              "--parity-samples", "256", "--no-second-noise"])
    assert d["scaling"] == "strong" and d["config"]["global_batch"] == 8192 and d["config"]["batch_per_gpu"] == 8192
    assert abs(d["value"] - 8192 * 50 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    cs = d["global_checksum"]
    assert cs["trajectories"] == 8192 and cs["sum_out"] > 0 and cs["sum_x_final"] == cs["sum_x_final"]
    assert d["kernels"]["fused"]["kernel"] == "fused_kf_gru_kernel_v3<2>" and d["parity"]["ok"] is True


def test_split_bf16_line_on_the_h128_layer_kernel():
    """--split-bf16 with hidden 128: the layer launches run on gru_layer_bf16_kernel (opt-in), the line says so (dtype, kernel, bf16
    peak) and its parity block keeps the unchanged 1e-5 GRU bar."""
    d = _run(["--split-bf16", "3", "--hidden", "128", "--layers", "2", "--batch", "32768", "--seq", "4", "--steps", "2", "--warmup", "1",
              "--cpu-seconds", "1", "--parity-samples", "2048"])
    assert d["dtype"].startswith("bf16x3") and d["kernels"]["gru_layer"]["kernel"] == "gru_layer_bf16_kernel<3>"
    ro = d["roofline"]
    assert ro["bound"] == "mfma" and "bf16" in ro["kernel"] and ro["peak"] > 2000 and ro["executed_bf16_TFLOPs"] > ro["achieved"]
    assert d["parity"]["gru_bar"] == 1e-5 and d["parity"]["gru_linf"] < 1e-5 and d["parity"]["ok"] is True
