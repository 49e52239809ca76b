"""GPU, needs >= 2 devices (skipped on a 1-GPU box): SURVEY.md 8(e) determinism check and the bench's own multi-GPU launch.
The world-size-2 logic is also covered on CPU with gloo (tests/test_dist_cpu.py, tests/test_bench_cpu.py)."""
import json
import os
import subprocess
import sys

import pytest
import torch

from conftest import ROOT

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")]

ENV = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
ENV.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def _json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_one_gpu_full_batch_equals_two_gpus_half_batches():
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                        "127.0.0.1", "--master-port", "29631", os.path.join(ROOT, "tests", "multi_gpu_worker.py")],
                       capture_output=True, text=True, timeout=900, env=ENV)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _json_line(r.stdout)
    assert d["world"] == 2 and d["backend"] == "nccl"
    assert d["infer_state_max_abs_diff"] == 0.0                     # lane placement does not change a trajectory's filter
    assert d["infer_out_max_abs_diff"] < 1e-6 and d["infer_status_nonzero"] == 0
    assert d["train_grad_max_abs_diff"] < 2e-5 * d["train_grad_scale"] + 1e-9      # fp32 reduction-order noise
    assert d["train_replica_weight_max_abs_diff"] == 0.0            # identical replicas after the update
    assert d["grad_bucket_bytes"] == 1689696


@pytest.mark.parametrize("mode", ["fused", "train"])
def test_bench_gpus_2_launches_itself(mode):
    args = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--cpu-seconds", "0", "--mode", mode]
    if mode == "fused":
        args += ["--batch", "8192", "--seq", "20"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=900, env=ENV)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["rccl_world_size"] == 2 and d["backend"] == "nccl"
    assert sorted(x["device"] for x in d["rank_devices"]) == [0, 1]
    if mode == "train":
        assert d["allreduce_us"] > 0 and d["grad_bucket_bytes"] == 1689696
    else:
        assert d["parity"]["state_linf"] < 1e-4
