"""GPU, ONE device: the world > 1 code paths of SURVEY.md 8(e) executed with real device tensors.

The driver's 8-GPU run is one shot and the development box has one MI355X, where RCCL refuses two ranks on the same device.
So the ranks here are fresh child processes (started by torch.distributed.run from this process: children, never an exec of
a process that has touched the GPU), every one on cuda:0 (`OS_SHARE_GPU=1`: device = local_rank % device_count), the process
group runs over gloo and `optistate_amd.train.all_reduce_` stages the flat bucket through host memory.  Everything else is the
code the 8-GPU job runs: `tests/multi_gpu_worker.py` (sharded `fused_run` gathered = the full-batch run; `DataParallelTrainer`
with the split all-reduce behind `os_gru_backward_mark`'s event = the single-process full-batch gradient, replicas identical
after the fused Adam) and `bench.py --gpus N --share-gpu` for both modes.  Reference step: gru/gru_train.py:232-251.
"""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

ENV = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
ENV.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def _json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def _run_worker(world, port, train_batch=None):
    env = dict(ENV, OS_SHARE_GPU="1")
    if train_batch is not None:
        env["OS_WORKER_TRAIN_BATCH"] = str(train_batch)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "multi_gpu_worker.py")],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    return _json_line(r.stdout)


def _check(d, world):
    assert d["world"] == world and d["backend"] == "gloo"
    assert len({x["pid"] for x in d["rank_devices"]}) == world                 # separate rank processes ...
    assert {x["device"] for x in d["rank_devices"]} == {0}                     # ... on one device
    assert d["infer_state_max_abs_diff"] == 0.0                                # lane placement does not change a trajectory's filter
    assert d["infer_out_max_abs_diff"] < 1e-6 and d["infer_status_nonzero"] == 0
    assert d["train_split_allreduce"] is True                                  # the side-stream half behind os_gru_backward_mark ran
    assert d["train_grad_max_abs_diff"] < 2e-5 * d["train_grad_scale"] + 1e-9  # fp32 reduction-order noise
    assert d["train_replica_weight_max_abs_diff"] == 0.0 and d["train_replica_divergence"] == 0.0
    # Adam's first step moves every weight by ~lr * sign(g): equal up to the sign of noise-level gradient entries
    assert d["train_weight_vs_single_process_max_abs_diff"] <= 2.0e-4 + 1e-7
    assert d["grad_bucket_bytes"] == 1689696


def test_two_ranks_on_one_gpu_equal_the_single_process_run():
    d = _run_worker(2, 29641)
    _check(d, 2)
    assert d["train_shard_sizes"] == [2048, 2048]


def test_three_ranks_ragged_shards_on_one_gpu():
    """4,099 windows over three ranks = 1367 / 1366 / 1366: the size-weighted bucket (count in the extra slot of the same
    all-reduce) reproduces the full-batch mean gradient from the ranks' local-mean gradients."""
    d = _run_worker(3, 29642, train_batch=4099)
    _check(d, 3)
    assert d["train_shard_sizes"] == [1367, 1366, 1366]


@pytest.mark.parametrize("mode", ["fused", "train"])
def test_bench_gpus_2_share_gpu(mode):
    args = ["--gpus", "2", "--share-gpu", "--steps", "2", "--warmup", "1", "--cpu-seconds", "0", "--mode", mode]
    if mode == "fused":
        args += ["--batch", "8192", "--seq", "20"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=900, env=ENV)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["rccl_world_size"] == 2 and d["backend"] == "gloo" and d["share_gpu"] is True
    assert [x["device"] for x in d["rank_devices"]] == [0, 0] and len({x["pid"] for x in d["rank_devices"]}) == 2
    if mode == "train":
        assert d["allreduce_us"] > 0 and d["grad_bucket_bytes"] == 1689696
        assert d["parity"]["replica_max_abs_diff"] == 0.0 and d["parity"]["ranks"] == 2
        assert d["config"]["global_batch"] == 2 * 8192
    else:
        assert d["parity"]["state_linf"] < 1e-4
        assert d["config"]["global_batch"] == 2 * 8192
