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


def _run_worker(world, port, train_batch=None, drop_rank=None):
    env = dict(ENV, OS_SHARE_GPU="1")
    if train_batch is not None:
        env["OS_WORKER_TRAIN_BATCH"] = str(train_batch)
    if drop_rank is not None:
        env["OS_WORKER_DROP_RANK"] = str(drop_rank)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "multi_gpu_worker.py")],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    if drop_rank is not None:
        lost = [l for l in r.stdout.splitlines() if l.startswith("LOST ")]
        assert len(lost) == 1, r.stdout
        return _json_line(r.stdout), json.loads(lost[0][5:])
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


def test_a_lost_producer_on_one_rank_never_reaches_the_collective():
    """ADVICE r5 (medium x2): rank 1's stacked launches lose a producer in every step (OS_STACK_DBG_DROP on that rank's context only).
    DataParallelTrainer.step verifies once between the backward and the all-reduce (os_stack_check) and redoes the step's forward,
    loss and backward with a launch per layer BEFORE the gradients are reduced: rank 1 reports three redone steps, rank 0 none, the
    replicas are identical and finite after three Adam steps and equal the clean single-process run on the full batch (fp32
    reduction order), and each rank's context is back in its own stack mode."""
    d, lost = _run_worker(2, 29644, drop_rank=1)
    _check(d, 2)
    by = {x["rank"]: x for x in lost["lost"]}
    assert by[1]["lost_steps"] == 3 and by[0]["lost_steps"] == 0
    assert by[0]["loss_finite"] and by[1]["loss_finite"] and by[0]["stack_mode_after"] == 1 and by[1]["stack_mode_after"] == 1
    assert lost["weights_finite"] and lost["replica_weight_max_abs_diff"] == 0.0
    assert lost["weight_vs_clean_run_max_abs_diff"] <= 3 * 2.0e-4 + 1e-7      # three steps of ~lr each; sign flips of noise-level entries


def test_bench_scaling_strong_two_ranks_equal_the_one_rank_checksum():
    """`bench.py --scaling strong`: the SAME 16,384 trajectories run by one rank and by two ranks (8,192 each, other tile shape) --
    the global checksums agree to fp32 noise (SURVEY 8(e) determinism check), `value` counts the global batch once."""
    outs = {}
    for n in (1, 2):
        args = ["--gpus", str(n), "--scaling", "strong", "--batch", "16384", "--seq", "20", "--steps", "2", "--warmup", "1", "--cpu-seconds", "0",
                "--parity-samples", "128", "--no-second-noise"] + (["--share-gpu"] if n > 1 else [])
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=900, env=ENV)
        assert r.returncode == 0, r.stderr[-3000:]
        outs[n] = _json_line(r.stdout)
    a, b = outs[1], outs[2]
    assert a["scaling"] == b["scaling"] == "strong" and a["config"]["global_batch"] == b["config"]["global_batch"] == 16384
    assert a["config"]["batch_per_gpu"] == 16384 and b["config"]["batch_per_gpu"] == 8192 and b["n_gpus"] == 2
    assert a["kernels"]["fused"]["kernel"] == "fused_kf_gru_kernel_v3<1>" and b["kernels"]["fused"]["kernel"] == "fused_kf_gru_kernel_v3<2>"
    ca, cb = a["global_checksum"], b["global_checksum"]
    assert ca["trajectories"] == cb["trajectories"] == 16384
    assert abs(ca["sum_x_final"] - cb["sum_x_final"]) < 1e-3 and abs(ca["sum_out"] - cb["sum_out"]) < 1e-2     # sums over 16,384 x 12 / x 24 values
    assert abs(b["value"] - 16384 * 20 / (b["ms_per_step"] * 1e-3)) < 1e-6 * b["value"]
    assert b["parity"]["state_linf"] < 1e-4


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
