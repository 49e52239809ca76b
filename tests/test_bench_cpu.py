"""CPU: the pieces of bench.py that run without a GPU -- the cpu_baseline leg (C oracle over the host cores, the only place
outside tests/ and smoke() that may call oracle/) and the core-count logic."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def test_usable_cores_is_positive_and_bounded():
    import bench
    n = bench.usable_cores()
    assert 1 <= n <= (os.cpu_count() or 1)


def test_cpu_baseline_object_has_the_contract_keys():
    import bench
    cb = bench.cpu_baseline(64, 1, 0.5)
    for k in ("value", "unit", "cores", "kind", "sample", "single_thread_value", "gru_half_torch_cpu"):
        assert k in cb, k
    assert cb["kind"] == "port" and cb["unit"] == "timesteps/s" and cb["value"] > 0 and cb["single_thread_value"] > 0
    assert cb["cores"] == bench.usable_cores()


def test_cpu_baseline_carries_the_reference_python_figure():
    import bench
    cb = bench.cpu_baseline(64, 1, 0.3, kf_only=True)
    assert cb["reference_python_steps_per_s"] == 3.05e3 and "build container" in cb["reference_python_note"]
    assert "gru_half_torch_cpu" not in cb and "KF float64" in cb["sample"]


def test_launch_command_is_the_contract_command():
    import bench
    cmd = bench.launch_command(4, ["--gpus", "4", "--steps", "3", "--warmup", "1"], 29512)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert "--nproc-per-node=4" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29512"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")


def test_gpus_2_starts_its_own_ranks_and_reports_the_process_group():
    """`python bench.py --gpus 2` (not under torchrun) must start the two ranks itself, before any GPU call.  Without a GPU
    the rendezvous self-test (--launch-check) runs the identical launch path over gloo: two rank processes join, all-reduce,
    and rank 0 prints ONE JSON line with the world size the process group reports."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env = {k: v for k, v in env.items() if k not in ("NCCL_PROTO", "NCCL_ALGO")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check", "--rccl-proto", "LL128"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["launch_check"] is True and d["rccl_world_size"] == 2 and d["sum_of_ones"] == 2.0
    # --rccl-proto reaches every rank's environment before its communicator exists (gloo ignores it; RCCL reads it there)
    assert all(x["rccl_env"] == {"NCCL_PROTO": "LL128"} for x in d["rank_devices"])
    assert sorted(x["rank"] for x in d["rank_devices"]) == [0, 1]
    assert len({x["pid"] for x in d["rank_devices"]}) == 2            # two separate rank processes


def test_scaling_strong_shards_one_fixed_batch_over_the_ranks():
    """`bench.py --gpus 3 --scaling strong --batch 65536`: --batch is the GLOBAL batch; rank r runs the contiguous shard
    shard_range(B, r, N) ("GPU g gets trajectories [g B/G, (g+1) B/G)").  The launcher self-test reports the shards
    every rank computed: they tile the batch exactly, ragged sizes differ by at most one."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "NCCL_PROTO", "NCCL_ALGO")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--launch-check", "--scaling", "strong", "--batch", "65536"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["scaling"] == "strong" and d["global_batch"] == 65536 and d["rccl_world_size"] == 3
    sh = d["shards"]
    assert sh[0][0] == 0 and sh[-1][1] == 65536 and all(sh[i][1] == sh[i + 1][0] for i in range(2))
    assert sorted(hi - lo for lo, hi in sh) == [21845, 21845, 21846]
    # the modes that do not shard trajectories refuse the flag
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--mode", "train", "--scaling", "strong"], capture_output=True, text=True,
                       timeout=120, env=env)
    assert r.returncode != 0 and "strong" in (r.stderr + r.stdout)


def test_a_rank_with_the_wrong_world_size_refuses():
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"], capture_output=True,
                       text=True, timeout=120, env=env)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_traffic_figure_is_served_only_with_matching_provenance(tmp_path, monkeypatch):
    """roofline.traffic comes from profiles/traffic.json (PMC passes of an earlier GPU visit): bench.load_traffic returns it only
    when the entry's recorded object key equals the key of the sources on disk -- a kernel changed under the same name reports
    null plus the reason in roofline.traffic_source."""
    import json
    import bench
    from optistate_amd import build as b
    key = b.object_key("fused_kernels.hip", b.toolchain_id())
    (tmp_path / "profiles").mkdir()
    tj = tmp_path / "profiles" / "traffic.json"
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    entry = {"fused_kf_gru_kernel_v2": 1.7e9,
             "fused_kf_gru_kernel_v2_detail": {"source_key": key, "source": "fused_kernels.hip", "collected": "2026-10-03"}}
    tj.write_text(json.dumps(entry))
    v, src = bench.load_traffic("fused_kf_gru_kernel_v2<true, false>", True)
    assert v == 1.7e9 and key in src and "2026-10-03" in src
    assert bench.load_traffic("fused_kf_gru_kernel_v2", False) == (None, "not collected for this shape / mode")
    entry["fused_kf_gru_kernel_v2_detail"]["source_key"] = "0" * 24
    tj.write_text(json.dumps(entry))
    v, src = bench.load_traffic("fused_kf_gru_kernel_v2", True)
    assert v is None and "stale" in src
    del entry["fused_kf_gru_kernel_v2_detail"]["source_key"]          # an entry from before provenance was recorded
    tj.write_text(json.dumps(entry))
    assert bench.load_traffic("fused_kf_gru_kernel_v2", True)[0] is None
    assert bench.load_traffic("kf_run_sym_kernel", True)[0] is None   # no entry at all
