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
