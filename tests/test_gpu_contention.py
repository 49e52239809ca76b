"""GPU: the layer-pipelined launches (gru_stack_kernel: layer l takes layer l - 1's steps through progress counters) while ANOTHER
process keeps every CU of the same GPU busy.  A consumer workgroup can then sit resident while its producer waits for a CU; the
design's answer is a bounded wait + error word + a launch-per-layer re-run.  Whatever happens underneath -- the
stacked launch goes through, or the engine falls back -- every call must return one of the two known-good results."""
import os
import subprocess
import sys
import time

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu

HOG = r'''
import sys, time, torch
sys.path.insert(0, sys.argv[1])
from optistate_amd import Engine, RNN, flatten_state_dict
eng = Engine(0)
m = RNN(188, 128, 4, 24, torch.device("cpu"))
eng.load_gru(flatten_state_dict(m.state_dict(), 4, "cuda"), 188, 128, 4, 24)
xs = torch.rand(40, 188, 32768, device="cuda")            # 256 tiles of 128 / 512 of 64 rows: every CU busy, ~15 ms per call
print("hog ready", flush=True)
t0 = time.time()
while time.time() - t0 < float(sys.argv[2]):
    eng.gru_forward_soa(xs)
    torch.cuda.synchronize()
print("hog done", flush=True)
'''


def test_stacked_launches_beside_a_process_that_fills_the_gpu():
    from optistate_amd import RNN
    torch.manual_seed(9)
    m = RNN(188, 128, 4, 24, torch.device("cuda")).to("cuda").eval()
    x = torch.rand(64, 10, 188, device="cuda")             # the reference's own batch (gru/gru_train.py:36): one stacked launch
    with torch.no_grad():
        m(x)
        eng = m._engine
        assert eng.kernel_name("gru_layer") in ("gru_stack_kernel", "gru_wide_kernel")      # one launch for the stack (H = 128 up to 512 windows: four CUs per (layer, tile))
        ref_stack = m(x).clone()                            # the stacked launch on an idle GPU
        eng.set_stack_mode(0)
        ref = m(x).clone()                                  # a launch per layer (other kernels: another summation order, ~1e-7 apart)
        eng.set_stack_mode(1)
    assert (ref - ref_stack).abs().max().item() < 1e-6
    hog = subprocess.Popen([sys.executable, "-c", HOG, ROOT, "12"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    try:
        line = hog.stdout.readline()
        while line and "hog ready" not in line:
            line = hog.stdout.readline()
        assert "hog ready" in line, line
        t0, n, worst = time.time(), 0, 0.0
        with torch.no_grad():
            while time.time() - t0 < 8.0:
                t1 = time.time()
                out = m(x)
                torch.cuda.synchronize()
                worst = max(worst, time.time() - t1)
                # bit for bit the idle-GPU stacked result, or -- if the launch lost a producer and the engine re-ran it -- the per-layer one
                assert torch.equal(out, ref_stack) or torch.equal(out, ref), n
                n += 1
        fb = getattr(eng, "stack_fallbacks", 0)
        print(f"{n} stacked forwards beside the hog: all equal to a known-good result; {fb} fell back to a launch per layer; slowest call {worst * 1e3:.1f} ms")
        assert n > 20
    finally:
        hog.wait(timeout=60)
    assert eng.kernel_name("gru_layer") in ("gru_stack_kernel", "gru_wide_kernel")     # the mode is back on after any fallback


def test_stacked_training_sweeps_beside_a_process_that_fills_the_gpu():
    """The batch-64 training step's two stacked launches (forward with saved activations, bwd_sweep_stack_kernel) beside the same hog:
    every gradient equals the idle-GPU one to the summation order of the dW atomics."""
    from optistate_amd import RNN, default_engine, flatten_state_dict
    torch.manual_seed(10)
    I, H, L, C, B, T = 188, 128, 4, 24, 64, 10
    m = RNN(I, H, L, C, torch.device("cuda")).to("cuda")
    eng = default_engine(0)
    eng.load_gru(flatten_state_dict(m.state_dict(), L, "cuda"), I, H, L, C)
    x = torch.rand(B, T, I, device="cuda"); y = torch.rand(B, C // 2, device="cuda")

    def grad():
        out = eng.gru_forward_train(x)
        _, dout, _ = eng.gru_loss(out, y, want_target=True)
        return out, eng.gru_backward(x, out, dout).clone()

    out0, g0 = grad()
    assert eng.kernel_name("gru_layer") in ("gru_stack_kernel", "gru_wide_kernel") and eng.kernel_name("train_sweep") in ("bwd_sweep_stack_kernel", "bwd_sweep_wide_kernel")
    scale = g0.abs().max().item()
    hog = subprocess.Popen([sys.executable, "-c", HOG, ROOT, "10"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    try:
        line = hog.stdout.readline()
        while line and "hog ready" not in line:
            line = hog.stdout.readline()
        assert "hog ready" in line, line
        t0, n = time.time(), 0
        while time.time() - t0 < 6.0:
            out, g = grad()
            torch.cuda.synchronize()
            assert torch.isfinite(g).all() and (out - out0).abs().max().item() < 1e-6, n
            assert (g - g0).abs().max().item() < 1e-4 * scale, n
            n += 1
        print(f"{n} forward + backward passes at batch 64 beside the hog: gradients equal the idle-GPU ones; "
              f"{getattr(eng, 'stack_fallbacks', 0)} fell back to a launch per layer")
        assert n > 20
    finally:
        hog.wait(timeout=60)


def test_one_long_kernel_on_30_of_32_cus_per_xcd_starves_a_tile_and_the_call_falls_back(tmp_path):
    """The pattern that CAN starve a producer (docs/stack_protocol.md: co-residency is a LIVENESS assumption): another process holds 240 of the
    256 CUs with ONE kernel that runs for seconds (tools/micro/cu_hog.hip: 150 KB of LDS per workgroup, so nothing of ours fits beside it) --
    two CUs stay free on every XCD, 16 in all, and gru_wide_kernel's 32 workgroups cannot all be resident.  The resident ones wait for
    partners that are not scheduled; their bounded wait expires (2^22 polls = 0.67 s), they poison their tiles and set the error word.
    Measured (tools/hog_diag.py): the launch itself drains only when the other process's kernel leaves -- the dispatcher does not place the
    remaining workgroups before that -- so the call returns -20 at the hog's end, whatever the poll bound; the engine then re-runs it
    with a launch per layer (0.8 ms) and the result is the per-layer one bit for bit.  (gru_stack_kernel, one CU per (layer, tile): its 8
    workgroups fit into the 16 free CUs and the call takes its usual 0.5 ms beside the same hog.)"""
    import shutil
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    exe = str(tmp_path / "cu_hog")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", os.path.join(ROOT, "tools", "micro", "cu_hog.hip"), "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    from optistate_amd import RNN
    torch.manual_seed(9)
    m = RNN(188, 128, 4, 24, torch.device("cuda")).to("cuda").eval()
    x = torch.rand(64, 10, 188, device="cuda")
    with torch.no_grad():
        m(x)
        eng = m._engine
        if eng.kernel_name("gru_layer") != "gru_wide_kernel":
            pytest.skip("the four-CUs-per-tile kernel is switched off in this environment")
        eng.set_stack_mode(0)
        ref = m(x).clone()                                  # a launch per layer, idle GPU
        eng.set_stack_mode(1)
    torch.cuda.synchronize()
    fb0 = eng.stack_fallbacks
    HOG_S = 3.0
    hog = subprocess.Popen([exe, "240", str(int(HOG_S * 1000))], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    try:
        line = hog.stdout.readline()
        assert "hog running" in line, line
        time.sleep(0.3)                                     # the hog's workgroups are resident
        t0 = time.time()
        with torch.no_grad():
            out = m(x)
        torch.cuda.synchronize()
        el = time.time() - t0
    finally:
        hog.wait(timeout=60)
    assert torch.equal(out, ref)                            # the per-layer result, nothing poisoned left behind
    assert eng.stack_fallbacks == fb0 + 1                   # -20 -> one fallback
    assert 0.5 < el < HOG_S + 1.0, el                       # at least the bounded wait; at most the hog's lifetime + the per-layer launches
    print(f"starved launch: -20 and the fallback to a launch per layer after {el:.2f} s (hog: {HOG_S} s on 240 of 256 CUs)")
    with torch.no_grad():
        assert torch.equal(m(x), m(x)) and eng.kernel_name("gru_layer") == "gru_wide_kernel"      # the stacked mode is back on


def test_the_drain_phase_of_the_qp_launch_needs_no_co_residency(tmp_path, monkeypatch):
    """Round 6: the filter step inside the QP launch (mpc_quad.hip) is one workgroup waiting for another inside a launch too -- but a
    drained wavefront only ever waits for a row that is RUNNING (docs/stack_protocol.md, last section).  Beside the same hog that
    starves the four-CUs-per-tile GRU kernel (240 of 256 CUs held by one kernel for seconds: of the QP launch's 2,048 wavefronts a
    few dozen are resident at a time) os_kf_mpc_run in its fused, two-part form completes with every status word clean and the
    numbers of the idle-GPU run, bit for bit."""
    import shutil
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    exe = str(tmp_path / "cu_hog")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", os.path.join(ROOT, "tools", "micro", "cu_hog.hip"), "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    from optistate_amd import Engine
    from optistate_amd.synth import synth_torch, Q_DEFAULT, R_DEFAULT
    monkeypatch.setenv("OS_MPC_PERSISTENT", "0")
    B, T = 32768, 4
    dev = torch.device("cuda:0")
    d = synth_torch(B, T, dev, seed=31)
    eng = Engine(0); eng.set_noise(Q_DEFAULT, R_DEFAULT)
    contact = eng.contact_soa_to_packed(d["contact"])
    ref = torch.zeros((T, 12, B), device=dev); ref[:, 5] = 0.28; ref[:, 9] = 0.1

    def run():
        x, P = d["x0"].clone(), d["P0"].clone()
        r = eng.kf_mpc_run(d["p"], d["dp"], d["imu"], contact, ref, x, P, want_iters=True)
        torch.cuda.synchronize()
        return r, x, P
    r0, x0, P0 = run()
    assert "filter step inside" in eng.kernel_name("mpc") and "two in flight" in eng.kernel_name("mpc")
    HOG_S = 3.0
    hog = subprocess.Popen([exe, "240", str(int(HOG_S * 1000))], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    try:
        line = hog.stdout.readline()
        assert "hog running" in line, line
        time.sleep(0.3)
        t0 = time.time()
        r1, x1, P1 = run()
        el = time.time() - t0
    finally:
        hog.wait(timeout=60)
    for k in ("x_out", "f", "iters", "status"):
        assert torch.equal(r0[k], r1[k]), k
    assert torch.equal(x0, x1) and torch.equal(P0, P1) and int(r1["status"].abs().max()) == 0
    print(f"os_kf_mpc_run (fused, two parts, {B} x {T}) beside a hog on 240 of 256 CUs: {el:.2f} s, identical to the idle-GPU run")
