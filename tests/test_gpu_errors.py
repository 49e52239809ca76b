"""GPU: the C-ABI's argument checking (return code < 0 + os_last_error instead of undefined behaviour): empty batches, null
required pointers, flag combinations the path does not define."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from optistate_amd import Engine
    return Engine(0)


def _bufs(B, T):
    z = lambda *s: torch.zeros(s, dtype=torch.float32, device="cuda")
    return dict(p=z(T, 12, B), f=z(T, 12, B), dp=z(T, 12, B), imu=z(T, 6, B), c=torch.zeros((T, B), dtype=torch.int32, device="cuda"),
                x=z(12, B), P=z(144, B), xo=z(T, 12, B), st=torch.zeros(B, dtype=torch.int32, device="cuda"))


def test_kf_run_rejects_bad_arguments(eng):
    from optistate_amd.engine import _ptr
    b = _bufs(4, 3)
    lib, h = eng.lib, eng._h
    call = lambda B, T, p, flags=1, x=b["x"]: lib.os_kf_run(h, B, T, _ptr(p), _ptr(b["f"]), _ptr(b["dp"]), _ptr(b["imu"]), _ptr(b["c"]),
                                                            None, _ptr(x), _ptr(b["P"]), _ptr(b["xo"]), None, None, None, _ptr(b["st"]),
                                                            flags, None)
    assert call(4, 3, b["p"]) == 0
    assert call(0, 3, b["p"]) < 0 and b"positive" in lib.os_last_error(h)          # empty batch
    assert call(4, 0, b["p"]) < 0                                                   # empty horizon
    assert call(4, 3, None) < 0 and b"null" in lib.os_last_error(h)                 # missing stream
    assert call(4, 3, b["p"], flags=1 | 2) < 0 and b"body_ref" in lib.os_last_error(h)   # dense F_d without body_ref
    assert lib.os_kf_run(None, 4, 3, *([None] * 13), 0, None) < 0                   # no context


def test_sequential_update_needs_diagonal_R(eng):
    from optistate_amd import Engine
    from optistate_amd.synth import Q_DEFAULT, R_DEFAULT
    e2 = Engine(0)
    R = R_DEFAULT.copy(); R[0, 1] = R[1, 0] = 1e-3
    e2.set_noise(Q_DEFAULT, R)
    b = _bufs(4, 2)
    with pytest.raises(RuntimeError, match="diagonal"):
        e2.kf_run(b["p"], b["f"], b["dp"], b["imu"], b["c"], b["x"], b["P"], sequential=True)
    r = e2.kf_run(b["p"], b["f"], b["dp"], b["imu"], b["c"], b["x"], b["P"])       # default falls back to the batch form
    assert r["x_out"].shape == (2, 12, 4)


def test_fused_and_gru_preconditions(eng):
    from optistate_amd import Engine
    e2 = Engine(0)
    b = _bufs(4, 2)
    mm = torch.zeros((2, 60), device="cuda"); mm[1] = 1
    from optistate_amd.engine import _ptr
    acc = torch.zeros((2, 6, 4), device="cuda"); out = torch.zeros((4, 24), device="cuda")
    rc = e2.lib.os_fused_run(e2._h, 4, 2, _ptr(b["p"]), _ptr(b["f"]), _ptr(b["dp"]), _ptr(b["imu"]), _ptr(b["c"]), _ptr(acc), None, None, 0,
                             _ptr(mm), _ptr(b["x"]), _ptr(b["P"]), _ptr(b["xo"]), _ptr(out), _ptr(b["st"]), 1 | 4, None)
    assert rc < 0 and b"os_gru_load" in e2.lib.os_last_error(e2._h)                 # fused path before any weights were loaded
    with pytest.raises(ValueError):
        e2.load_gru(torch.zeros(10, device="cuda"), 60, 64, 1, 24)                  # wrong flat-parameter count


def test_window_stream_and_split_bf16_preconditions():
    """os_gru_forward_windows / os_gru_bands / os_gru_set_split_bf16: return codes for what they do not define."""
    from optistate_amd import Engine, RNN, flatten_state_dict
    from optistate_amd.engine import _ptr
    e2 = Engine(0)
    lib, h = e2.lib, e2._h
    rows = torch.rand(20, 60, device="cuda"); out = torch.zeros(20, 24, device="cuda")
    assert lib.os_gru_forward_windows(h, 20, 10, _ptr(rows), _ptr(out), None) < 0 and b"os_gru_load" in lib.os_last_error(h)     # no weights yet
    m = RNN(60, 64, 1, 24, torch.device("cpu"))
    e2.load_gru(flatten_state_dict(m.state_dict(), 1, "cuda"), 60, 64, 1, 24)
    assert lib.os_gru_forward_windows(h, 20, 10, _ptr(rows), _ptr(out), None) == 0
    assert lib.os_gru_forward_windows(h, 20, 21, _ptr(rows), _ptr(out), None) < 0 and b"window" in lib.os_last_error(h)          # window longer than the stream
    assert lib.os_gru_forward_windows(h, 0, 1, _ptr(rows), _ptr(out), None) < 0                                                  # empty stream
    assert lib.os_gru_forward_windows(h, 20, 10, None, _ptr(out), None) < 0                                                      # null rows
    with pytest.raises(ValueError):
        e2.gru_forward_windows(rows, 0)
    m32 = RNN(60, 32, 1, 24, torch.device("cpu"))                                                                                  # hidden 32: not a window-stream shape
    e2.load_gru(flatten_state_dict(m32.state_dict(), 1, "cuda"), 60, 32, 1, 24)
    assert lib.os_gru_forward_windows(h, 20, 10, _ptr(rows), _ptr(out), None) == -4 and not e2.gru_windows_supported()
    assert lib.os_gru_set_split_bf16(h, 1) == -4 and lib.os_gru_set_split_bf16(h, 4) == -4 and lib.os_gru_set_split_bf16(h, 3) == 0
    assert lib.os_gru_set_split_bf16(h, 0) == 0 and lib.os_gru_set_split_bf16(None, 3) < 0


def test_engine_refuses_host_strided_and_float64_tensors(eng):
    """The C-ABI sees only addresses: the Python engine refuses what the library would misread (a host tensor, a strided view, float64)."""
    b = _bufs(8, 3)
    args = lambda **kw: [kw.get(k, b[k]) for k in ("p", "f", "dp", "imu", "c", "x", "P")]
    assert eng.kf_run(*args())["x_out"].shape == (3, 12, 8)
    with pytest.raises(TypeError, match="device tensor"):
        eng.kf_run(*args(p=b["p"].cpu()))
    with pytest.raises(ValueError, match="contiguous"):
        eng.kf_run(*args(f=torch.zeros((3, 8, 12), device="cuda").permute(0, 2, 1)))
    with pytest.raises(TypeError, match="float32"):
        eng.kf_run(*args(imu=b["imu"].double()))


def test_mpc_rejects_bad_arguments(eng):
    from optistate_amd.engine import _ptr
    lib, h = eng.lib, eng._h
    z = torch.zeros((12, 2), device="cuda"); c = torch.zeros(2, dtype=torch.int32, device="cuda")
    st = torch.zeros(2, dtype=torch.int32, device="cuda")
    assert lib.os_mpc_solve(h, 0, _ptr(z), _ptr(z), _ptr(z), _ptr(c), _ptr(z), None, None, _ptr(st), 0, None) < 0
    assert lib.os_mpc_solve(h, 2, None, _ptr(z), _ptr(z), _ptr(c), _ptr(z), None, None, _ptr(st), 0, None) < 0
    w = (C.c_double * 12)(*([1.0] * 12))
    assert lib.os_mpc_set_weights(h, w, C.c_double(-1.0), C.c_double(0.6), C.c_double(150.0)) < 0
    assert lib.os_mpc_set_weights(h, None, C.c_double(1e-6), C.c_double(0.6), C.c_double(150.0)) < 0


# ---- the progress-counter kernels' error path (include/optistate_hip.h: os_gru_set_stack) ----
def _ref_forward(x, sd_flat, dims):
    import os
    from optistate_amd import Engine
    old = os.environ.get("OS_GRU_STACK")
    os.environ["OS_GRU_STACK"] = "0"
    try:
        e = Engine(0)
        e.load_gru(sd_flat, *dims)
        return e.gru_forward(x)
    finally:
        if old is None:
            del os.environ["OS_GRU_STACK"]
        else:
            os.environ["OS_GRU_STACK"] = old


def test_lost_producer_fails_the_call_and_the_engine_falls_back_to_a_launch_per_layer(monkeypatch):
    """OS_STACK_DBG_DROP withholds layer 1's publishes from step 3 on, OS_STACK_DBG_POLLS shortens the bounded wait: layer 2's wait
    expires, it sets the context's error word and poisons its input.  The raw C call returns -20 and names the kernel; the Python
    engine re-runs the call with a launch per layer and returns the right numbers (VERDICT r4 missing 4: no silent NaN)."""
    import ctypes as C
    import torch
    from optistate_amd import Engine, RNN, flatten_state_dict
    from optistate_amd.engine import _ptr
    dims = (188, 128, 4, 24)
    torch.manual_seed(2)
    m = RNN(*dims, torch.device("cpu"))
    flat = flatten_state_dict(m.state_dict(), 4, "cuda")
    x = torch.rand(64, 10, 188, device="cuda")
    ref = _ref_forward(x, flat, dims)
    monkeypatch.setenv("OS_STACK_DBG_DROP", "1,3"); monkeypatch.setenv("OS_STACK_DBG_POLLS", "3000"); monkeypatch.setenv("OS_GRU_VEC", "0")
    eng = Engine(0)
    eng.load_gru(flat, *dims)
    out = torch.empty((64, 24), device="cuda")
    rc = eng.lib.os_gru_forward(eng._h, 64, 10, _ptr(x), _ptr(out), None, eng._stream())
    assert rc == -20 and (b"gru_stack_kernel" in eng.lib.os_last_error(eng._h) or b"gru_wide_kernel" in eng.lib.os_last_error(eng._h)) and b"os_gru_set_stack" in eng.lib.os_last_error(eng._h)
    got = eng.gru_forward(x)                                   # the engine: same failure, then a launch per layer
    assert eng.stack_fallbacks == 1 and torch.isfinite(got).all()
    assert (got - ref).abs().max().item() < 2e-6
    # asynchronous mode: the call returns at once; the NEXT call on the context reports it
    eng.set_stack_mode(2)
    rc = eng.lib.os_gru_forward(eng._h, 64, 10, _ptr(x), _ptr(out), None, eng._stream())
    assert rc == 0
    torch.cuda.synchronize()
    assert not torch.isfinite(out).all()
    rc = eng.lib.os_gru_forward(eng._h, 64, 10, _ptr(x), _ptr(out), None, eng._stream())
    assert rc == -20 and b"EARLIER" in eng.lib.os_last_error(eng._h)
    torch.cuda.synchronize()


def test_adam_kernel_skips_its_update_behind_a_lost_producer(monkeypatch):
    """Asynchronous mode (os_gru_set_stack 2): the optimiser step enqueued behind a stacked launch that lost a producer must leave
    the weights and moments alone (gru/gru_train.py:247-249 would step Adam on NaN gradients)."""
    import torch
    from optistate_amd import Engine, RNN, flatten_state_dict
    from optistate_amd.engine import _ptr
    dims = (188, 128, 4, 24)
    torch.manual_seed(2)
    m = RNN(*dims, torch.device("cpu"))
    flat = flatten_state_dict(m.state_dict(), 4, "cuda")
    x = torch.rand(64, 10, 188, device="cuda")
    monkeypatch.setenv("OS_STACK_DBG_DROP", "1,3"); monkeypatch.setenv("OS_STACK_DBG_POLLS", "400000"); monkeypatch.setenv("OS_GRU_VEC", "0")
    eng = Engine(0)
    eng.load_gru(flat, *dims)
    eng.set_stack_mode(2)
    w = flat.clone(); w0 = w.clone()
    g = torch.full_like(w, float("nan")); mo = torch.zeros_like(w); v = torch.zeros_like(w)
    out = torch.empty((64, 24), device="cuda")
    assert eng.lib.os_gru_forward(eng._h, 64, 10, _ptr(x), _ptr(out), None, eng._stream()) == 0          # in flight: its wait takes ~0.1 s
    eng.adam_step(w, g, mo, v, 1e-4, 0.9, 0.999, 1e-8, 1)                                                # enqueued behind it
    torch.cuda.synchronize()
    assert torch.equal(w, w0) and float(mo.abs().sum()) == 0.0 and float(v.abs().sum()) == 0.0
    with pytest.raises(RuntimeError, match="EARLIER"):
        eng.adam_step(w, g, mo, v, 1e-4, 0.9, 0.999, 1e-8, 1)
    g.fill_(0.5)
    eng.adam_step(w, g, mo, v, 1e-4, 0.9, 0.999, 1e-8, 1)                                                # the word is cleared: updates run again
    torch.cuda.synchronize()
    assert not torch.equal(w, w0)


def test_trainer_redoes_a_lost_step_before_the_optimiser_and_keeps_the_callers_stack_mode(monkeypatch):
    """DataParallelTrainer runs its stacked launches asynchronously and verifies ONCE per step (os_stack_check) before the
    all-reduce / Adam.  With layer 1 withheld (OS_STACK_DBG_DROP) every step loses a producer: the step's forward, loss and backward
    are redone with a launch per layer, the loss is finite, and two steps end on the weights of a clean trainer (gru/gru_train.py:232-249).
    The context's own mode is restored exactly: a process that opted out of stacked launches (OS_GRU_STACK=0) stays opted out."""
    import torch
    from optistate_amd import RNN, engine as _engine
    from optistate_amd.train import DataParallelTrainer
    dims = (188, 128, 4, 24)
    g = torch.Generator(device="cuda"); g.manual_seed(11)
    x, y = torch.rand(64, 10, 188, device="cuda", generator=g), torch.rand(64, 12, device="cuda", generator=g)

    def run(env):
        for k in ("OS_STACK_DBG_DROP", "OS_STACK_DBG_POLLS", "OS_GRU_STACK"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        _engine._default_engines.__dict__.pop("engines", None)          # a fresh per-thread context: os_create reads the knobs
        torch.manual_seed(4)
        m = RNN(*dims, torch.device("cuda")).to("cuda")
        tr = DataParallelTrainer(m, lr=1e-4)
        mode0 = tr.eng._stack_mode
        losses = [tr.step(x, y) for _ in range(2)]
        torch.cuda.synchronize()
        return tr, mode0, losses

    try:
        clean, mode_clean, _ = run({})
        assert mode_clean == 1 and clean.eng._stack_mode == 1 and clean.lost_steps == 0
        lossy, mode_l, losses = run({"OS_STACK_DBG_DROP": "1,3", "OS_STACK_DBG_POLLS": "3000"})
        assert lossy.lost_steps == 2 and lossy.eng._stack_mode == 1 and all(torch.isfinite(l).all() for l in losses)
        assert torch.isfinite(lossy.bucket.w).all()
        assert (lossy.bucket.w - clean.bucket.w).abs().max().item() <= 2 * 2.0e-4 + 1e-7      # per-layer vs stacked launches: fp32 noise through two Adam steps
        off, mode_off, _ = run({"OS_GRU_STACK": "0", "OS_STACK_DBG_DROP": "1,3", "OS_STACK_DBG_POLLS": "3000"})
        assert mode_off == 0 and off.eng._stack_mode == 0 and off.eng.lib.os_gru_get_stack(off.eng._h) == 0 and off.lost_steps == 0
        asyn, mode_a, _ = run({"OS_GRU_STACK": "2"})
        assert mode_a == 2 and asyn.eng._stack_mode == 2 and asyn.eng.lib.os_gru_get_stack(asyn.eng._h) == 2
    finally:
        _engine._default_engines.__dict__.pop("engines", None)


def test_two_threads_two_contexts_equal_the_sequential_results():
    """SURVEY 8(b) threading contract: different contexts from different threads, each on its own stream, concurrently -- the
    Kalman run and the GRU forward of each thread equal what the same calls return one after the other; default_engine() is per
    thread, so two RNN modules evaluated from two threads never share a context's loaded weights."""
    import threading
    import numpy as np
    import torch
    from optistate_amd import Engine, RNN, flatten_state_dict, default_engine
    from optistate_amd.synth import synth_torch, Q_DEFAULT, R_DEFAULT, Q_FITTED, R_FITTED
    dev = torch.device("cuda", 0)
    jobs = []
    for i, (Q, R, dims) in enumerate(((Q_DEFAULT, R_DEFAULT, (60, 64, 1, 24)), (Q_FITTED, R_FITTED, (188, 128, 4, 24)))):
        torch.manual_seed(10 + i)
        m = RNN(*dims, torch.device("cpu"))
        jobs.append(dict(Q=Q, R=R, dims=dims, flat=flatten_state_dict(m.state_dict(), dims[2], dev), d=synth_torch(4096, 50, dev, seed=50 + i),
                         x=torch.rand(512, 10, dims[0], device=dev), model=RNN(*dims, dev).to(dev).eval()))

    def work(j, res, reps):
        torch.cuda.set_device(0)
        eng = Engine(0)
        eng.set_noise(j["Q"], j["R"])
        eng.load_gru(j["flat"], *j["dims"])
        st = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st):
            for _ in range(reps):
                d = j["d"]
                c = eng.contact_soa_to_packed(d["contact"])
                x, P = d["x0"].clone(), d["P0"].clone()
                r = eng.kf_run(d["p"], d["f"], d["dp"], d["imu"], c, x, P)
                o = eng.gru_forward(j["x"])
                with torch.no_grad():
                    mo = j["model"](j["x"])                 # through the module: this thread's default engine
            st.synchronize()
        res.update(x_out=r["x_out"].clone(), out=o.clone(), mout=mo.clone(), engine=id(default_engine(0)))

    seq = [dict(), dict()]
    for j, r in zip(jobs, seq):
        work(j, r, 1)
    par = [dict(), dict()]
    ths = [threading.Thread(target=work, args=(j, r, 20)) for j, r in zip(jobs, par)]
    [t.start() for t in ths]; [t.join() for t in ths]
    for s, p in zip(seq, par):
        assert p and torch.equal(s["x_out"], p["x_out"]) and torch.equal(s["out"], p["out"]) and torch.equal(s["mout"], p["mout"])
    assert par[0]["engine"] != par[1]["engine"]                 # one default context per thread
