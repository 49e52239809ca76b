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
