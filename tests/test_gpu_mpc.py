"""GPU parity of the batched convex-MPC force QP (os_mpc_solve) against oracle/mpc_oracle.py (KKT-certified float64 solution of
the problem misc/force_controller.py:70-162 states).  PARITY UNPINNED against qpOASES itself (absent)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import mpc_oracle as mo   # noqa: E402


def _problems(n, seed, scale_cycle=(0.3, 1.0, 3.0)):
    rng = np.random.default_rng(seed)
    pats = [(1, 0, 0, 1), (0, 1, 1, 0), (1, 1, 1, 1), (1, 1, 0, 1), (0, 0, 1, 0), (0, 0, 0, 0), (1, 0, 1, 1), (1, 1, 1, 0)]
    X, R, P, Cn = [], [], [], []
    for t in range(n):
        s = scale_cycle[t % len(scale_cycle)]
        x = np.array([0, 0, 0, 0, 0, 0.28, 0, 0, 0, 0, 0, 0.]) + s * rng.normal(0, [0.05] * 3 + [0.02] * 3 + [0.2] * 3 + [0.1] * 3)
        ref = np.array([0, 0, 0, 0, 0, 0.28, 0, 0, 0, 0.1, 0, 0.]) + s * rng.normal(0, [0.02] * 3 + [0.01] * 3 + [0.05] * 3 + [0.05] * 3)
        p = np.array([0.2, 0.1, -0.28, 0.2, -0.1, -0.28, -0.2, 0.1, -0.28, -0.2, -0.1, -0.28]) + rng.normal(0, 0.01, 12)
        X.append(x); R.append(ref); P.append(p); Cn.append(pats[rng.integers(0, len(pats))])
    f32 = lambda a: np.asarray(a, np.float32)
    return f32(X), f32(R), f32(P), np.asarray(Cn, np.uint8)


def _oracle_kw():
    # the library holds dt, mass and inertia as float32 (os_kf_config); give the oracle the same numbers
    return dict(dt=float(np.float32(0.01)))


def _solve_gpu(eng, X, R, P, Cn, want_all=True):
    dev = eng.device
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a.T)).to(dev)
    c = torch.as_tensor(Cn).to(dev).contiguous().view(torch.int32).reshape(-1)
    return eng.mpc_solve(t(X), t(R), t(P), c, want_all=want_all)


@pytest.fixture(scope="module")
def eng():
    from optistate_amd import Engine
    return Engine(0)


def test_mpc_matches_certified_oracle(eng, monkeypatch):
    X, R, P, Cn = _problems(96, seed=11)
    r = _solve_gpu(eng, X, R, P, Cn)
    u = r["u"].cpu().numpy().T.astype(np.float64)
    assert int(r["status"].abs().max()) == 0
    monkeypatch.setattr(mo, "MASS", float(np.float32(8.8)))
    monkeypatch.setattr(mo, "INERTIA", np.asarray(np.float32([0.05530364, 0.06011944, 0.10530434]), np.float64))
    worst = 0.0
    for k in range(X.shape[0]):
        f_o, u_o, info = mo.mpc_forces(X[k].astype(np.float64), R[k].astype(np.float64), P[k].astype(np.float64), Cn[k], **_oracle_kw())
        assert info["stationarity"] < 1e-9
        worst = max(worst, np.abs(u[k] - u_o).max())
    # float32 outputs of forces up to 150 N: 1e-5 N is the rounding of the store; 2e-4 N bar on all 60 controls
    assert worst < 2e-4, worst


def test_mpc_constraints_hold_and_swing_zero(eng):
    X, R, P, Cn = _problems(512, seed=5, scale_cycle=(1.0, 3.0, 6.0))
    r = _solve_gpu(eng, X, R, P, Cn)
    assert int(r["status"].abs().max()) == 0
    u = r["u"].cpu().numpy().T.reshape(-1, 5, 4, 3)
    for leg in range(4):
        sw = Cn[:, leg] == 0
        assert np.all(u[sw][:, :, leg, :] == 0.0)
        st = Cn[:, leg] == 1
        fz = u[st][:, :, leg, 2]
        assert fz.min() >= 0.0 and fz.max() <= 150.0 + 1e-4
        assert np.all(np.abs(u[st][:, :, leg, 0]) <= 0.6 * fz + 1e-4)
        assert np.all(np.abs(u[st][:, :, leg, 1]) <= 0.6 * fz + 1e-4)
    it = r["iters"].cpu().numpy()
    assert it.min() >= 0 and it.max() < 200      # 0 iterations: no leg on the ground
