"""G12 pins oracle/mpc_oracle.py (and the C oracle's predict_mpc step) to the convex-MPC QP AS THE REFERENCE ASSEMBLES IT.

tests/golden/mpc_g12_qp.npz was produced by tools/gen_golden_mpc.py running misc/force_controller.py:47-225 and
kalman_filter/kalman_filter.py:140-182 UNMODIFIED over an evaluating casadi stand-in (tools/casadi_eval.py) that extracts
H, g and every constraint row by exact polynomial evaluation.  Label: formulation = reference; solver = certified stand-in
(qpOASES absent; the QP is strictly convex, so the minimiser is unique)."""
import os

import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import mpc_oracle as mo

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def g():
    return np.load(os.path.join(GOLD, "mpc_g12_qp.npz"))


def _canon_rows(A, b, is_eq):
    """Set of constraints as sorted tuples: ('eq', row, rhs) with the first non-zero positive, ('ge', row, rhs) for row.u + rhs >= 0;
    rows without variables dropped (they hold as constants: checked by the caller)."""
    out = set()
    for r in range(A.shape[0]):
        a, c = np.round(A[r], 12) + 0.0, round(float(b[r]), 12) + 0.0
        if not np.any(a != 0):
            assert (c == 0) if is_eq[r] else (c >= 0)
            continue
        if is_eq[r]:
            s = np.sign(a[np.nonzero(a)[0][0]])
            a, c = a * s, c * s
            out.add(("eq", tuple(a + 0.0), c + 0.0))
        else:
            out.add(("ge", tuple(a), c))
    return out


def test_cost_matrices_equal_the_references(g):
    """(a) of the round-3 brief: mpc_oracle.build_qp's H, q ARE the reference's, to 1e-12 of their scale, after the index map
    between Opti's decision-variable order (f1, f2, f3, f4 column-major) and the oracle's stage-major order."""
    m = g["opti_to_stage"]
    for k in range(g["x"].shape[0]):
        H, q, C, lo, hi = mo.build_qp(g["x"][k], g["body_ref"][k], g["p"][k], g["contact"][k])
        Hr = g["H"][k][np.ix_(m, m)]
        qr = 0.5 * g["g"][k][m]                      # reference cost = c + g.u + u.H.u ; oracle = u.H.u + 2 q.u
        assert np.abs(H - Hr).max() <= 1e-12 * np.abs(Hr).max()
        assert np.abs(q - qr).max() <= 1e-12 * max(1.0, np.abs(qr).max())


def test_constraint_set_equals_the_references(g):
    """Every row the reference hands to subject_to (swing-zero equalities, bounded(-0, fz, 150), the four friction bounded(...)
    per leg and stage, `force_controller.py:107-162`) and nothing else."""
    m = g["opti_to_stage"]
    for k in range(g["x"].shape[0]):
        _, _, C, lo, hi = mo.build_qp(g["x"][k], g["body_ref"][k], g["p"][k], g["contact"][k])
        ref = _canon_rows(g["A"][k][:, m], g["b"][k], g["is_eq"][k])
        A2, b2, e2 = [], [], []
        for r in range(C.shape[0]):
            if lo[r] == hi[r]:
                A2.append(C[r]); b2.append(-lo[r]); e2.append(True)
                continue
            if np.isfinite(lo[r]):
                A2.append(C[r]); b2.append(-lo[r]); e2.append(False)
            if np.isfinite(hi[r]):
                A2.append(-C[r]); b2.append(hi[r]); e2.append(False)
        mine = _canon_rows(np.array(A2).reshape(-1, 60), np.array(b2), np.array(e2, dtype=bool))
        assert mine == ref, (k, g["contact"][k], len(mine), len(ref))


def test_fixture_solution_is_certified_independently(g):
    """The committed u* against the committed H, g, A, b only (no oracle code): feasible, and the gradient lies in the cone of
    the active rows (non-negative least squares from scipy) -- the KKT conditions of the reference's own QP."""
    from scipy.optimize import nnls
    for k in range(g["x"].shape[0]):
        H, gg, A, b, eq, u = g["H"][k], g["g"][k], g["A"][k], g["b"][k], g["is_eq"][k], g["u"][k]
        s = A @ u + b
        assert np.all(np.abs(s[eq]) < 1e-9) and np.all(s[~eq] > -1e-9)
        grad = 2 * H @ u + gg                       # stationarity: grad = sum lam_r A_r, lam_r >= 0 on active inequalities
        act = np.where(~eq & (s < 1e-7) & np.any(A != 0, axis=1))[0]
        cols = [A[r] for r in act] + [A[r] for r in np.where(eq)[0]] + [-A[r] for r in np.where(eq)[0]]
        if not cols:
            assert np.abs(grad).max() < 1e-9
            continue
        lam, res = nnls(np.array(cols).T, grad, maxiter=2000)
        assert res <= 1e-9 * max(1.0, np.abs(gg).max()), (k, res)
        assert g["kkt_stationarity"][k] < 1e-9 and g["kkt_primal"][k] <= 1e-10


def test_oracle_solution_and_forces_match_fixture(g):
    m = g["opti_to_stage"]
    for k in range(g["x"].shape[0]):
        f, u, info = mo.mpc_forces(g["x"][k], g["body_ref"][k], g["p"][k], g["contact"][k])
        assert np.abs(u - g["u"][k][m]).max() < 1e-7
        # sol.value(controls) is the (12, N) matrix vertcat(f1..f4) (force_controller.py:56); column 0 is what the filter applies
        assert np.abs(u.reshape(5, 12).T - g["forces"][k]).max() < 1e-7
        assert np.abs(f - g["forces"][k][:, 0]).max() < 1e-7


def test_c_oracle_predict_mpc_step_matches_reference_with_qp_forces(g):
    """x and P after predict_mpc (kalman_filter.py:153-162) with the QP's own forces: the C oracle's mode-1 predict, update
    switched off by comparing the pre-update quantities it returns."""
    Q, R = g["Q"], g["R"]
    for k in range(g["x"].shape[0]):
        x_next = co.next_state(g["x"][k], g["p"][k].copy(), g["forces"][k][:, 0])[0]
        assert np.abs(x_next - g["x_next"][k]).max() < 1e-12


def test_trajectory_with_qp_in_the_loop(g):
    """estimate_state_mpc over T = 60 with the QP solved at every step from the running state (kalman_filter.py:176-182): the
    two oracles chained (mpc_oracle forces -> C oracle mode-1 step) reproduce the reference's states, forces, rotated p, trace."""
    Q, R = g["Q"], g["R"]
    B, T = g["t_x"].shape[:2]
    for b in range(B):
        x = g["t_x0"][b].astype(np.float64).copy(); P = Q.copy()
        for t in range(T):
            f, _, info = mo.mpc_forces(x, g["t_body_ref"][b, t].astype(np.float64), g["t_p"][b, t].astype(np.float64), g["t_contact"][b, t])
            assert np.abs(f - g["t_f"][b, t]).max() < 1e-6, (b, t)
            r = co.kf_run_batch(g["t_p"][b:b + 1, t:t + 1], f.reshape(1, 1, 12), g["t_dp"][b:b + 1, t:t + 1], g["t_imu"][b:b + 1, t:t + 1],
                                g["t_contact"][b:b + 1, t:t + 1], x.reshape(1, 12), P.reshape(1, 144), Q, R,
                                body_ref=g["t_body_ref"][b:b + 1, t:t + 1], mode=1)
            x = r["x_final"][0].copy(); P = r["P_final"][0].copy()
            assert np.abs(x - g["t_x"][b, t]).max() < 1e-9, (b, t)
            assert abs(r["P_trace"][0, 0] - g["t_P_trace"][b, t]) < 1e-7 * max(1.0, g["t_P_trace"][b, t])   # P ~ 40 and ill-conditioned under the all-ones exp(dt F)
            assert np.abs(r["p_rot"][0, 0] - g["t_p_rot"][b, t]).max() < 1e-9
        assert np.abs(P.reshape(12, 12) - g["t_P_final"][b]).max() < 1e-7 * np.abs(g["t_P_final"][b]).max()
