"""TEST INFRASTRUCTURE -- CPU restatement (numpy, float64) of the reference's convex MPC force QP.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file; the product path never does.

PARITY UNPINNED: the reference solves this QP with casadi 3.6.2 + qpOASES (environment.yml:25;
misc/force_controller.py:47,225; called from kalman_filter/kalman_filter.py:150), neither of which is present here, and
the reference holds no golden vectors for it.  What is restated here is the PROBLEM (cost, dynamics, constraints) exactly
as misc/force_controller.py:70-162 builds it; the strictly convex QP has a unique minimiser, and `solve` returns it with
a KKT certificate (stationarity / feasibility / complementarity residuals) that the tests check, so any correct solver --
qpOASES included -- must agree with it to its own tolerance.

Problem (StanceController, N = 5, dt = DT_mpc = 0.01, settings.py:5):
  variables   u_i in R^12 (3 forces x 4 legs), i = 0..N-1                              force_controller.py:52-56
  dynamics    x_{i+1} = (I + A_i dt) x_i + B_i dt u_i + dt g                           :93
              A_i, B_i from body_mpc[:, i] (angles) and p_mpc[:, i]                    :77-88, :181-222
              body_mpc[:, 0] = current x, body_mpc[:, 1:] = body_ref; p_mpc[:, :] = p  kalman_filter.py:141-146
  cost        sum_i (x_{i+1} - ref_{i+1})^T Q (x_{i+1} - ref_{i+1}) + u_i^T R u_i      :98-105  (terminal P = Q, kalman_filter.py:72)
  constraints contact == 0: u_leg = 0; contact == 1: 0 <= fz <= 150, |fx| <= mu fz, |fy| <= mu fz, mu = 0.6   :107-162
The filter uses column 0 of the solution (kalman_filter.py:161).
"""
import numpy as np

N_HORIZON = 5
DT = 0.01
MASS = 8.8
INERTIA = np.array([55303643.08, 60119440.34, 105304340.05]) / 1e9
MU = 0.6
FZ_MAX = 150.0
Q_WEIGHTS = np.array([10.0, 10.0, 10.0, 100.0, 100.0, 100.0, 1.0, 1.0, 5.0, 1.0, 1.0, 1.0])   # kalman_filter.py:64
R_WEIGHT = 1e-6                                                                                 # kalman_filter.py:66
GRAV = np.array([0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, -9.81])


def rotation(thx, thy, thz):
    """Rz Ry Rx (force_controller.py:170-179)."""
    cx, sx, cy, sy, cz, sz = np.cos(thx), np.sin(thx), np.cos(thy), np.sin(thy), np.cos(thz), np.sin(thz)
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1.0]])
    Ry = np.array([[cy, 0, sy], [0, 1.0, 0], [-sy, 0, cy]])
    Rx = np.array([[1.0, 0, 0], [0, cx, -sx], [0, sx, cx]])
    return Rz @ Ry @ Rx


def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


def dyn_matrices(angles, p):
    """A (12x12), B (12x12) of force_controller.py:181-222 for one horizon step (float, NOT the truncated int64 A of next_state)."""
    R = rotation(*angles)
    A = np.zeros((12, 12)); A[0:3, 6:9] = R.T; A[3:6, 9:12] = np.eye(3)
    Ihat_inv = np.linalg.inv(R @ np.diag(INERTIA) @ R.T)
    B = np.zeros((12, 12))
    for j in range(4):
        B[6:9, 3 * j:3 * j + 3] = Ihat_inv @ skew(R @ p[3 * j:3 * j + 3])
        B[9:12, 3 * j:3 * j + 3] = np.eye(3) / MASS
    return A, B


def build_qp(x, body_ref, p, contact, q_weights=Q_WEIGHTS, r_weight=R_WEIGHT, dt=DT, n=N_HORIZON):
    """Dense condensed QP  min u^T H u + 2 q^T u  (u = [u_0; ...; u_{n-1}], 12 n variables) and its constraints.

    Returns H, q, C, lo, hi with lo <= C u <= hi."""
    x = np.asarray(x, float).reshape(12); body_ref = np.asarray(body_ref, float).reshape(12)
    p = np.asarray(p, float).reshape(12); contact = np.asarray(contact, float).reshape(4)
    nv = 12 * n
    W = np.diag(q_weights)
    # affine map of the states: x_k = Phi_k + S_k u
    Phi = x.copy(); S = np.zeros((12, nv))
    H = r_weight * np.eye(nv); q = np.zeros(nv)
    for i in range(n):
        ang = x[0:3] if i == 0 else body_ref[0:3]
        A, B = dyn_matrices(ang, p)
        Ad = np.eye(12) + A * dt
        Phi = Ad @ Phi + dt * GRAV
        S = Ad @ S
        S[:, 12 * i:12 * i + 12] += B * dt
        e = Phi - body_ref
        H += S.T @ W @ S
        q += S.T @ W @ e
    rows, lo, hi = [], [], []
    for k in range(n):
        for j in range(4):
            ix, iy, iz = 12 * k + 3 * j, 12 * k + 3 * j + 1, 12 * k + 3 * j + 2
            def row(pairs):
                r = np.zeros(nv)
                for idx, v in pairs: r[idx] = v
                return r
            if contact[j] == 0:          # swing: force zero (:114-123)
                for idx in (ix, iy, iz):
                    rows.append(row([(idx, 1.0)])); lo.append(0.0); hi.append(0.0)
            if contact[j] == 1:          # stance: friction pyramid (:144-155)
                rows.append(row([(iz, 1.0)])); lo.append(0.0); hi.append(FZ_MAX)
                rows.append(row([(ix, 1.0), (iz, -MU)])); lo.append(-np.inf); hi.append(0.0)
                rows.append(row([(ix, 1.0), (iz, MU)])); lo.append(0.0); hi.append(np.inf)
                rows.append(row([(iy, 1.0), (iz, -MU)])); lo.append(-np.inf); hi.append(0.0)
                rows.append(row([(iy, 1.0), (iz, MU)])); lo.append(0.0); hi.append(np.inf)
    C = np.array(rows) if rows else np.zeros((0, nv))
    return H, q, C, np.array(lo), np.array(hi)


def _kkt_solve(H, q, C, act_lo, act_hi, lo, hi):
    """Equality-constrained QP on a fixed active set: min u^T H u + 2 q^T u  s.t. C_a u = b_a."""
    idx = np.concatenate([np.where(act_lo)[0], np.where(act_hi & ~act_lo)[0]])
    b = np.concatenate([lo[act_lo], hi[act_hi & ~act_lo]])
    Ca = C[idx]
    # drop linearly dependent rows (e.g. fx = +-mu fz with fz = 0)
    keep = []
    Qr = np.zeros((0, C.shape[1]))
    for r in range(Ca.shape[0]):
        v = Ca[r] - (Qr.T @ (Qr @ Ca[r]) if len(Qr) else 0)
        if np.linalg.norm(v) > 1e-9:
            Qr = np.vstack([Qr, v / np.linalg.norm(v)]); keep.append(r)
    Ca, b, idx = Ca[keep], b[keep], idx[keep]
    nv, na = H.shape[0], Ca.shape[0]
    K = np.block([[2 * H, Ca.T], [Ca, np.zeros((na, na))]])
    sol = np.linalg.solve(K, np.concatenate([-2 * q, b]))
    return sol[:nv], sol[nv:], idx


def solve(H, q, C, lo, hi, max_iter=200):
    """Primal active-set iteration started from an ADMM estimate; returns (u, info) with a KKT certificate.

    info: {'stationarity', 'primal', 'dual', 'iters'} -- residuals of 2(Hu+q) + C^T lam = 0, lo <= Cu <= hi and the
    multiplier signs.  Raises if the certificate is not met (the tests rely on that)."""
    nv = H.shape[0]
    m = C.shape[0]
    # --- ADMM (OSQP-style splitting) for a good active-set guess ---
    sigma, rho = 1e-9, 1e-3
    M = np.linalg.inv(2 * H + sigma * np.eye(nv) + rho * C.T @ C)
    u = np.zeros(nv); z = np.zeros(m); y = np.zeros(m)
    for it in range(4000):
        u = M @ (sigma * u - 2 * q + C.T @ (rho * z - y))
        Cu = C @ u
        z = np.clip(Cu + y / rho, lo, hi)
        y = y + rho * (Cu - z)
    tol = 1e-7
    act_lo = (np.abs(z - lo) < tol) & (y < 0) | (lo == hi)
    act_hi = (np.abs(z - hi) < tol) & (y > 0) | (lo == hi)
    for it in range(max_iter):
        u, lam, idx = _kkt_solve(H, q, C, act_lo, act_hi, lo, hi)
        Cu = C @ u
        viol_lo = lo - Cu; viol_hi = Cu - hi
        worst = max(viol_lo.max(initial=0), viol_hi.max(initial=0))
        if worst > 1e-10:
            # add the most violated constraint
            if viol_lo.max(initial=0) >= viol_hi.max(initial=0): act_lo[np.argmax(viol_lo)] = True
            else: act_hi[np.argmax(viol_hi)] = True
            continue
        # multiplier signs: for C u >= lo the multiplier must be <= 0 in this sign convention, for C u <= hi >= 0
        lam_full = np.zeros(m); lam_full[idx] = lam
        bad = []
        for r in idx:
            if lo[r] == hi[r]: continue
            if act_lo[r] and lam_full[r] > 1e-12: bad.append((lam_full[r], r, 'lo'))
            if act_hi[r] and not act_lo[r] and lam_full[r] < -1e-12: bad.append((-lam_full[r], r, 'hi'))
        if not bad:
            break
        _, r, side = max(bad)
        if side == 'lo': act_lo[r] = False
        else: act_hi[r] = False
    else:
        raise RuntimeError("mpc_oracle.solve: active-set iteration did not terminate")
    stat = np.abs(2 * (H @ u + q) + C.T @ lam_full).max()
    scale = max(1.0, np.abs(2 * q).max())
    info = {"stationarity": stat / scale, "primal": worst, "iters": it + 1, "active": int((act_lo | act_hi).sum())}
    if info["stationarity"] > 1e-9:
        raise RuntimeError(f"mpc_oracle.solve: KKT certificate failed {info}")
    return u, info


def mpc_forces(x, body_ref, p, contact, **kw):
    """Column 0 of the optimal controls (kalman_filter.py:152,161): the 12 ground-reaction forces applied at this step."""
    H, q, C, lo, hi = build_qp(x, body_ref, p, contact, **kw)
    u, info = solve(H, q, C, lo, hi)
    return u[:12].copy(), u, info
