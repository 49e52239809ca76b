"""GPU parity of the batched convex-MPC force QP (os_mpc_solve) against oracle/mpc_oracle.py (KKT-certified float64 solution of
the problem misc/force_controller.py:70-162 states; pinned to the reference's own QP assembly by G12 below).  qpOASES itself is absent."""
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


def test_two_pass_form_hands_long_problems_over_exactly(monkeypatch):
    """OS_MPC_CAP = 3: the sixteen-lanes-per-QP rows (mpc_quad.hip) give every problem up after three active-set iterations and the
    wavefront-per-QP instance continues from the hand-over record (point + faces): the same controls, the same total iteration count
    as the one-pass solve and as the wavefront-per-QP solver alone (OS_MPC_QUAD=0)."""
    from optistate_amd import Engine
    X, R, P, Cn = _problems(512, seed=21)
    res = {}
    for name, env in (("wave", {"OS_MPC_QUAD": "0"}), ("quad", {"OS_MPC_QUAD": "1", "OS_MPC_CAP": "0"}), ("two-pass", {"OS_MPC_QUAD": "1", "OS_MPC_CAP": "3"})):
        for k in ("OS_MPC_QUAD", "OS_MPC_CAP"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        r = _solve_gpu(Engine(0), X, R, P, Cn)
        res[name] = (r["u"].cpu().numpy(), r["iters"].cpu().numpy(), r["status"].cpu().numpy())
    for name in ("quad", "two-pass"):
        assert np.abs(res[name][0] - res["wave"][0]).max() < 1e-5 and np.array_equal(res[name][1], res["wave"][1]) and not res[name][2].any(), name
    assert (res["wave"][1] > 3).sum() > 50          # the second pass had work to do


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


def _oracle_mpc_filter(d, Q, R, T, B):
    """estimate_state_mpc per trajectory and step with the two oracles: QP forces (mpc_oracle) from the state before the
    predict, then one predict_mpc + update step of the C oracle (mode 1)."""
    from oracle import c_oracle as co
    xs = np.zeros((B, T, 12)); fs = np.zeros((B, T, 12))
    for b in range(B):
        x = d["x0"][b].astype(np.float64).copy(); P = Q.copy()
        for t in range(T):
            # the device path holds the state in float32 between steps
            x32 = x.astype(np.float32).astype(np.float64)
            f, _, info = mo.mpc_forces(x32, d["body_ref"][b, t].astype(np.float64), d["p"][b, t].astype(np.float64), d["contact"][b, t],
                                       **_oracle_kw())
            f32 = f.astype(np.float32).astype(np.float64)
            r = co.kf_run_batch(d["p"][b:b + 1, t:t + 1], f32.reshape(1, 1, 12), d["dp"][b:b + 1, t:t + 1], d["imu"][b:b + 1, t:t + 1],
                                d["contact"][b:b + 1, t:t + 1], x.reshape(1, 12), P.reshape(1, 144), Q, R,
                                body_ref=d["body_ref"][b:b + 1, t:t + 1], mode=1)
            x = r["x_final"][0].copy(); P = r["P_final"][0].copy()
            xs[b, t] = x; fs[b, t] = f
    return xs, fs


def _mpc_traj_inputs(B, T, seed):
    from optistate_amd.synth import synth_numpy
    d = synth_numpy(B, T, seed=seed)
    rng = np.random.default_rng(seed + 100)
    tt = np.arange(T) * 0.01
    ref = np.zeros((B, T, 12), np.float32)
    ref[:, :, 0] = 0.02 * np.sin(3 * tt); ref[:, :, 1] = 0.02 * np.cos(2 * tt); ref[:, :, 5] = 0.28; ref[:, :, 9] = 0.1
    ref += rng.normal(0, 0.005, ref.shape).astype(np.float32)
    d["body_ref"] = ref
    return d


def test_kf_mpc_run_matches_oracles(eng, monkeypatch):
    """os_kf_mpc_run = estimate_state_mpc over B x T (kalman_filter.py:176-182): forces and states against the oracles."""
    from optistate_amd.synth import Q_DEFAULT, R_DEFAULT
    monkeypatch.setattr(mo, "MASS", float(np.float32(8.8)))
    monkeypatch.setattr(mo, "INERTIA", np.asarray(np.float32([0.05530364, 0.06011944, 0.10530434]), np.float64))
    B, T = 3, 12
    d = _mpc_traj_inputs(B, T, seed=21)
    eng.set_noise(Q_DEFAULT, R_DEFAULT)
    s = {k: eng.pack(torch.as_tensor(d[k])) for k in ("p", "dp", "imu", "body_ref")}
    c = eng.pack_contact(torch.as_tensor(d["contact"]))
    x = torch.as_tensor(d["x0"].T.copy()).cuda()
    P = torch.as_tensor(np.tile(Q_DEFAULT.astype(np.float32).reshape(144, 1), (1, B))).cuda()
    r = eng.kf_mpc_run(s["p"], s["dp"], s["imu"], c, s["body_ref"], x, P, want_iters=True)
    assert int(r["status"].abs().max()) == 0
    xs, fs = _oracle_mpc_filter(d, Q_DEFAULT, R_DEFAULT, T, B)
    x_gpu = eng.unpack(r["x_out"]).cpu().numpy(); f_gpu = eng.unpack(r["f"]).cpu().numpy()
    assert np.abs(f_gpu - fs).max() < 5e-3            # N; forces ~20-40 N (float32 state feedback between steps)
    assert np.abs(x_gpu - xs).max() < 1e-4            # the state bar of the filter path
    assert int(r["iters"].min()) >= 1
    # the warm start (previous step's active set) must not change the answer
    x2 = torch.as_tensor(d["x0"].T.copy()).cuda()
    P2 = torch.as_tensor(np.tile(Q_DEFAULT.astype(np.float32).reshape(144, 1), (1, B))).cuda()
    r2 = eng.kf_mpc_run(s["p"], s["dp"], s["imu"], c, s["body_ref"], x2, P2, cold_start=True)
    assert float((r2["f"] - r["f"]).abs().max()) < 1e-3 and float((r2["x_out"] - r["x_out"]).abs().max()) < 1e-5
    # sequential scalar updates (diagonal R) instead of the batch form: same posterior
    x3 = torch.as_tensor(d["x0"].T.copy()).cuda()
    P3 = torch.as_tensor(np.tile(Q_DEFAULT.astype(np.float32).reshape(144, 1), (1, B))).cuda()
    r3 = eng.kf_mpc_run(s["p"], s["dp"], s["imu"], c, s["body_ref"], x3, P3, sequential=True, want_p_rot=True, want_trace=True)
    x3o = eng.unpack(r3["x_out"]).cpu().numpy()
    assert np.abs(x3o - xs).max() < 1e-4 and int(r3["status"].abs().max()) == 0
    assert r3["ptrace"].shape == (T, B) and bool(torch.isfinite(r3["ptrace"]).all())


def test_dropin_estimate_state_mpc_solves_the_qp(eng, monkeypatch):
    """The reference's call, no forces supplied (kalman_filter.py:176-182): the class solves the MPC on the GPU."""
    from optistate_amd import Kalman_Filter
    from optistate_amd.synth import Q_DEFAULT, R_DEFAULT
    monkeypatch.setattr(mo, "MASS", float(np.float32(8.8)))
    monkeypatch.setattr(mo, "INERTIA", np.asarray(np.float32([0.05530364, 0.06011944, 0.10530434]), np.float64))
    T = 6
    d = _mpc_traj_inputs(1, T, seed=4)
    xs, fs = _oracle_mpc_filter(d, Q_DEFAULT, R_DEFAULT, T, 1)
    kf = Kalman_Filter()
    kf.x[:] = d["x0"][0].reshape(12, 1)
    for t in range(T):
        p = d["p"][0, t].astype(np.float64).reshape(12, 1)
        x = kf.estimate_state_mpc(d["imu"][0, t].reshape(6, 1), p, d["dp"][0, t].reshape(12, 1), d["body_ref"][0, t].reshape(12, 1),
                                  d["contact"][0, t].reshape(4, 1))
        assert kf.f.shape == (12, 5)
        assert np.abs(kf.f[:, 0] - fs[0, t]).max() < 5e-3
        assert np.abs(x.ravel() - xs[0, t]).max() < 1e-4


def test_mpc_matches_committed_fixture(eng):
    """tests/golden/mpc_g9_oracle.npz (tools/gen_golden.py g9: oracle-certified solutions; includes an 'unconstrained' leg,
    contact byte 2, and force-cap cases).  The fixture was solved with the exact constants; the library holds mass and
    inertia as float32, hence the slightly wider bar than in test_mpc_matches_certified_oracle."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "mpc_g9_oracle.npz"))
    X, R, P = (np.asarray(g[k], np.float32) for k in ("x", "body_ref", "p"))
    Cn = np.asarray(g["contact"], np.uint8)
    r = _solve_gpu(eng, X, R, P, Cn)
    assert int(r["status"].abs().max()) == 0
    u = r["u"].cpu().numpy().T.astype(np.float64)
    assert np.abs(u - g["u"]).max() < 2e-3


def test_fit_noise_covariances_matches_oracle_loop(eng, monkeypatch):
    """Q/R fitting branch of data_conversion_Kalman_to_Training.py:31-104 (one batch on the device) against the same loop
    written with the oracles, with and without the script's `measurement_data.append(KF.z)` aliasing."""
    from oracle import c_oracle as co
    from optistate_amd import pipeline
    monkeypatch.setattr(mo, "MASS", float(np.float32(8.8)))
    monkeypatch.setattr(mo, "INERTIA", np.asarray(np.float32([0.05530364, 0.06011944, 0.10530434]), np.float64))
    T = 24
    d = _mpc_traj_inputs(1, T, seed=9)
    rng = np.random.default_rng(5)
    mocap = (np.tile(np.array([0, 0, 0, 0, 0, 0.28, 0, 0, 0, 0.1, 0, 0.]), (T, 1)) + rng.normal(0, 0.02, (T, 12))).astype(np.float32)
    p, dp, imu, contact = d["p"][0], d["dp"][0], d["imu"][0], d["contact"][0]
    xm, zs = [], []
    for i in range(T - 1):
        x = mocap[i].astype(np.float64)
        f, _, _ = mo.mpc_forces(x, x, p[i].astype(np.float64), contact[i], **_oracle_kw())
        xn, _ = co.next_state(x, p[i].astype(np.float64), f.astype(np.float32).astype(np.float64), dt=float(np.float32(0.01)))
        xm.append(xn)
        od = co.get_odom(p[i + 1], dp[i + 1], contact[i + 1], imu[i + 1])
        zs.append(np.concatenate([imu[i + 1][0:3], [od[0]], imu[i + 1][3:6], od[1:4]]))
    xm, zs = np.array(xm), np.array(zs)
    sel = [0, 1, 2, 5, 6, 7, 8, 9, 10, 11]
    gt1 = mocap[1:].astype(np.float64)
    for alias in (True, False):
        Q, R = pipeline.fit_noise_covariances(eng, p, dp, imu, contact, mocap, alias_measurements=alias)
        q_ref = np.var(gt1 - xm, axis=0)
        r_ref = np.var(gt1[:, sel] - (zs[-1][None, :] if alias else zs), axis=0)
        assert np.allclose(np.diag(Q), q_ref, rtol=2e-3, atol=1e-9), (np.diag(Q), q_ref)
        assert np.allclose(np.diag(R), r_ref, rtol=2e-3, atol=1e-9)
        assert np.count_nonzero(Q - np.diag(np.diag(Q))) == 0


def test_feature_rows_with_mpc_forces(eng):
    """60-column rows of data_conversion_Kalman_to_Training.py:245-254 built from the estimate_state_mpc loop: the force
    columns are KF2.f[:, 0], the p columns are the world-rotated feet (next_state mutates p in place)."""
    from oracle import c_oracle as co
    from optistate_amd import pipeline
    from optistate_amd.synth import Q_DEFAULT, R_DEFAULT
    B, T = 2, 6
    d = _mpc_traj_inputs(B, T, seed=33)
    d["ref"] = d["body_ref"]
    rows, x_hist, forces, status = pipeline.kalman_feature_rows_mpc(eng, d, Q_DEFAULT, R_DEFAULT, d["x0"])
    assert rows.shape == (B, T, 60) and int(status.abs().max()) == 0
    rows = rows.cpu().numpy(); xh = x_hist.cpu().numpy()
    assert np.array_equal(rows[:, :, 0:12], xh) and np.array_equal(rows[:, :, 18:30], forces.cpu().numpy())
    assert np.array_equal(rows[:, :, 12:18], d["accel"]) and np.array_equal(rows[:, :, 42:54], d["dp"])
    # p columns: R(x_prior) p; the prior of step t is the posterior of step t-1 (x0 for t = 0)
    for b in range(B):
        prior = d["x0"][b].astype(np.float64)
        for t in range(T):
            Rm = co.rotation(*prior[0:3])
            pw = (Rm @ d["p"][b, t].astype(np.float64).reshape(4, 3).T).T.reshape(12)
            assert np.abs(rows[b, t, 30:42] - pw).max() < 1e-5
            prior = xh[b, t].astype(np.float64)


def test_mpc_custom_weights_and_limits(eng, monkeypatch):
    """os_mpc_set_weights: other Q/R weights, friction coefficient and force cap (the reference hard-codes its own at
    kalman_filter.py:64-70 / force_controller.py:147-149) against the oracle with the same numbers."""
    from optistate_amd import Engine
    e2 = Engine(0)
    wq = np.array([5.0, 20.0, 1.0, 50.0, 80.0, 200.0, 0.5, 2.0, 1.0, 3.0, 0.2, 1.0])
    e2.mpc_set_weights(wq, r_weight=1e-4, mu=0.35, fz_max=60.0)
    monkeypatch.setattr(mo, "MASS", float(np.float32(8.8)))
    monkeypatch.setattr(mo, "INERTIA", np.asarray(np.float32([0.05530364, 0.06011944, 0.10530434]), np.float64))
    monkeypatch.setattr(mo, "MU", 0.35)
    monkeypatch.setattr(mo, "FZ_MAX", 60.0)
    X, R, P, Cn = _problems(24, seed=77)
    r = _solve_gpu(e2, X, R, P, Cn)
    assert int(r["status"].abs().max()) == 0
    u = r["u"].cpu().numpy().T.astype(np.float64)
    for k in range(X.shape[0]):
        _, u_o, info = mo.mpc_forces(X[k].astype(np.float64), R[k].astype(np.float64), P[k].astype(np.float64), Cn[k],
                                     q_weights=wq, r_weight=1e-4, **_oracle_kw())
        assert np.abs(u[k] - u_o).max() < 2e-4
    with pytest.raises(RuntimeError):
        e2.mpc_set_weights(wq, r_weight=0.0)          # R must stay positive definite


def test_mpc_nonfinite_input_terminates_and_flags(eng):
    X, R, P, Cn = _problems(8, seed=2)
    X[3, 7] = np.nan
    r = _solve_gpu(eng, X, R, P, Cn)
    st = r["status"].cpu().numpy(); u = r["u"].cpu().numpy().T
    ok = [k for k in range(8) if k != 3]
    assert np.all(st[ok] == 0) and np.all(np.isfinite(u[ok]))
    # the poisoned problem must not hang the wavefront: it ends at the iteration cap (bit 2) or with non-finite forces
    assert (st[3] & 4) or not np.all(np.isfinite(u[3]))


def test_persistent_kernel_agrees_with_the_launch_sequence(monkeypatch):
    """os_kf_mpc_run has two forms (one persistent kernel up to 32 trajectories per CU, the per-step launch sequence above):
    same forces and states on a trot with phase changes, to the state bar -- the persistent kernel carries P in float64
    between steps, the sequence rounds it to float32 at every step."""
    from optistate_amd import Engine
    from optistate_amd.synth import synth_torch, Q_DEFAULT, R_DEFAULT
    dev = torch.device("cuda:0")
    B, T = 16, 60
    d = synth_torch(B, T, dev, seed=5)
    ref = torch.zeros((T, 12, B), device=dev); ref[:, 5] = 0.28; ref[:, 9] = 0.1
    out = {}
    for mode in ("2", "0"):
        monkeypatch.setenv("OS_MPC_PERSISTENT", mode)
        e = Engine(0); e.set_noise(Q_DEFAULT, R_DEFAULT)
        contact = e.contact_soa_to_packed(d["contact"])
        x, P = d["x0"].clone(), d["P0"].clone()
        r = e.kf_mpc_run(d["p"], d["dp"], d["imu"], contact, ref, x, P, want_iters=True, want_trace=True)
        e.profile(True)
        e.kf_mpc_run(d["p"], d["dp"], d["imu"], contact, ref, d["x0"].clone(), d["P0"].clone())
        prof = e.profile_read()
        out[mode] = (r, x.clone(), prof)
    rp, xp, pp = out["2"]; rs, xs_, ps = out["0"]
    assert pp["mpc"][1] == 1 and ps["mpc"][1] >= T                              # one launch against at least one per step
    assert int(rp["status"].abs().max()) == 0 and int(rs["status"].abs().max()) == 0
    assert float((rp["x_out"] - rs["x_out"]).abs().max()) < 1e-4
    assert float((rp["f"] - rs["f"]).abs().max()) < 5e-3
    assert float((xp - xs_).abs().max()) < 1e-4
    assert float((rp["ptrace"] - rs["ptrace"]).abs().max()) < 1e-3 * float(rs["ptrace"].abs().max())


# ---- G12: the QP as the REFERENCE assembles it (tools/gen_golden_mpc.py: misc/force_controller.py:47-225 and
# kalman_filter.py:140-182 run unmodified over the evaluating casadi stand-in).  formulation = reference, solver = certified
# stand-in; tests/test_oracle_mpc_g12.py pins the oracle to the same file on the CPU.
def _g12():
    import os
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "mpc_g12_qp.npz"))


def test_mpc_solve_matches_reference_formulation_g12(eng):
    """os_mpc_solve on the 32 reference-assembled problems (all 16 contact patterns, twice): all 60 controls <= 2e-4 N of the
    minimiser of the reference's own H, g, A; sol.value(controls)[:, 0] is what predict_mpc applies (kalman_filter.py:152,161)."""
    g = _g12()
    X, R, P = (np.asarray(g[k], np.float32) for k in ("x", "body_ref", "p"))
    assert np.array_equal(X.astype(np.float64), g["x"])          # the fixture's inputs are float32-representable
    r = _solve_gpu(eng, X, R, P, np.asarray(g["contact"], np.uint8))
    assert int(r["status"].abs().max()) == 0
    u = r["u"].cpu().numpy().T.astype(np.float64)                # stage-major [12 k + 3 leg + i]
    ref = g["forces"].transpose(0, 2, 1).reshape(-1, 60)         # (12, N) matrix -> stage-major
    assert np.abs(u - ref).max() < 2e-4, np.abs(u - ref).max()
    assert np.abs(r["f"].cpu().numpy().T - g["forces"][:, :, 0]).max() < 2e-4
    # and the GPU's answer is feasible for the reference's own rows, in Opti's variable order
    m = g["opti_to_stage"]
    for k in range(X.shape[0]):
        uo = np.zeros(60); uo[m] = u[k]
        s = g["A"][k] @ uo + g["b"][k]
        assert np.all(np.abs(s[g["is_eq"][k]]) < 1e-4) and np.all(s[~g["is_eq"][k]] > -1e-4)
        cost = lambda v: v @ g["H"][k] @ v + g["g"][k] @ v
        assert cost(uo) <= cost(g["u"][k]) + 1e-6 * max(1.0, abs(cost(g["u"][k])))


def test_kf_mpc_run_matches_reference_trajectory_g12(eng):
    """os_kf_mpc_run against estimate_state_mpc run by the reference with the QP in the loop (T = 60, fitted Q/R)."""
    g = _g12()
    B, T = g["t_x"].shape[:2]
    eng.set_noise(g["Q"], g["R"])
    s = {k: eng.pack(torch.as_tensor(g["t_" + k])) for k in ("p", "dp", "imu", "body_ref")}
    c = eng.pack_contact(torch.as_tensor(g["t_contact"]))
    for kw in (dict(), dict(sequential=True)):
        x = torch.as_tensor(g["t_x0"].T.copy()).cuda()
        P = torch.as_tensor(np.tile(g["Q"].astype(np.float32).reshape(144, 1), (1, B))).cuda()
        r = eng.kf_mpc_run(s["p"], s["dp"], s["imu"], c, s["body_ref"], x, P, want_p_rot=True, want_trace=True, **kw)
        assert int(r["status"].abs().max()) == 0
        x_gpu = eng.unpack(r["x_out"]).cpu().numpy(); f_gpu = eng.unpack(r["f"]).cpu().numpy()
        assert np.abs(x_gpu - g["t_x"]).max() < 1e-4, np.abs(x_gpu - g["t_x"]).max()
        # forces: the QP is a near dead-beat controller (R = 1e-6), ~3.5e3 N per metre of state error, fed back from a float32 state
        assert np.abs(f_gpu - g["t_f"]).max() < 5e-3, np.abs(f_gpu - g["t_f"]).max()
        assert np.abs(eng.unpack(r["p_rot"]).cpu().numpy() - g["t_p_rot"]).max() < 1e-5
        pt = r["ptrace"].cpu().numpy().T
        assert np.abs(pt / g["t_P_trace"] - 1).max() < 1e-3
    from optistate_amd.synth import Q_DEFAULT, R_DEFAULT
    eng.set_noise(Q_DEFAULT, R_DEFAULT)


def test_dropin_estimate_state_mpc_matches_reference_trajectory_g12():
    """The drop-in class called exactly as the reference's loop calls it (data_conversion_Kalman_to_Training.py:136-203)."""
    from optistate_amd import Kalman_Filter
    g = _g12()
    T = g["t_x"].shape[1]
    kf = Kalman_Filter()
    kf.x = g["t_x0"][0].astype(np.float64).reshape(12, 1).copy()
    kf.P = g["Q"].copy(); kf.Q = g["Q"].copy(); kf.R = g["R"].copy()
    for t in range(T):
        p = g["t_p"][0, t].astype(np.float64).reshape(12, 1)
        x = kf.estimate_state_mpc(g["t_imu"][0, t].reshape(6, 1).astype(np.float64), p, g["t_dp"][0, t].reshape(12, 1).astype(np.float64),
                                  g["t_body_ref"][0, t].reshape(12, 1).astype(np.float64), g["t_contact"][0, t].reshape(4, 1).astype(np.float64))
        assert np.abs(x.ravel() - g["t_x"][0, t]).max() < 1e-4, t
        assert np.abs(kf.f[:, 0] - g["t_f"][0, t]).max() < 5e-3, t
        assert np.abs(p.ravel() - g["t_p_rot"][0, t]).max() < 1e-5          # p is rotated in place, as next_state does
        assert abs(kf.P_trace / g["t_P_trace"][0, t] - 1) < 1e-3


@pytest.mark.parametrize("B,T,shards,mix", [(4096, 8, "1", True), (4096, 6, "1", False), (9000, 5, "2", False), (32816, 4, "2", True), (65536, 3, "2", False)])
def test_filter_step_inside_the_qp_launch_equals_the_separate_launches(monkeypatch, B, T, shards, mix):
    """Round 6: at large batch the filter step of a trajectory runs inside the QP launch that solved its forces (mpc_quad.hip drain
    phase; the forces cross CUs through agent-scope stores and a per-trajectory mark), and a batch of two 16,384s or more runs as
    two (OS_MPC_SHARDS) concurrent parts.  Same building blocks in the same order as kf_dense_rows_kernel: x_out, f, P, status and the
    iteration counts are IDENTICAL to the separate launches (OS_MPC_FUSE_KF=0).  (More than two parts is a development setting and is
    not run here: every extra part is another stream = hardware queue of the process for its lifetime, and with four of them the GPU's
    queue scheduler no longer keeps this process's queue mapped beside another process's seconds-long kernel -- which is what
    test_gpu_contention.py's starvation scenario relies on when it runs later in the same process; tools/hog_mpc_diag.py.)
    mix: trajectories with zero and one leg on the
    ground (the one-leg instance marks, the two-leg instance steps all of them) and, in one step, with three (that step falls back
    to the separate launches, which also keeps the batch in one part)."""
    from optistate_amd import Engine
    from optistate_amd.synth import synth_torch, Q_DEFAULT, R_DEFAULT
    dev = torch.device("cuda:0")
    monkeypatch.setenv("OS_MPC_PERSISTENT", "0")
    monkeypatch.setenv("OS_MPC_SHARDS", shards)
    d = synth_torch(B, T, dev, seed=77)
    c4 = d["contact"].clone()                               # [T][4][B] uint8
    if mix:
        c4[:, :, 5::7] = 0; c4[:, 1, 5::7] = 1             # one leg
        c4[2:, :, 3::11] = 0                                # none from step 2 on
        if shards == "1":
            c4[3, :, 8::13] = 1; c4[3, 0, 8::13] = 0        # three legs in step 3
    ref = torch.zeros((T, 12, B), device=dev); ref[:, 5] = 0.28; ref[:, 9] = 0.1
    out = {}
    for fuse in ("0", "1"):
        monkeypatch.setenv("OS_MPC_FUSE_KF", fuse)
        e = Engine(0); e.set_noise(Q_DEFAULT, R_DEFAULT)
        contact = e.contact_soa_to_packed(c4)
        x, P = d["x0"].clone(), d["P0"].clone()
        r = e.kf_mpc_run(d["p"], d["dp"], d["imu"], contact, ref, x, P, want_iters=True, want_p_rot=True)
        e.profile(True)
        e.kf_mpc_run(d["p"], d["dp"], d["imu"], contact, ref, d["x0"].clone(), d["P0"].clone())
        out[fuse] = (r, x, P, e.profile_read())
    r0, x0, P0, p0 = out["0"]; r1, x1, P1, p1 = out["1"]
    for k in ("x_out", "f", "iters", "status", "p_rot"):
        assert torch.equal(r0[k], r1[k]), k
    assert torch.equal(x0, x1) and torch.equal(P0, P1)
    assert int(r1["status"].abs().max()) == 0 and int(r1["iters"].max()) > 3
    # the separate form launches the filter kernel every step, the fused form only in the step with a three-leg trajectory
    assert p0["kf"][1] == T and p1.get("kf", (0.0, 0))[1] == (1 if (mix and shards == "1") else 0)
    assert p1["mpc"][1] == T * (int(shards) if not (mix and shards == "1") else 1)


@pytest.mark.parametrize("B,T", [(4096, 12), (6000, 8), (16384, 5), (40000, 4)])
def test_a_row_per_trajectory_for_all_steps_agrees_with_the_launch_sequence(monkeypatch, B, T):
    """Round 6: batches of 8 .. 200 trajectories per CU run kf_mpc_rows_kernel -- a 16-lane row owns a trajectory for all T steps (QP ->
    filter step -> next QP, the next QP's record built in LDS from the state the filter step has just produced, the warm start in the
    row's registers) -- one wavefront per SIMD up to 4,096 trajectories, two above.  Against the per-step launch sequence
    (OS_MPC_PERSISTENT=0): the same iteration counts, states to 2e-6, forces to 1e-3 N (the sequence's numbers up to the last bits of
    the record: the same source inlined into another kernel contracts differently), P identical (it does not depend on the forces).
    Some trajectories are airborne in some steps (no QP: zero forces, the filter step only)."""
    from optistate_amd import Engine
    from optistate_amd.synth import synth_torch, Q_DEFAULT, R_DEFAULT
    dev = torch.device("cuda:0")
    d = synth_torch(B, T, dev, seed=91)
    c4 = d["contact"].clone()
    # a trot: every trajectory on one diagonal pair of legs, a third of them switching to the other pair every other step (same leg
    # COUNT and ranks: the warm start carries over by rank -- in the rows kernel it never leaves the row's registers)
    c4[:] = 0; c4[:, 0] = 1; c4[:, 3] = 1
    sw = torch.arange(B, device=dev) % 3 == 1
    for t in range(1, T, 2):
        c4[t, 0, sw] = 0; c4[t, 3, sw] = 0; c4[t, 1, sw] = 1; c4[t, 2, sw] = 1
    c4[2:4, :, 7::9] = 0                                     # airborne in steps 2 and 3
    ref = torch.zeros((T, 12, B), device=dev); ref[:, 5] = 0.28; ref[:, 9] = 0.1
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("OS_MPC_PERSISTENT", mode)
        e = Engine(0); e.set_noise(Q_DEFAULT, R_DEFAULT)
        contact = e.contact_soa_to_packed(c4)
        x, P = d["x0"].clone(), d["P0"].clone()
        r = e.kf_mpc_run(d["p"], d["dp"], d["imu"], contact, ref, x, P, want_iters=True, want_p_rot=True)
        torch.cuda.synchronize()
        out[mode] = (r, x, P, e.kernel_name("mpc"))
    rs, xs, Ps, ks = out["0"]; rr, xr, Pr, kr = out["1"]
    assert kr.startswith("kf_mpc_rows_kernel") and "quad" in ks, (kr, ks)
    assert torch.equal(rr["iters"], rs["iters"]) and int(rr["iters"].max()) > 3 and int(rr["iters"][2, 7::9].max()) == 0
    assert int(rr["status"].abs().max()) == 0 and torch.equal(rr["status"], rs["status"])
    assert float((rr["x_out"] - rs["x_out"]).abs().max()) < 2e-6 and float((xr - xs).abs().max()) < 2e-6
    assert float((rr["f"] - rs["f"]).abs().max()) < 1e-3 and float((rr["p_rot"] - rs["p_rot"]).abs().max()) < 1e-5
    assert torch.equal(Pr, Ps)
    assert float(rr["f"][2, :, 7::9].abs().max()) == 0.0


def test_rows_form_leaves_mixed_leg_counts_to_the_other_forms(monkeypatch):
    """A batch in the rows form's range with one- AND two-leg trajectories is not its case (one variable layout per launch): the
    wavefront-per-trajectory kernel takes it (B <= 32 per CU), same numbers as the launch sequence to the form-against-form bars."""
    from optistate_amd import Engine
    from optistate_amd.synth import synth_torch, Q_DEFAULT, R_DEFAULT
    dev = torch.device("cuda:0")
    B, T = 4096, 6
    d = synth_torch(B, T, dev, seed=92)
    c4 = d["contact"].clone()
    c4[:, :, 5::7] = 0; c4[:, 1, 5::7] = 1                  # one leg
    ref = torch.zeros((T, 12, B), device=dev); ref[:, 5] = 0.28; ref[:, 9] = 0.1
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("OS_MPC_PERSISTENT", mode)
        e = Engine(0); e.set_noise(Q_DEFAULT, R_DEFAULT)
        contact = e.contact_soa_to_packed(c4)
        x, P = d["x0"].clone(), d["P0"].clone()
        r = e.kf_mpc_run(d["p"], d["dp"], d["imu"], contact, ref, x, P, want_iters=True)
        torch.cuda.synchronize()
        out[mode] = (r, x, e.kernel_name("mpc"))
    assert out["1"][2] == "kf_mpc_persistent_kernel", out["1"][2]
    assert float((out["1"][0]["x_out"] - out["0"][0]["x_out"]).abs().max()) < 1e-4 and float((out["1"][0]["f"] - out["0"][0]["f"]).abs().max()) < 5e-3
    assert int(out["1"][0]["status"].abs().max()) == 0
