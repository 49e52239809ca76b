"""GPU parity: HIP Kalman kernels (through the C-ABI) vs golden vectors from the reference and vs the CPU oracle."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu

STATE_TOL = 1e-4       # BASELINE.json north_star: state vector l_inf < 1e-4 (fp32 kernels vs float64 reference)


@pytest.fixture(scope="module")
def eng():
    from optistate_amd import Engine
    return Engine(0)


def soa(eng, g, keys=("p", "f", "dp", "imu")):
    d = {k: eng.pack(torch.as_tensor(np.asarray(g[k], dtype=np.float32))) for k in keys}
    d["contact"] = eng.pack_contact(torch.as_tensor(np.asarray(g["contact"])))
    return d


def run(eng, g, Q, R, B, **kw):
    d = soa(eng, g)
    eng.set_noise(Q, R)
    x = torch.as_tensor(np.asarray(g["x0"], dtype=np.float32).T.copy()).cuda()
    P = torch.as_tensor(np.tile(np.asarray(Q, dtype=np.float32).reshape(144, 1), (1, B))).cuda()
    r = eng.kf_run(d["p"], d["f"], d["dp"], d["imu"], d["contact"], x, P, **kw)
    torch.cuda.synchronize()
    r["x_final"], r["P_final"] = x, P
    return r


# kernel families: batch (Cholesky), sequential full-P lanes, sequential symmetric lanes, 16-lanes-per-trajectory rows
# + the north_star's literal layout (one wavefront per trajectory, P in LDS, float64): built to be measured, never the default
VARIANTS = [dict(sequential=False, symmetric=False), dict(sequential=True, symmetric=False, lane_per_trajectory=True),
            dict(sequential=True, symmetric=True, lane_per_trajectory=True), dict(sequential=True, symmetric=True),
            dict(sequential=False, symmetric=False, wave_per_trajectory=True)]
VIDS = ["batch", "seq-lanes", "sym-lanes", "rows", "wave-per-trajectory"]


@pytest.mark.parametrize("s", [0, 1])
@pytest.mark.parametrize("v", VARIANTS, ids=VIDS)
def test_g3_trajectory_matches_reference(eng, s, v):
    g = load_golden("kf_g3_traj.npz")
    sequential = v["sequential"]
    r = run(eng, g, g[f"Q{s}"], g[f"R{s}"], 2, want_p_rot=True, want_trace=True, want_gain=True, **v)
    xo = eng.unpack(r["x_out"]).cpu().numpy()
    pr = eng.unpack(r["p_rot"]).cpu().numpy()
    for b in range(2):
        assert np.abs(xo[b] - g[f"s{s}_b{b}_x"]).max() < STATE_TOL
        assert np.abs(pr[b] - g[f"s{s}_b{b}_p_rot"]).max() < 1e-5
        ptr = r["P_trace"].cpu().numpy()[:, b]
        assert np.abs(ptr / g[f"s{s}_b{b}_P_trace"] - 1).max() < 1e-3
        # K_gain = np.trace(K) (kalman_filter.py:174) from every family: the batch form sums the K it built, the sequential /
        # symmetric / 16-lane forms evaluate trace(P+ H^T R^-1) on their posterior (they never form K)
        kg = r["K_gain"].cpu().numpy()[:, b]
        # bar: ABSOLUTE 1e-4, the state bar (K_gain is a sum of ten gains, ~2 here; measured 5e-8 ... 5e-7 for every family under the
        # default noise and 1e-7 (float64 families) / 3.3e-5 (float32 posterior / R = 1e-4) under the fitted set: tools/kgain_error.py;
        # until round 5 this was a relative 1e-3)
        assert np.abs(kg - g[f"s{s}_b{b}_K_gain"]).max() < 1e-4, (np.abs(kg - g[f"s{s}_b{b}_K_gain"]).max(), np.abs(g[f"s{s}_b{b}_K_gain"]).max())
        Pf = r["P_final"].cpu().numpy()[:, b].reshape(12, 12)
        ref = g[f"s{s}_b{b}_P_final"]
        assert np.abs(Pf - ref).max() < 1e-3 * np.abs(ref).max()
    assert int(r["status"].abs().sum()) == 0


@pytest.mark.parametrize("s", [0, 1])
@pytest.mark.parametrize("v", VARIANTS, ids=VIDS)
def test_g4_batch_matches_reference(eng, s, v):
    g = load_golden("kf_g4_batch.npz")
    B = g["p"].shape[0]
    r = run(eng, g, g[f"Q{s}"], g[f"R{s}"], B, **v)
    xo = eng.unpack(r["x_out"]).cpu().numpy()
    err = np.abs(xo - g[f"s{s}_x"]).max()
    assert err < STATE_TOL, err


def test_truncation_quirk_theta_zero_start(eng):
    """x0 with theta == 0 exactly: the reference integrates omega into theta on the first predict only
    (int64 A block == I); fp32 cos() would keep doing so for |theta| < 3e-4."""
    g = load_golden("kf_g2_next_state.npz")
    n = g["x"].shape[0]
    x = torch.as_tensor(np.asarray(g["x"], dtype=np.float32).T.copy()).cuda()
    P = torch.zeros((144, n), dtype=torch.float32).cuda()
    p = torch.as_tensor(np.asarray(g["p"], dtype=np.float32).T.copy()).cuda()
    f = torch.as_tensor(np.asarray(g["f"], dtype=np.float32).T.copy()).cuda()
    from optistate_amd.engine import _ptr
    eng._check(eng.lib.os_kf_predict(eng._h, n, _ptr(p), _ptr(f), None, _ptr(x), _ptr(P), None, 0, eng._stream()),
               "os_kf_predict")
    torch.cuda.synchronize()
    xn = x.cpu().numpy().T
    # inputs were rounded to fp32, so recompute the expectation with the oracle on the rounded inputs
    from oracle import c_oracle as orc
    worst = 0.0
    for i in range(n):
        ref, prot = orc.next_state(np.float32(g["x"][i]).astype(np.float64), np.float32(g["p"][i]).astype(np.float64),
                                   np.float32(g["f"][i]).astype(np.float64))
        worst = max(worst, np.abs(xn[i] - ref).max())
        assert np.abs(p.cpu().numpy().T[i] - prot).max() < 1e-5
    assert worst < 2e-5, worst


def test_large_batch_vs_oracle(eng):
    """Synthetic B=2048, T=50 against the float64 C oracle (same fp32-rounded inputs)."""
    from optistate_amd.synth import synth_numpy, Q_FITTED, R_FITTED
    from oracle import c_oracle as orc
    B, T = 2048, 50
    d = synth_numpy(B, T, seed=5)
    ref = orc.kf_run_batch(d["p"], d["f"], d["dp"], d["imu"], d["contact"], d["x0"], np.tile(Q_FITTED, (B, 1, 1)),
                           Q_FITTED, R_FITTED, aux=False)
    g = dict(d)
    for v in VARIANTS:
        r = run(eng, g, Q_FITTED, R_FITTED, B, **v)
        xo = eng.unpack(r["x_out"]).cpu().numpy()
        err = np.abs(xo - ref["x"]).max()
        assert err < STATE_TOL, (v, err)
        Pf = r["P_final"].cpu().numpy().T.reshape(B, 12, 12)
        assert np.abs(Pf - ref["P_final"]).max() < 1e-3 * np.abs(ref["P_final"]).max()
        assert int(r["status"].abs().sum()) == 0


def test_dense_fd_variant_matches_reference_g8(eng):
    g = load_golden("kf_g8_mpc.npz")
    d = soa(eng, g)
    br = eng.pack(torch.as_tensor(np.asarray(g["body_ref"], dtype=np.float32)))
    eng.set_noise(g["Q"], g["R"])
    x = torch.as_tensor(np.asarray(g["x0"], dtype=np.float32).T.copy()).cuda()
    P = torch.as_tensor(np.tile(np.asarray(g["Q"], dtype=np.float32).reshape(144, 1), (1, 2))).cuda()
    r = eng.kf_run(d["p"], d["f"], d["dp"], d["imu"], d["contact"], x, P, body_ref=br, dense_fd=True, sequential=False)
    xo = eng.unpack(r["x_out"]).cpu().numpy()
    for b in range(2):
        err = np.abs(xo[b] - g[f"b{b}_x"]).max()
        # the dense ~all-ones F_d makes P ill-conditioned between predict and update: that variant's covariance
        # arithmetic runs in float64 inside the kernel, so the usual bar holds
        assert err < STATE_TOL, err


def test_status_flags_nonfinite_input(eng):
    from optistate_amd.synth import synth_numpy, Q_DEFAULT, R_DEFAULT
    d = synth_numpy(64, 4, seed=9)
    d["imu"][3, 1, 0] = np.nan
    for v in VARIANTS:
        r = run(eng, d, Q_DEFAULT, R_DEFAULT, 64, **v)
        st = r["status"].cpu().numpy()
        assert st[3] != 0 and (np.delete(st, 3) == 0).all(), v


def test_pack_unpack_roundtrip(eng):
    a = torch.randn(37, 11, 12)
    s = eng.pack(a)
    assert s.shape == (11, 12, 37)
    assert torch.equal(s.cpu(), a.permute(1, 2, 0))
    assert torch.equal(eng.unpack(s).cpu(), a)


DROPIN_TOL = 1e-8      # the drop-in class computes in float64 (os_kf_step): reference precision, not the batched kernels' 1e-4


def test_dropin_kalman_filter_class(eng):
    """The reference's call sequence on the drop-in class (one os_kf_step launch per method, float64 on one wavefront)."""
    from optistate_amd import Kalman_Filter
    g = load_golden("kf_g3_traj.npz")
    for s in (0, 1):
        kf = Kalman_Filter()
        kf.x[:] = g["x0"][0].reshape(12, 1)
        kf.Q = g[f"Q{s}"].copy(); kf.R = g[f"R{s}"].copy(); kf.P = g[f"Q{s}"].copy()
        for t in range(40):
            p = g["p"][0, t].astype(np.float64).reshape(12, 1)
            z_before = kf.z.copy()
            od = kf.get_odom(p, g["dp"][0, t].reshape(12, 1), g["contact"][0, t].reshape(4, 1), g["imu"][0, t].reshape(6, 1))
            assert od.shape == (4, 1) and np.array_equal(kf.z, z_before)        # get_odom returns, set_measurements stores
            kf.set_measurements(g["imu"][0, t].reshape(6, 1), od)
            kf.predict(p, g["f"][0, t].reshape(12, 1))
            assert np.abs(p.ravel() - g[f"s{s}_b0_p_rot"][t]).max() < DROPIN_TOL  # p mutated in place
            assert np.abs(kf.x_model.ravel() - g[f"s{s}_b0_x_prior"][t]).max() < DROPIN_TOL
            kf.update()
            assert kf.x.shape == (12, 1) and kf.x.dtype == np.float64 and kf.P.shape == (12, 12)
            assert np.abs(kf.x.ravel() - g[f"s{s}_b0_x"][t]).max() < DROPIN_TOL
            assert abs(kf.K_gain - g[f"s{s}_b0_K_gain"][t]) < DROPIN_TOL
            assert abs(kf.P_trace / g[f"s{s}_b0_P_trace"][t] - 1) < 1e-9
            if t in (0, 1):
                assert np.abs(kf.K - g[f"s{s}_b0_K{t}"]).max() < DROPIN_TOL      # the 12x10 gain itself


def test_dropin_fused_step_is_the_four_call_sequence(eng):
    """Kalman_Filter.step = get_odom + set_measurements + predict + update in ONE launch: same numbers as the four calls
    (bit for bit: the same kernel code runs either way) and as the reference's trajectory (G3, both noise sets, 200 steps)."""
    from optistate_amd import Kalman_Filter
    g = load_golden("kf_g3_traj.npz")
    for s in (0, 1):
        for b in (0, 1):
            kf, kf4 = Kalman_Filter(), Kalman_Filter()
            for k in (kf, kf4):
                k.x[:] = g["x0"][b].reshape(12, 1)
                k.Q = g[f"Q{s}"].copy(); k.R = g[f"R{s}"].copy(); k.P = g[f"Q{s}"].copy()
            for t in range(g["p"].shape[1]):
                a = lambda key, n: g[key][b, t].astype(np.float64).reshape(n, 1)
                p = a("p", 12)
                x = kf.step(p, a("f", 12), a("dp", 12), a("imu", 6), g["contact"][b, t].reshape(4, 1))
                assert x is kf.x
                assert np.abs(x.ravel() - g[f"s{s}_b{b}_x"][t]).max() < DROPIN_TOL, (s, b, t)
                assert np.abs(p.ravel() - g[f"s{s}_b{b}_p_rot"][t]).max() < DROPIN_TOL
                if t < 25:
                    p4 = a("p", 12)
                    kf4.set_measurements(a("imu", 6), kf4.get_odom(p4, a("dp", 12), g["contact"][b, t].reshape(4, 1), a("imu", 6)))
                    kf4.predict(p4, a("f", 12)); kf4.update()
                    assert np.array_equal(kf4.x, kf.x) and np.array_equal(kf4.P, kf.P) and np.array_equal(p4, p)
            assert abs(kf.P_trace / g[f"s{s}_b{b}_P_trace"][-1] - 1) < 1e-9
            assert np.abs(kf.P - g[f"s{s}_b{b}_P_final"]).max() < 1e-9 * np.abs(g[f"s{s}_b{b}_P_final"]).max()


def test_dropin_update_raises_like_numpy_on_a_singular_s(eng):
    """np.linalg.inv raises LinAlgError on a singular S (kalman_filter.py:168); so does the drop-in (status bit 0)."""
    from optistate_amd import Kalman_Filter
    kf = Kalman_Filter()
    kf.P = np.zeros((12, 12)); kf.R = np.zeros((10, 10))
    with pytest.raises(np.linalg.LinAlgError):
        kf.update()


def _spd(rng, n, scale):
    A = rng.normal(size=(n, n))
    return (A @ A.T / n + np.eye(n)) * scale


@pytest.mark.parametrize("v", VARIANTS, ids=VIDS)
def test_full_process_noise_matrix(eng, v):
    """Non-diagonal Q (the reference's are diagonal, the interface is not): every kernel family's general-Q instantiation."""
    from optistate_amd.synth import synth_numpy, R_FITTED
    from oracle import c_oracle as orc
    rng = np.random.default_rng(12)
    Q = _spd(rng, 12, 1e-3)
    B, T = 96, 40
    d = synth_numpy(B, T, seed=6)
    ref = orc.kf_run_batch(d["p"], d["f"], d["dp"], d["imu"], d["contact"], d["x0"], np.tile(Q, (B, 1, 1)), Q, R_FITTED, aux=False)
    r = run(eng, d, Q, R_FITTED, B, **v)
    assert np.abs(eng.unpack(r["x_out"]).cpu().numpy() - ref["x"]).max() < STATE_TOL
    Pf = r["P_final"].cpu().numpy().T.reshape(B, 12, 12)
    assert np.abs(Pf - ref["P_final"]).max() < 1e-3 * np.abs(ref["P_final"]).max()


def test_full_measurement_noise_matrix_uses_batch_update(eng):
    """Non-diagonal R: only the batch (Cholesky) form applies; the sequential forms must refuse it."""
    from optistate_amd.synth import synth_numpy, Q_DEFAULT
    from oracle import c_oracle as orc
    rng = np.random.default_rng(13)
    R = _spd(rng, 10, 1e-2)
    B, T = 64, 30
    d = synth_numpy(B, T, seed=7)
    ref = orc.kf_run_batch(d["p"], d["f"], d["dp"], d["imu"], d["contact"], d["x0"], np.tile(Q_DEFAULT, (B, 1, 1)), Q_DEFAULT, R)
    r = run(eng, d, Q_DEFAULT, R, B, want_gain=True)                  # sequential=None -> batch, because R is not diagonal
    assert np.abs(eng.unpack(r["x_out"]).cpu().numpy() - ref["x"]).max() < STATE_TOL
    assert np.abs(r["K_gain"].cpu().numpy().T - ref["K_gain"]).max() < 1e-3
    with pytest.raises(RuntimeError):
        run(eng, d, Q_DEFAULT, R, B, sequential=True)


def test_dropin_estimate_state_mpc_with_supplied_forces(eng):
    """estimate_state_mpc on the drop-in class with the QP's forces passed in (G8: the reference run with primed forces)."""
    from optistate_amd import Kalman_Filter
    g = load_golden("kf_g8_mpc.npz")
    kf = Kalman_Filter()
    kf.x[:] = g["x0"][0].reshape(12, 1)
    kf.Q = g["Q"].copy(); kf.R = g["R"].copy(); kf.P = g["Q"].copy()
    for t in range(8):
        p = g["p"][0, t].astype(np.float64).reshape(12, 1)
        x = kf.estimate_state_mpc(g["imu"][0, t].reshape(6, 1), p, g["dp"][0, t].reshape(12, 1), g["body_ref"][0, t].reshape(12, 1),
                                  g["contact"][0, t].reshape(4, 1), f=g["f"][0, t])
        assert x is kf.x
        assert np.abs(x.ravel() - g["b0_x"][t]).max() < 1e-7              # float64 step; the dense F_d amplifies rounding ~1e4 x
        assert np.abs(p.ravel() - g["b0_p_rot"][t]).max() < DROPIN_TOL


@pytest.mark.parametrize("v", VARIANTS, ids=VIDS)
def test_long_horizon_config1_length(eng, v):
    """T = 4063 steps (one full trajectory of the reference pipeline: settings.py:15-16 cut-offs 430..4494): the float32 filter
    must not drift away from the float64 reference over the whole horizon."""
    from optistate_amd.synth import synth_numpy, Q_FITTED, R_FITTED
    from oracle import c_oracle as orc
    B, T = 24, 4063
    d = synth_numpy(B, T, seed=77)
    ref = orc.kf_run_batch(d["p"], d["f"], d["dp"], d["imu"], d["contact"], d["x0"], np.tile(Q_FITTED, (B, 1, 1)), Q_FITTED,
                           R_FITTED, aux=False)
    r = run(eng, d, Q_FITTED, R_FITTED, B, **v)
    err = np.abs(eng.unpack(r["x_out"]).cpu().numpy() - ref["x"])
    assert err.max() < STATE_TOL, err.max()
    assert err[:, -500:].max() < 2 * err[:, :500].max() + 1e-5          # no growth with time


def test_rotation_accuracy_for_large_angles(eng):
    """The kernels' branch-free sincos (Cody-Waite by pi/2 + minimax) against the float64 oracle for yaw/roll/pitch up to
    +-60 rad (unwrapped yaw), through the odometry kernel (z[7:10] = R(imu theta) . v_body)."""
    from oracle import c_oracle as orc
    from optistate_amd.engine import _ptr
    rng = np.random.default_rng(4)
    n = 4096
    th = rng.uniform(-60.0, 60.0, (n, 3)); th[:64] = rng.uniform(-1e-3, 1e-3, (64, 3)); th[64:128, 2] = np.pi * rng.integers(-8, 9, 64) / 2
    imu = np.concatenate([th, rng.normal(0, 0.5, (n, 3))], axis=1).astype(np.float32)
    p = rng.normal(0, 0.3, (n, 12)).astype(np.float32); dp = rng.normal(0, 0.5, (n, 12)).astype(np.float32)
    contact = np.zeros((n, 4), dtype=np.uint8)
    for i in range(n):
        contact[i, rng.permutation(4)[:1 + i % 3]] = 1
    packed = torch.as_tensor(contact.view(np.int32).reshape(n).copy()).cuda()
    z = torch.empty((10, n), dtype=torch.float32, device="cuda")
    up = lambda a: torch.as_tensor(a.T.copy()).cuda()
    pt, dpt, it = up(p), up(dp), up(imu)
    eng._check(eng.lib.os_kf_odom(eng._h, n, _ptr(pt), _ptr(dpt), _ptr(packed), _ptr(it), _ptr(z), eng._stream()), "os_kf_odom")
    zz = z.cpu().numpy().T
    worst = 0.0
    for i in range(n):
        od = orc.get_odom(p[i].astype(np.float64), dp[i].astype(np.float64), contact[i], imu[i].astype(np.float64))
        worst = max(worst, abs(zz[i, 3] - od[0]), np.abs(zz[i, 7:10] - od[1:]).max())
    assert worst < 2e-6, worst


@pytest.mark.parametrize("B,T", [(1, 1), (65, 1), (65, 2), (130, 3), (200, 7)])
def test_sym_lane_kernel_short_and_ragged(eng, B, T):
    """kf_run_sym_kernel takes its step inputs through an LDS-DMA double buffer requested one step ahead: the first step,
    T = 1 (nothing to prefetch), odd / even buffer parity and partly filled wavefronts against the float64 C oracle."""
    from optistate_amd.synth import synth_numpy, Q_FITTED, R_FITTED
    from oracle import c_oracle as orc
    d = synth_numpy(B, T, seed=11 + B + T)
    ref = orc.kf_run_batch(d["p"], d["f"], d["dp"], d["imu"], d["contact"], d["x0"], np.tile(Q_FITTED, (B, 1, 1)),
                           Q_FITTED, R_FITTED, aux=False)
    r = run(eng, dict(d), Q_FITTED, R_FITTED, B, sequential=True, symmetric=True, lane_per_trajectory=True)
    xo = eng.unpack(r["x_out"]).cpu().numpy()
    assert np.abs(xo - ref["x"]).max() < STATE_TOL
    Pf = r["P_final"].cpu().numpy().T.reshape(B, 12, 12)
    assert np.abs(Pf - ref["P_final"]).max() < 1e-3 * np.abs(ref["P_final"]).max()
    assert int(r["status"].abs().sum()) == 0


@pytest.mark.parametrize("s", [0, 1])
@pytest.mark.parametrize("sequential", [False, True], ids=["batch", "sequential"])
def test_update_entry_point_gain_matches_reference(eng, s, sequential):
    """os_kf_odom -> os_kf_predict -> os_kf_update through the C-ABI on step 0 of G3: K (12 x 10), K_gain = np.trace(K) and the
    posterior against the reference's (kalman_filter.py:164-174).  The sequential form never builds K during the update: K and
    K_gain come from the posterior, K = P+ H^T R^-1 -- the same matrix for the optimal gain."""
    g = load_golden("kf_g3_traj.npz")
    Q, R = g[f"Q{s}"], g[f"R{s}"]
    eng.set_noise(Q, R)
    col = lambda k, n: torch.as_tensor(np.ascontiguousarray(np.asarray(g[k][:, 0, :n], dtype=np.float32).T)).cuda()
    p, f, dp, imu = col("p", 12), col("f", 12), col("dp", 12), col("imu", 6)
    c = torch.as_tensor(np.asarray(g["contact"][:, 0])).cuda().contiguous().view(torch.int32).reshape(-1)
    x = torch.as_tensor(np.asarray(g["x0"], dtype=np.float32).T.copy()).cuda()
    P = torch.as_tensor(np.tile(np.asarray(Q, dtype=np.float32).reshape(144, 1), (1, 2))).cuda()
    z = eng.kf_odom(p, dp, c, imu)
    eng.kf_predict(p, f, x, P)
    r = eng.kf_update(z, x, P, sequential=sequential, want_K=True)
    torch.cuda.synchronize()
    assert int(r["status"].abs().sum()) == 0
    for b in range(2):
        K = r["K"].cpu().numpy()[:, b].reshape(12, 10)
        Kref = g[f"s{s}_b{b}_K0"]
        assert np.abs(K - Kref).max() < 2e-5 * max(1.0, np.abs(Kref).max()), np.abs(K - Kref).max()
        assert abs(float(r["K_gain"][b]) - g[f"s{s}_b{b}_K_gain"][0]) < 1e-4 * max(1.0, abs(g[f"s{s}_b{b}_K_gain"][0]))
        assert abs(float(r["P_trace"][b]) / g[f"s{s}_b{b}_P_trace"][0] - 1) < 1e-4
        assert np.abs(x.cpu().numpy()[:, b] - g[f"s{s}_b{b}_x"][0]).max() < STATE_TOL


@pytest.mark.parametrize("B", [1, 5, 37, 4100])
@pytest.mark.parametrize("sequential", [False, True], ids=["batch", "seq"])
def test_dense_rows_kernel_matches_oracle_predict_mpc(eng, B, sequential):
    """kf_dense_rows_kernel (float64, 16 lanes per trajectory) against the C oracle's predict_mpc path (kalman_filter.py:
    153-174 with supplied forces) for both update forms, ragged batch sizes (partial waves / workgroups), P_trace, K_gain, the
    rotated foot positions and the final state; the reference-generated G8 pins the same path above."""
    from oracle import c_oracle as orc
    from optistate_amd.synth import synth_numpy, Q_FITTED, R_FITTED
    T = 40
    d = synth_numpy(B, T, seed=100 + B)
    rng = np.random.default_rng(B)
    d["body_ref"] = np.zeros((B, T, 12), dtype=np.float32)
    d["body_ref"][..., 0:3] = d["imu"][..., 0:3] + rng.normal(0, 0.01, (B, T, 3)).astype(np.float32)
    ref = orc.kf_run_batch(d["p"], d["f"], d["dp"], d["imu"], d["contact"], d["x0"], np.tile(Q_FITTED, (B, 1, 1)), Q_FITTED, R_FITTED,
                           body_ref=d["body_ref"], mode=1)
    s = soa(eng, d)
    br = eng.pack(torch.as_tensor(d["body_ref"]))
    eng.set_noise(Q_FITTED, R_FITTED)
    x = torch.as_tensor(d["x0"].T.copy()).cuda()
    P = torch.as_tensor(np.tile(Q_FITTED.astype(np.float32).reshape(144, 1), (1, B))).cuda()
    r = eng.kf_run(s["p"], s["f"], s["dp"], s["imu"], s["contact"], x, P, body_ref=br, dense_fd=True, sequential=sequential,
                   want_p_rot=True, want_trace=True, want_gain=True)
    assert eng.kernel_name("kf").startswith("kf_dense_rows_kernel")
    xo = eng.unpack(r["x_out"]).cpu().numpy()
    assert np.abs(xo - ref["x"]).max() < STATE_TOL
    assert np.abs(eng.unpack(r["p_rot"]).cpu().numpy() - ref["p_rot"]).max() < 1e-5
    assert np.abs(r["P_trace"].cpu().numpy().T / ref["P_trace"] - 1).max() < 1e-3
    kg = r["K_gain"].cpu().numpy().T
    assert np.abs(kg - ref["K_gain"]).max() < 1e-3 * max(1.0, np.abs(ref["K_gain"]).max())
    assert np.abs(x.cpu().numpy().T - ref["x_final"]).max() < STATE_TOL
    Pf = P.cpu().numpy().T.reshape(B, 12, 12)
    assert np.abs(Pf - ref["P_final"]).max() < 1e-3 * np.abs(ref["P_final"]).max()
    assert int(eng.failed(r["status"]).sum()) == 0


def test_dense_rows_kernel_full_matrix_noise_and_feature_rows(eng):
    """Non-diagonal Q and R (batch update only) and the FEAT instantiation (two-kernel fused path with dense_fd)."""
    from oracle import c_oracle as orc
    from optistate_amd.synth import synth_numpy, Q_DEFAULT, R_DEFAULT
    B, T = 19, 25
    rng = np.random.default_rng(5)
    A = rng.normal(0, 1, (12, 12)); Q = Q_DEFAULT + 1e-4 * (A @ A.T)
    A = rng.normal(0, 1, (10, 10)); R = R_DEFAULT + 1e-3 * (A @ A.T)
    d = synth_numpy(B, T, seed=8)
    d["body_ref"] = np.zeros((B, T, 12), dtype=np.float32); d["body_ref"][..., 0:3] = d["imu"][..., 0:3]
    ref = orc.kf_run_batch(d["p"], d["f"], d["dp"], d["imu"], d["contact"], d["x0"], np.tile(Q, (B, 1, 1)), Q, R, body_ref=d["body_ref"], mode=1)
    s = soa(eng, d)
    br = eng.pack(torch.as_tensor(d["body_ref"]))
    eng.set_noise(Q, R)
    x = torch.as_tensor(d["x0"].T.copy()).cuda()
    P = torch.as_tensor(np.tile(Q.astype(np.float32).reshape(144, 1), (1, B))).cuda()
    r = eng.kf_run(s["p"], s["f"], s["dp"], s["imu"], s["contact"], x, P, body_ref=br, dense_fd=True, sequential=False)
    assert np.abs(eng.unpack(r["x_out"]).cpu().numpy() - ref["x"]).max() < STATE_TOL
    assert int(eng.failed(r["status"]).sum()) == 0


@pytest.mark.parametrize("dense", [False, True], ids=["predict(p,f)", "predict_mpc"])
def test_batch_update_stays_with_the_reference_over_long_ill_conditioned_runs(eng, dense):
    """Round 5, found by tools/fuzz_kf.py: K from a SYMMETRISED S (a Cholesky of its lower triangle) lets the antisymmetric part of P
    grow step by step under the reference's covariance update P -= K H P; with the fitted noise set (cond(S) ~ 1e6) and hostile
    inputs (flight phases, fast yaw) the float64 filter was lost after 70-110 steps.  The reference inverts S as it is
    (kalman_filter.py:169); so does update_batch_row now (LU of the full S).  160 steps, batch form, against the oracle."""
    from oracle import c_oracle as orc
    from optistate_amd.synth import synth_numpy, Q_FITTED, R_FITTED
    B, T = 600, 160
    d = synth_numpy(B, T, seed=1055, hostile=True)
    kw = {}
    if dense:
        d["body_ref"] = np.zeros((B, T, 12), dtype=np.float32); d["body_ref"][..., 0:3] = d["imu"][..., 0:3]
        kw = dict(body_ref=eng.pack(torch.as_tensor(d["body_ref"])), dense_fd=True)
    ref = orc.kf_run_batch(d["p"], d["f"], d["dp"], d["imu"], d["contact"], d["x0"], np.tile(Q_FITTED, (B, 1, 1)), Q_FITTED, R_FITTED,
                           body_ref=d.get("body_ref"), mode=1 if dense else 0)
    s = soa(eng, d)
    eng.set_noise(Q_FITTED, R_FITTED)
    x = torch.as_tensor(d["x0"].T.copy()).cuda()
    P = torch.as_tensor(np.tile(Q_FITTED.astype(np.float32).reshape(144, 1), (1, B))).cuda()
    r = eng.kf_run(s["p"], s["f"], s["dp"], s["imu"], s["contact"], x, P, sequential=False, symmetric=False, want_trace=True, **kw)
    assert eng.kernel_name("kf").startswith("kf_dense_rows_kernel<BATCH")
    assert int(eng.failed(r["status"]).sum()) == 0 and int((ref["status"] != 0).sum()) == 0
    xo = eng.unpack(r["x_out"]).cpu().numpy()
    assert np.abs(xo - ref["x"]).max() < STATE_TOL
    assert np.abs(r["P_trace"].cpu().numpy().T / ref["P_trace"] - 1).max() < 1e-3
    Pf = P.cpu().numpy().T.reshape(B, 12, 12)
    assert np.abs(Pf - np.swapaxes(Pf, 1, 2)).max() < 1e-5 * np.abs(Pf).max()           # P stays symmetric to rounding, as the reference's does


def test_dropin_class_stays_with_the_reference_on_an_ill_conditioned_run(eng):
    """The drop-in Kalman_Filter (one wavefront, float64) through 160 hostile steps with the fitted noise set -- the conditions under
    which a gain from a symmetrised S loses P's symmetry (see the test above): against the oracle, and P symmetric to rounding."""
    from optistate_amd import Kalman_Filter
    from oracle import c_oracle as orc
    from optistate_amd.synth import synth_numpy, Q_FITTED, R_FITTED
    B, T = 3, 160
    d = synth_numpy(B, T, seed=1055, hostile=True)
    ref = orc.kf_run_batch(d["p"], d["f"], d["dp"], d["imu"], d["contact"], d["x0"], np.tile(Q_FITTED, (B, 1, 1)), Q_FITTED, R_FITTED)
    for b in range(B):
        kf = Kalman_Filter()
        kf.x[:] = d["x0"][b].astype(np.float64).reshape(12, 1)
        kf.Q = Q_FITTED.copy(); kf.R = R_FITTED.copy(); kf.P = Q_FITTED.copy()
        for t in range(T):
            a = lambda key, n: d[key][b, t].astype(np.float64).reshape(n, 1)
            x = kf.step(a("p", 12), a("f", 12), a("dp", 12), a("imu", 6), d["contact"][b, t].reshape(4, 1))
            assert np.abs(x.ravel() - ref["x"][b, t]).max() < 1e-6, (b, t)
        assert np.abs(kf.P - kf.P.T).max() < 1e-12 * np.abs(kf.P).max()
        assert np.abs(kf.P - ref["P_final"][b]).max() < 1e-6 * np.abs(ref["P_final"][b]).max()
