"""GPU: the harness mirror (feature rows, min-max, sliding windows, de-normalised predictions with error bands)
against golden G7 and the oracle."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def test_feature_rows_minmax_windows_match_reference_layout():
    from optistate_amd import Engine, RNN
    from optistate_amd import pipeline as pl
    from oracle import c_oracle as orc
    g3, g7 = load_golden("kf_g3_traj.npz"), load_golden("kf_g7_feature.npz")
    eng = Engine(0)
    traj = {k: g3[k][0:1] for k in ("p", "f", "dp", "imu", "accel", "contact")}
    rows, x_hist, status = pl.kalman_feature_rows(eng, traj, g3["Q0"], g3["R0"], g3["x0"][0:1])
    assert int(status.abs().sum()) == 0
    rows = rows[0]
    assert np.abs(rows.cpu().numpy() - g7["rows"]).max() < 1e-4              # column order + rotated-p rule
    mn, mx = pl.fit_minmax(rows)
    assert np.abs(mn.cpu().numpy() - g7["min_vals"]).max() < 1e-4
    assert np.abs(mx.cpu().numpy() - g7["max_vals"]).max() < 1e-4
    norm = pl.normalize(rows, mn, mx)
    # per-column bound: a wrong min / max column moves a normalised value by O(1), float32 arithmetic by
    # scale_j * (filter error ~1e-6 + float32 rounding of the raw value) -- nothing in between passes
    scale = 1.0 / (g7["max_vals"] - g7["min_vals"])
    maxabs = np.maximum(np.abs(g7["max_vals"]), np.abs(g7["min_vals"]))
    tol = scale * (1e-6 + 1.2e-7 * maxabs) + 1e-6
    err = np.abs(norm.cpu().numpy() - g7["normalized"]).max(axis=0)
    assert (err < tol).all(), (np.argmax(err / tol), float((err / tol).max()))
    # sliding windows of 10 + labels at i+9, batched GRU, de-normalised bands (gru_test.py)
    T = rows.shape[0]
    labels = torch.rand(T, 12, device=rows.device)
    win, lab = pl.make_windows(norm, labels, 10)
    assert win.shape == (T - 9, 10, 60) and lab.shape == (T - 9, 12)
    assert torch.equal(win[5, 3], norm[8]) and torch.equal(lab[5], labels[14])
    torch.manual_seed(4)
    m = RNN(60, 64, 1, 24, torch.device("cuda")).to("cuda").eval()
    min_v, max_v = torch.zeros(12, device="cuda") - 2.0, torch.zeros(12, device="cuda") + 3.0
    pred, above, below = pl.predict_windows(m, win, min_v, max_v)
    ref, _, _ = orc.gru_forward(win.cpu().numpy(), orc.flatten_state_dict(m.state_dict(), 1), 60, 64, 1, 24)
    assert np.abs(pred.cpu().numpy() - (ref[:, :12] * 5.0 - 2.0)).max() < 1e-4
    assert np.abs(above.cpu().numpy() - ((ref[:, :12] + ref[:, 12:]) * 5.0 - 2.0)).max() < 1e-4
    assert np.abs(below.cpu().numpy() - ((ref[:, :12] - ref[:, 12:]) * 5.0 - 2.0)).max() < 1e-4


def test_edge_shapes_single_trajectory_single_step():
    """B = 1 and T = 1 through every batched entry point."""
    from optistate_amd import Engine, RNN, flatten_state_dict
    from optistate_amd.synth import synth_numpy, Q_DEFAULT, R_DEFAULT
    from oracle import c_oracle as orc
    eng = Engine(0)
    eng.set_noise(Q_DEFAULT, R_DEFAULT)
    torch.manual_seed(1)
    m = RNN(60, 64, 1, 24, torch.device("cpu"))
    eng.load_gru(flatten_state_dict(m.state_dict(), 1), 60, 64, 1, 24)
    for B, T in ((1, 1), (1, 7), (3, 1)):
        d = synth_numpy(B, T, seed=B * 10 + T)
        ref = orc.kf_run_batch(d["p"], d["f"], d["dp"], d["imu"], d["contact"], d["x0"], np.tile(Q_DEFAULT, (B, 1, 1)),
                               Q_DEFAULT, R_DEFAULT)
        s = {k: eng.pack(torch.as_tensor(d[k])) for k in ("p", "f", "dp", "imu", "accel")}
        c = eng.pack_contact(torch.as_tensor(d["contact"]))
        mm = torch.stack([torch.full((60,), -30.0), torch.full((60,), 30.0)]).cuda()
        for two in (False, True):
            x = torch.as_tensor(d["x0"].T.copy()).cuda()
            P = torch.as_tensor(np.tile(Q_DEFAULT.astype(np.float32).reshape(144, 1), (1, B))).cuda()
            r = eng.fused_run(s["p"], s["f"], s["dp"], s["imu"], c, s["accel"], mm, x, P, two_kernel=two)
            torch.cuda.synchronize()
            assert np.abs(eng.unpack(r["x_out"]).cpu().numpy() - ref["x"]).max() < 1e-4
            rows = np.concatenate([ref["x"], d["accel"], d["f"], ref["p_rot"], d["dp"], d["imu"]], axis=2)
            ro, _, _ = orc.gru_forward((rows + 30.0) / 60.0, orc.flatten_state_dict(m.state_dict(), 1), 60, 64, 1, 24)
            assert np.abs(r["out"].cpu().numpy() - ro).max() < 1e-4


def test_argument_errors_are_reported_not_crashes():
    from optistate_amd import Engine
    eng = Engine(0)
    with pytest.raises(RuntimeError):
        eng.gru_forward_soa(torch.zeros(3, 60, 4, device="cuda"))          # no weights loaded
    import ctypes as C
    out = torch.zeros(4, 24, device="cuda")
    rc = eng.lib.os_gru_forward_soa(eng._h, 4, 3, C.c_void_p(out.data_ptr()), C.c_void_p(out.data_ptr()), None, None)
    assert rc == -5 and b"os_gru_load" in eng.lib.os_last_error(eng._h)     # the C-ABI reports, it does not crash
    eng.set_noise(np.eye(12) * 0.01, np.eye(10) * 0.01 + 0.001)             # non-diagonal R
    z = torch.zeros(2, 12, 8, device="cuda")
    with pytest.raises(RuntimeError):
        eng.kf_run(z, z, z, torch.zeros(2, 6, 8, device="cuda"), torch.zeros(2, 8, dtype=torch.int32, device="cuda"),
                   torch.zeros(12, 8, device="cuda"), torch.zeros(144, 8, device="cuda"), sequential=True)


def test_empty_and_oversize_batches_are_refused_cleanly():
    """B = 0 / T = 0 (the reference would simply loop zero times) and a batch beyond the 32-bit buffer-offset range must
    come back as argument errors from the C-ABI, without launching anything."""
    import ctypes as C
    from optistate_amd import Engine
    eng = Engine(0)
    buf = torch.zeros(4096, device="cuda")
    p = C.c_void_p(buf.data_ptr())
    for B, T in ((0, 5), (5, 0)):
        rc = eng.lib.os_kf_run(eng._h, B, T, p, p, p, p, p, None, p, p, p, None, None, None, p, 1, None)
        assert rc == -2 and b"positive" in eng.lib.os_last_error(eng._h)
    rc = eng.lib.os_kf_run(eng._h, 8_000_000, 1, p, p, p, p, p, None, p, p, p, None, None, None, p, 1 | 4, None)
    assert rc == -2 and b"too large" in eng.lib.os_last_error(eng._h)
    torch.cuda.synchronize()


# ---- G13: what the reference's second script writes (data_conversion_Kalman_to_Training.py executed unmodified by
# tools/gen_golden_etl.py; QP formulation = reference, solver = certified stand-in).  tests/test_oracle_etl_g13.py pins the
# oracles to the same file on the CPU.
def test_noise_fit_matches_the_scripts_q_r_pkl_g13():
    """pipeline.fit_noise_covariances against Q_R.pkl (:31-109): one batch of independent one-step predictions from the
    ground truth; float32 device arithmetic, float64 variance."""
    from optistate_amd import Engine
    from optistate_amd import pipeline as pl
    g = load_golden("etl_g13.npz")
    eng = Engine(0)
    k = 2                                                             # the script fits on the LAST trajectory only
    n = g[f"k{k}_p_list_est"].shape[0]                                # traj_length = len(p_list_ref): the shorter lists
    Q, R = pl.fit_noise_covariances(eng, g[f"k{k}_p_list_est"], g[f"k{k}_dp_list"], g[f"k{k}_imu_list"][:n], g[f"k{k}_contact_list"],
                                    g[f"k{k}_mocap_list"][:n], alias_measurements=True)
    assert np.count_nonzero(Q - np.diag(np.diag(Q))) == 0 and np.count_nonzero(R - np.diag(np.diag(R))) == 0
    eq, er = np.abs(np.diag(Q) / np.diag(g["fit_Q"]) - 1).max(), np.abs(np.diag(R) / np.diag(g["fit_R"]) - 1).max()
    assert eq < 2e-3 and er < 2e-3, (eq, er)


def test_mpc_feature_rows_match_rnn_data_pkl_g13():
    """pipeline.kalman_feature_rows_mpc (os_kf_mpc_run) against rnn_data.pkl's state_INPUT (:136-254), both trajectories as one
    batch of two, with the script's fitted Q and R[0:3] = 1e-4."""
    from optistate_amd import Engine
    from optistate_amd import pipeline as pl
    g = load_golden("etl_g13.npz")
    eng = Engine(0)
    n = g["k1_p_list_est"].shape[0]
    st = lambda name, lo, hi: np.stack([g[f"k{k}_{name}"][:n, lo:hi] for k in (1, 2)])
    traj = dict(p=st("p_list_est", 0, 12), dp=st("dp_list", 0, 12), imu=st("imu_list", 0, 6), accel=st("imu_list", 6, 12),
                ref=st("ref_list", 0, 12), contact=st("contact_list", 0, 4).astype(np.uint8))
    x0 = np.stack([g[f"k{k}_mocap_list"][0] for k in (1, 2)])         # KF2.x[:] = mocap_list[0] (:137-138)
    rows, x_hist, forces, status = pl.kalman_feature_rows_mpc(eng, traj, g["fit_Q"], g["run_R"], x0)
    assert int(status.abs().max()) == 0
    rows = rows.cpu().numpy().astype(np.float64)
    want = np.stack([g[f"k{k}_state_INPUT"] for k in (1, 2)])
    e = np.abs(rows - want)
    assert e[:, :, 0:12].max() < 1e-4, e[:, :, 0:12].max()                       # the state bar
    assert e[:, :, 30:42].max() < 1e-5                                          # p rotated in place by next_state
    assert e[:, :, 12:18].max() < 1e-6 * max(1.0, np.abs(want[:, :, 12:18]).max()) and e[:, :, 42:60].max() < 1e-6 * max(1.0, np.abs(want[:, :, 42:60]).max())
    # forces: a near dead-beat controller (R = 1e-6) fed back from a float32 state, saturating at 150 N
    assert e[:, :, 18:30].max() < 2e-2, e[:, :, 18:30].max()
    assert np.median(e[:, :, 18:30]) < 1e-4
    from optistate_amd.synth import Q_DEFAULT, R_DEFAULT
    eng.set_noise(Q_DEFAULT, R_DEFAULT)
