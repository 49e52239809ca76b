"""Pins the CPU oracle (oracle/*.c) to golden vectors produced by the unmodified reference
(tools/gen_golden.py).  CPU only."""
import numpy as np
import pytest

from oracle import c_oracle as orc
from conftest import load_golden

TOL = 1e-12


def test_g1_rotation_odom_measurement():
    g = load_golden("kf_g1_odom.npz")
    for i in range(g["th"].shape[0]):
        R = orc.rotation(*g["th"][i])
        assert np.abs(R - g["R"][i]).max() <= TOL
        od = orc.get_odom(g["p"][i], g["dp"][i], g["contact"][i], g["imu"][i])
        assert np.abs(od - g["odom"][i]).max() <= TOL
        z = np.concatenate([g["imu"][i][0:3], od[0:1], g["imu"][i][3:6], od[1:4]])
        assert np.abs(z - g["z"][i]).max() <= TOL


def test_g2_next_state_and_truncation_quirk():
    g = load_golden("kf_g2_next_state.npz")
    n = g["x"].shape[0]
    n_trunc_nonzero = 0
    for i in range(n):
        xn, prot = orc.next_state(g["x"][i], g["p"][i], g["f"][i], float(g["dt"]))
        assert np.abs(xn - g["x_next"][i]).max() <= 1e-11, i
        assert np.abs(prot - g["p_rot"][i]).max() <= TOL, i
        n_trunc_nonzero += int(np.any(g["A_block"][i] != 0))
    # the fixture must exercise both branches of the int64 truncation
    assert 0 < n_trunc_nonzero < n
    # theta == 0 exactly: A block is the identity, so theta integrates omega
    assert np.array_equal(g["A_block"][0], np.eye(3))


@pytest.mark.parametrize("s", [0, 1])
def test_g3_trajectories(s):
    g = load_golden("kf_g3_traj.npz")
    Q, R = g[f"Q{s}"], g[f"R{s}"]
    r = orc.kf_run_batch(g["p"], g["f"], g["dp"], g["imu"], g["contact"], g["x0"],
                         np.tile(Q, (2, 1, 1)), Q, R)
    for b in range(2):
        assert np.abs(r["x"][b] - g[f"s{s}_b{b}_x"]).max() <= 1e-11
        assert np.abs(r["x_prior"][b] - g[f"s{s}_b{b}_x_prior"]).max() <= 1e-11
        assert np.abs(r["p_rot"][b] - g[f"s{s}_b{b}_p_rot"]).max() <= TOL
        assert np.abs(r["P_trace"][b] / g[f"s{s}_b{b}_P_trace"] - 1).max() <= 1e-10
        assert np.abs(r["K_gain"][b] - g[f"s{s}_b{b}_K_gain"]).max() <= 1e-10
        Pf = g[f"s{s}_b{b}_P_final"]
        assert np.abs(r["P_final"][b] - Pf).max() <= 1e-10 * np.abs(Pf).max()
        assert r["status"][b] == 0


def test_g3_gain_matrix_first_step():
    g = load_golden("kf_g3_traj.npz")
    for s in (0, 1):
        Q, R = g[f"Q{s}"], g[f"R{s}"]
        r = orc.kf_run_batch(g["p"][:, :1], g["f"][:, :1], g["dp"][:, :1], g["imu"][:, :1], g["contact"][:, :1],
                             g["x0"], np.tile(Q, (2, 1, 1)), Q, R)
        # re-do the update of step 0 to read K: prior state and covariance after predict
        import ctypes as C
        x = r["x_prior"][0, 0].copy()
        # P after predict = Fd Q Fd^T + Q; rebuild through the oracle's predict
        P = Q.copy().astype(np.float64)
        xx = g["x0"][0].astype(np.float64).copy(); pp = g["p"][0, 0].astype(np.float64).copy()
        ff = g["f"][0, 0].astype(np.float64).copy()
        orc.lib().ok_predict(orc._d(xx), orc._d(P), orc._d(pp), orc._d(ff), orc._d(np.ascontiguousarray(Q, dtype=np.float64)),
                             C.c_double(orc.DT), C.c_double(orc.MASS), orc._d(orc.INERTIA), C.c_double(orc.GZ))
        assert np.abs(xx - x).max() <= 1e-13
        z = g[f"s{s}_b0_z"][0]
        _, _, K, kg, st = orc.update(xx, P, z, R)
        assert st == 0
        assert np.abs(K - g[f"s{s}_b0_K0"]).max() <= 1e-10
        assert abs(kg - g[f"s{s}_b0_K_gain"][0]) <= 1e-10


@pytest.mark.parametrize("s", [0, 1])
def test_g4_batch(s):
    g = load_golden("kf_g4_batch.npz")
    Q, R = g[f"Q{s}"], g[f"R{s}"]
    B = g["p"].shape[0]
    r = orc.kf_run_batch(g["p"], g["f"], g["dp"], g["imu"], g["contact"], g["x0"], np.tile(Q, (B, 1, 1)), Q, R)
    assert np.abs(r["x"] - g[f"s{s}_x"]).max() <= 1e-10
    assert np.abs(r["p_rot"] - g[f"s{s}_p_rot"]).max() <= TOL
    assert np.abs(r["P_trace"] / g[f"s{s}_P_trace"] - 1).max() <= 1e-9
    assert np.abs(r["K_gain"] - g[f"s{s}_K_gain"]).max() <= 1e-9


def test_g7_feature_row_order_and_normalisation():
    g3 = load_golden("kf_g3_traj.npz")
    g7 = load_golden("kf_g7_feature.npz")
    T = g7["rows"].shape[0]
    for t in (0, 1, 57, T - 1):
        row = orc.feature_row(g3["s0_b0_x"][t], g3["accel"][0, t], g3["f"][0, t], g3["s0_b0_p_rot"][t],
                              g3["dp"][0, t], g3["imu"][0, t])
        assert np.abs(row - g7["rows"][t]).max() <= TOL
        nrow = orc.feature_row(g3["s0_b0_x"][t], g3["accel"][0, t], g3["f"][0, t], g3["s0_b0_p_rot"][t],
                               g3["dp"][0, t], g3["imu"][0, t], g7["min_vals"], g7["max_vals"])
        assert np.abs(nrow - g7["normalized"][t]).max() <= 1e-12
    # the feature row records the WORLD-rotated p, not the body-frame input
    assert np.abs(g7["rows"][5][30:42] - g3["p"][0, 5]).max() > 1e-4


def test_g8_estimate_state_mpc_with_supplied_forces():
    g = load_golden("kf_g8_mpc.npz")
    Q, R = g["Q"], g["R"]
    r = orc.kf_run_batch(g["p"], g["f"], g["dp"], g["imu"], g["contact"], g["x0"], np.tile(Q, (2, 1, 1)), Q, R,
                         body_ref=g["body_ref"], mode=1)
    for b in range(2):
        assert np.abs(r["x"][b] - g[f"b{b}_x"]).max() <= 1e-10
        # dense ~all-ones F_d makes P = F_d P F_d^T a sum over all 144 entries with heavy cancellation in
        # the update: summation-order noise is amplified (~1e-9 here), hence the looser bound on P
        assert np.abs(r["P_trace"][b] / g[f"b{b}_P_trace"] - 1).max() <= 1e-7
        # element-wise exp makes F_d dense ~ones (kalman_filter.py:157)
        assert g[f"b{b}_Fd_minmax"][:, 0].min() > 0.98


@pytest.mark.parametrize("name", ["small", "ref"])
def test_g5_gru_forward(name):
    g = load_golden(f"gru_g5_{name}.npz")
    I, H, L, Cc = [int(v) for v in g["dims"]]
    parts = []
    for l in range(L):
        for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
            parts.append(g[f"w:gru.{k}_l{l}"].astype(np.float64).ravel())
    parts += [g["w:fc.weight"].astype(np.float64).ravel(), g["w:fc.bias"].astype(np.float64).ravel()]
    w = np.concatenate(parts)
    out, hl, seq = orc.gru_forward(g["x"], w, I, H, L, Cc, want_seq=True)
    assert np.abs(out - g["out"]).max() <= 2e-6
    assert np.abs(hl - g["h_last"]).max() <= 2e-6
    assert np.abs(seq - g["seq_top"]).max() <= 2e-6


def test_g6_training_target_and_loss():
    g = load_golden("gru_g6_train.npz")
    loss, tgt = orc.gru_train_loss(g["outputs"], g["labels"])
    assert np.abs(tgt - g["target"]).max() <= 1e-7
    assert abs(loss - float(g["loss"])) <= 1e-7
