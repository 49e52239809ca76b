"""G13 pins the oracles (and, through them, the GPU tests' reference) to what the reference's second script writes:
data_collection/data_conversion_Kalman_to_Training.py executed unmodified (tools/gen_golden_etl.py) -- the Q/R fit of
:31-109 (Q_R.pkl) and the estimate_state_mpc loop with its 60-column rows of :115-336 (rnn_data.pkl, p_trace).
QP: formulation = reference, solver = certified stand-in (see tests/test_oracle_mpc_g12.py)."""
import os

import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import mpc_oracle as mo

SEL = [0, 1, 2, 5, 6, 7, 8, 9, 10, 11]


@pytest.fixture(scope="module")
def g():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "etl_g13.npz"))


def test_noise_fit_matches_the_scripts_q_r_pkl(g):
    """:47-104 on the LAST trajectory (the script rebuilds its lists per trajectory and fits after the loop): every step starts
    from the mocap row, predict_mpc with x_ref = the same row, z from step i+1; R from T copies of the LAST z because
    `measurement_data.append(KF.z)` (:74) stores one aliased array.  Population variance."""
    k = 2
    mocap, p, dp, imu, contact = (g[f"k{k}_{n}"] for n in ("mocap_list", "p_list_est", "dp_list", "imu_list", "contact_list"))
    n = p.shape[0]                                                   # traj_length = len(p_list_ref) (:45)
    xm, z_last = [], None
    for i in range(n - 1):
        x = mocap[i]
        f, _, info = mo.mpc_forces(x, x, p[i], contact[i])
        xn, _ = co.next_state(x, p[i].copy(), f)
        xm.append(xn)
        od = co.get_odom(p[i + 1], dp[i + 1], contact[i + 1].astype(np.uint8), imu[i + 1][0:6])
        z_last = np.concatenate([imu[i + 1][0:3], [od[0]], imu[i + 1][3:6], od[1:4]])
    gt1 = mocap[1:n]
    Q = np.var(gt1 - np.array(xm), axis=0)
    R = np.var(gt1[:, SEL] - z_last[None, :], axis=0)
    assert np.count_nonzero(g["fit_Q"] - np.diag(np.diag(g["fit_Q"]))) == 0
    assert np.abs(Q / np.diag(g["fit_Q"]) - 1).max() < 1e-7, np.abs(Q / np.diag(g["fit_Q"]) - 1).max()
    assert np.abs(R / np.diag(g["fit_R"]) - 1).max() < 1e-9
    # the run below uses the fitted R with its first three entries overwritten (:142-144)
    want = np.diag(g["fit_R"]).copy(); want[0:3] = 1e-4
    assert np.array_equal(np.diag(g["run_R"]), want)


@pytest.mark.parametrize("k,T", [(1, 45), (2, 209)])
def test_filter_loop_rows_match_rnn_data_pkl(g, k, T):
    """:136-254: KF2.x = mocap_list[0], P = Q, estimate_state_mpc per step, row = [x | imu_list[i][6:12] | KF2.f[:, 0] |
    p (rotated in place by next_state) | dp | imu[0:6]].  Trajectory 1 crosses its dropped mocap frame only in the labels;
    trajectory 2 is checked over its whole length."""
    Q, R = g["fit_Q"], g["run_R"]
    p, dp, imu, contact, ref, mocap = (g[f"k{k}_{n}"] for n in ("p_list_est", "dp_list", "imu_list", "contact_list", "ref_list", "mocap_list"))
    rows, ptr = g[f"k{k}_state_INPUT"], g[f"k{k}_p_trace"]
    assert rows.shape == (p.shape[0], 60)
    x = mocap[0].copy(); P = Q.copy()
    for t in range(T):
        f, _, _ = mo.mpc_forces(x, ref[t], p[t], contact[t])
        r = co.kf_run_batch(p[t].reshape(1, 1, 12), f.reshape(1, 1, 12), dp[t].reshape(1, 1, 12), imu[t, 0:6].reshape(1, 1, 6),
                            contact[t].astype(np.uint8).reshape(1, 1, 4), x.reshape(1, 12), P.reshape(1, 144), Q, R,
                            body_ref=ref[t].reshape(1, 1, 12), mode=1)
        x = r["x_final"][0].copy(); P = r["P_final"][0].copy()
        row = np.concatenate([x, imu[t, 6:12], f, r["p_rot"][0, 0], dp[t], imu[t, 0:6]])
        assert np.abs(row[0:12] - rows[t, 0:12]).max() < 1e-8, (t, np.abs(row[0:12] - rows[t, 0:12]).max())
        assert np.abs(row[18:30] - rows[t, 18:30]).max() < 1e-5, t
        assert np.abs(row[30:42] - rows[t, 30:42]).max() < 1e-9, t
        assert np.array_equal(row[12:18], rows[t, 12:18]) and np.array_equal(row[42:60], rows[t, 42:60])
        assert abs(r["P_trace"][0, 0] / ptr[t] - 1) < 1e-6
    assert np.array_equal(g[f"k{k}_state_MOCAP"], mocap[:rows.shape[0]])
    assert np.array_equal(g[f"k{k}_state_T265"], g[f"k{k}_t265_list"][:rows.shape[0]])
