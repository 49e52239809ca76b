"""CPU: the reference's on-disk formats (synthetic pickles with the schema of data_conversion_raw_to_Kalman.py:443-447)."""
import pickle

import numpy as np

from optistate_amd import io as osio
from optistate_amd.synth import synth_numpy


def _fake_saved_trajectories(tmp_path, lengths=(37, 52)):
    data = {}
    for i, n in enumerate(lengths):
        d = synth_numpy(1, n, seed=40 + i)
        col = lambda a, w: [np.asarray(a[0, t], dtype=np.float64).reshape(w, 1) for t in range(n)]
        imu12 = np.concatenate([d["imu"], d["accel"]], axis=2)
        data[i + 1] = {"p_list_est": col(d["p"], 12), "p_list_ref": col(d["p"], 12), "dp_list": col(d["dp"], 12),
                       "imu_list": col(imu12, 12), "contact_list": [d["contact"][0, t].reshape(4, 1) for t in range(n)],
                       "t265_list": col(d["p"], 12), "mocap_list": col(np.tile(d["x0"][:, None, :], (1, n, 1)), 12),
                       "ref_list": col(d["p"], 12), "time_list": list(np.arange(n) * 0.01)}
    p = tmp_path / "saved_trajectories.pkl"
    with open(p, "wb") as fh:
        pickle.dump(data, fh)
    return p, data


def test_saved_trajectories_to_batch(tmp_path):
    p, data = _fake_saved_trajectories(tmp_path)
    b = osio.trajectories_to_batch(osio.load_saved_trajectories(p))
    assert b["p"].shape == (2, 52, 12) and b["imu"].shape == (2, 52, 6) and b["accel"].shape == (2, 52, 6)
    assert b["contact"].dtype == np.uint8 and b["contact"].shape == (2, 52, 4)
    assert list(b["lengths"]) == [37, 52]
    assert np.allclose(b["p"][0, 10], np.asarray(data[1]["p_list_est"][10]).ravel())
    assert np.allclose(b["accel"][1, 5], np.asarray(data[2]["imu_list"][5]).ravel()[6:12])
    assert np.array_equal(b["p"][0, 36], b["p"][0, 51])                     # ragged: last step repeated
    assert np.allclose(b["x0"][0], np.asarray(data[1]["mocap_list"][0]).ravel())
    # default forces: m*g shared by the stance legs, on z
    t = 3
    c = b["contact"][0, t]
    assert np.allclose(b["f"][0, t, 2::3], c * 8.8 * 9.81 / max(c.sum(), 1))


def test_rnn_data_and_scaling_roundtrip(tmp_path):
    rng = np.random.default_rng(0)
    rows, mocap = rng.normal(size=(2, 9, 60)), rng.normal(size=(2, 9, 12))
    osio.save_rnn_data(tmp_path / "rnn_data.pkl", rows, mocap, lengths=[9, 7])
    with open(tmp_path / "rnn_data.pkl", "rb") as fh:
        raw = pickle.load(fh)
    assert sorted(raw.keys()) == [1, 2] and set(raw[1].keys()) == {"state_INPUT", "state_MOCAP", "state_T265"}
    assert len(raw[2]["state_INPUT"]) == 7 and len(raw[1]["state_INPUT"][0]) == 60
    kf, mc = osio.load_rnn_data(tmp_path / "rnn_data.pkl")
    assert kf.shape == (16, 60) and np.allclose(kf[:9], rows[0]) and np.allclose(mc[9:], mocap[1, :7])
    mn, mx = kf.min(0), kf.max(0)
    osio.save_scaling_params(tmp_path / "scaling_params.pkl", mn, mx, mc.min(0), mc.max(0))
    sp = osio.load_scaling_params(tmp_path / "scaling_params.pkl")
    assert set(sp) == {"min_vals_KF", "max_vals_KF", "min_vals_VIC", "max_vals_VIC"} and np.allclose(sp["max_vals_KF"], mx)


def _synthetic_mat(path, N=64, seed=0):
    """A raw log with the schema of data_collection/data_conversion_raw_to_Kalman.py:43-57 (shapes (N,4,3), (N,3), (1,N), ...)."""
    import scipy.io
    from scipy.spatial.transform import Rotation as Rot
    rng = np.random.default_rng(seed)
    t = np.cumsum(rng.uniform(0.009, 0.011, N)).reshape(1, N)
    quat = Rot.from_euler("xyz", rng.normal(0, 0.2, (N, 3))).as_quat()
    mocap = np.concatenate([rng.normal(0, 300, (N, 3)) + np.array([1000.0, -500.0, 280.0]), quat], axis=1)
    mocap[20] = 0.0                                                      # a dropped mocap frame (:133-137)
    d = dict(foot_state_history=rng.normal(0, 0.1, (N, 4, 3)), footSteps_ref=rng.normal(0, 0.1, (N, 4, 3)),
             bodyCM_ref=rng.normal(0, 0.1, (N, 3)), bodyR_ref=rng.normal(0, 0.1, (N, 3)), control_history=rng.normal(0, 1, (N, 12)),
             liftLeg_ref=(rng.random((N, 4)) < 0.5).astype(np.float64), body_state_history=rng.normal(0, 0.1, (N, 12)),
             time_history=t, imu=rng.normal(0, 1, (N, 6)), encoder_history=rng.normal(0, 0.5, (N, 4, 3)),
             depth4=rng.random((N, 8, 8)), mocap_history=mocap)
    scipy.io.savemat(path, d)
    return d


def test_mat_loader_reproduces_the_scripts_indexing_rules(tmp_path):
    from scipy.spatial.transform import Rotation as Rot
    from optistate_amd.io import load_mat_trajectory, trajectories_to_batch
    raw = _synthetic_mat(str(tmp_path / "traj.mat"))
    cutoff, end = 5, 50
    jac = lambda theta, leg: np.eye(3) * (100.0 + leg)                   # stand-in for scaler_kin's leg Jacobian (mm)
    d = load_mat_trajectory(str(tmp_path / "traj.mat"), cutoff=cutoff, end_cutoff=end, leg_jacobian=jac, want_depth=True)
    # list lengths: imu/mocap/t265/ref/time cover range(cutoff-1, end-1); p/dp/contact range(cutoff, end-1) (:176,233 vs :328,388,412)
    for k in ("imu_list", "mocap_list", "t265_list", "ref_list", "time_list"):
        assert len(d[k]) == end - cutoff, k
    for k in ("p_list_est", "p_list_ref", "dp_list", "contact_list"):
        assert len(d[k]) == end - cutoff - 1, k
    t = raw["time_history"].reshape(-1)
    t265 = raw["body_state_history"].copy()
    for i in range(cutoff - 1, 63):                                      # finite differences written into row i+1 (:158-173)
        dt = t[i + 1] - t[i]
        t265[i + 1, 6:9] = (t265[i + 1, 0:3] - t265[i, 0:3]) / dt
        t265[i + 1, 9:12] = (t265[i + 1, 3:6] - t265[i, 3:6]) / dt
    for k in (0, 7, end - cutoff - 1):
        i = cutoff - 1 + k
        want = np.concatenate([t265[i, 0:3], t265[i, 6:9], raw["imu"][i, 3:6], raw["imu"][i, 0:3]])        # :269
        assert np.allclose(d["imu_list"][k].ravel(), want, atol=1e-12)
        assert d["time_list"][k] == t[i]
        assert np.allclose(d["t265_list"][k].ravel(), t265[i], atol=1e-12)
        assert np.allclose(d["ref_list"][k].ravel()[:6], np.concatenate([raw["bodyR_ref"][i], raw["bodyCM_ref"][i]]))
        assert d["ref_list"][k][6:9].sum() == 0.0 and d["ref_list"][k][11, 0] == 0.0
    for k in (0, 11, end - cutoff - 2):
        i = cutoff + k                                                   # p is one log row ahead of imu at the same list index
        assert np.array_equal(d["p_list_est"][k].ravel(), raw["foot_state_history"][i].reshape(12))
        assert np.array_equal(d["p_list_ref"][k].ravel(), raw["footSteps_ref"][i].reshape(12))
        assert np.array_equal(d["contact_list"][k].ravel(), (raw["liftLeg_ref"][i] == 0).astype(int))
        dth = (raw["encoder_history"][i + 1] - raw["encoder_history"][i]) / (t[i + 1] - t[i])
        want = np.concatenate([(100.0 + j) * dth[j] / 1000.0 for j in range(4)])                              # :388-399
        assert np.allclose(d["dp_list"][k].ravel(), want, atol=1e-12)
    assert d["dp_available"] and d["depth_u8"].shape == (end - cutoff - 1, 8, 8) and d["depth_u8"].dtype == np.uint8
    assert np.array_equal(d["depth_u8"][3], (raw["depth4"][cutoff + 3] * 255).astype(np.uint8))
    # mocap alignment: row 0's pose is the origin (z + 0.28), later rows relative to it (:101-156)
    m = raw["mocap_history"]
    rot0 = Rot.from_quat(m[0, 3:])
    i = cutoff - 1 + 3 + 1                                               # mocap_list[3] describes row i (= loop index + 1)
    pos = rot0.inv().apply(m[i, 0:3] / 1000.0 - m[0, 0:3] / 1000.0) + np.array([0, 0, 0.28])
    eul = (rot0.inv() * Rot.from_quat(m[i, 3:])).as_euler("xyz")
    assert np.allclose(d["mocap_list"][3].ravel()[3:6], pos, atol=1e-9)
    assert np.allclose(d["mocap_list"][3].ravel()[0:3], eul, atol=1e-9)
    # the dropped frame (row 20) borrowed row 19 AS ALREADY TRANSFORMED, divided by 1000 again (:135): reproduce that
    k20 = 20 - cutoff                                                    # mocap_list[k] = row cutoff + k
    pos19 = rot0.inv().apply(m[19, 0:3] / 1000.0 - m[0, 0:3] / 1000.0) + np.array([0, 0, 0.28])
    want20 = rot0.inv().apply(pos19 / 1000.0 - m[0, 0:3] / 1000.0) + np.array([0, 0, 0.28])
    assert np.allclose(d["mocap_list"][k20].ravel()[3:6], want20, atol=1e-9)
    # the dict feeds trajectories_to_batch like a pickle entry; n = the shorter lists
    b = trajectories_to_batch({1: d})
    assert b["p"].shape == (1, end - cutoff - 1, 12) and b["imu"].shape == (1, end - cutoff - 1, 6)
    assert np.allclose(b["imu"][0, 0], d["imu_list"][0].ravel()[:6].astype(np.float32))
    assert np.allclose(b["accel"][0, 0], d["imu_list"][0].ravel()[6:].astype(np.float32))
    assert np.allclose(b["x0"][0], d["mocap_list"][0].ravel().astype(np.float32))


def test_mat_loader_without_a_jacobian_says_so(tmp_path):
    from optistate_amd.io import load_mat_trajectory
    _synthetic_mat(str(tmp_path / "t.mat"), seed=1)
    d = load_mat_trajectory(str(tmp_path / "t.mat"), cutoff=5, end_cutoff=30)
    assert d["dp_available"] is False and all(float(np.abs(v).sum()) == 0.0 for v in d["dp_list"])
    import pytest
    with pytest.raises(ValueError):
        load_mat_trajectory(str(tmp_path / "t.mat"), cutoff=40, end_cutoff=30)
