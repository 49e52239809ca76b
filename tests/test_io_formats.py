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
