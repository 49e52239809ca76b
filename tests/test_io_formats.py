"""CPU: the reference's on-disk formats (synthetic pickles with the schema of data_conversion_raw_to_Kalman.py:443-447) and the raw
`.mat` ETL pinned to the reference script's own output (G13)."""
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


def _write_g13_log(path, seed, N, drop_rows):
    """The raw log the G13 fixture was generated from (tests/golden_recipes.g13_raw: arithmetic-only, rebuilt bit for bit)."""
    import scipy.io
    from golden_recipes import g13_raw
    raw = g13_raw(seed, N, drop_rows=drop_rows)
    scipy.io.savemat(path, raw)
    return raw


LISTS = ("p_list_est", "p_list_ref", "dp_list", "imu_list", "contact_list", "t265_list", "mocap_list", "ref_list")


def _g13():
    import os
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "etl_g13.npz"))


def test_mat_loader_reproduces_the_reference_etl_script_g13(tmp_path):
    """G13 (tools/gen_golden_etl.py): data_collection/data_conversion_raw_to_Kalman.py executed UNMODIFIED on these logs
    (cv2.imwrite recorded, scaler_kin's leg Jacobian = golden_recipes.g13_leg_jacobian, cutoffs 430 / 640).  Every list of
    saved_trajectories.pkl -- the off-by-one windows (:176,233 vs :328,388,412), the in-place mocap re-expression with its
    dropped-frame quirk (:133-137, trajectory 1 row 450), T265 finite differences (:158-173), the imu row order (:269),
    contact = (liftLeg == 0) (:409-418), dp through the Jacobian / 1000 (:388-404) and the 8-bit depth frames (:427-433)."""
    from golden_recipes import g13_leg_jacobian, G13_SEED, G13_ROWS, G13_CUTOFF, G13_END
    from optistate_amd.io import load_mat_trajectory, trajectories_to_batch
    g = _g13()
    for k, (seed, drops) in enumerate(((G13_SEED, (450,)), (G13_SEED + 1, ())), start=1):
        path = str(tmp_path / f"traj_{k}.mat")
        _write_g13_log(path, seed, G13_ROWS, drops)
        d = load_mat_trajectory(path, cutoff=G13_CUTOFF, end_cutoff=G13_END, leg_jacobian=g13_leg_jacobian, want_depth=True)
        for name in LISTS:
            want = g[f"k{k}_{name}"]
            got = np.stack([np.asarray(v, dtype=np.float64).reshape(-1) for v in d[name]])
            assert got.shape == want.shape, (k, name, got.shape, want.shape)
            assert np.abs(got - want).max() <= 1e-12 * max(1.0, np.abs(want).max()), (k, name, np.abs(got - want).max())
        assert np.array_equal(np.asarray(d["time_list"], dtype=np.float64), g[f"k{k}_time_list"])
        assert np.array_equal(d["depth_u8"], g[f"k{k}_depth_u8"])
        assert len(d["imu_list"]) == G13_END - G13_CUTOFF and len(d["p_list_est"]) == G13_END - G13_CUTOFF - 1
        assert d["dp_available"]
    # the dict feeds trajectories_to_batch like a pickle entry; n = the shorter lists; x0 = mocap_list[0] (Kalman_to_Training.py:137-138)
    b = trajectories_to_batch({1: d})
    n = G13_END - G13_CUTOFF - 1
    assert b["p"].shape == (1, n, 12) and b["imu"].shape == (1, n, 6)
    assert np.array_equal(b["imu"][0], g["k2_imu_list"][:n, 0:6].astype(np.float32))
    assert np.array_equal(b["accel"][0], g["k2_imu_list"][:n, 6:12].astype(np.float32))
    assert np.array_equal(b["x0"][0], g["k2_mocap_list"][0].astype(np.float32))
    assert np.array_equal(b["body_ref"][0], g["k2_ref_list"][:n].astype(np.float32))
    assert np.array_equal(b["contact"][0], g["k2_contact_list"].astype(np.uint8))


def test_mat_loader_at_the_shipped_cutoffs_g13(tmp_path):
    """The same script at settings.py:15-16's own 430 / 4494 on a 4,500-row log with two dropped frames: list lengths, column
    sums and sampled entries (first two, middle, last two)."""
    from golden_recipes import g13_leg_jacobian, G13_SEED
    from optistate_amd.io import load_mat_trajectory
    g = _g13()
    path = str(tmp_path / "full.mat")
    _write_g13_log(path, G13_SEED + 2, 4500, (450, 3000))
    d = load_mat_trajectory(path, leg_jacobian=g13_leg_jacobian)              # defaults = the shipped cutoffs
    for name in LISTS:
        got = np.stack([np.asarray(v, dtype=np.float64).reshape(-1) for v in d[name]])
        assert got.shape[0] == int(g[f"full_len_{name}"][0]), name
        assert np.abs(got[[0, 1, 2031, -2, -1]] - g[f"full_pick_{name}"]).max() <= 1e-12 * max(1.0, np.abs(g[f"full_pick_{name}"]).max()), name
        assert np.abs(got.sum(0) - g[f"full_sum_{name}"]).max() <= 1e-9 * max(1.0, np.abs(g[f"full_sum_{name}"]).max()), name


def test_mat_loader_without_a_jacobian_says_so(tmp_path):
    from optistate_amd.io import load_mat_trajectory
    _write_g13_log(str(tmp_path / "t.mat"), 1, 64, ())
    d = load_mat_trajectory(str(tmp_path / "t.mat"), cutoff=5, end_cutoff=30)
    assert d["dp_available"] is False and all(float(np.abs(v).sum()) == 0.0 for v in d["dp_list"])
    import pytest
    with pytest.raises(ValueError):
        load_mat_trajectory(str(tmp_path / "t.mat"), cutoff=40, end_cutoff=30)
