"""On-disk formats of the reference pipeline around the hot path, host side only.

* `saved_trajectories.pkl` (data_collection/data_conversion_raw_to_Kalman.py:443-447): {traj_num: {'p_list_est', 'p_list_ref',
  'dp_list', 'imu_list' (12 columns: Euler angles, rates, angular acc, linear acc), 'contact_list', 't265_list', 'mocap_list',
  'ref_list', 'time_list'}}, every entry a per-step list of (k,1) arrays.  `load_saved_trajectories` ->
  `trajectories_to_batch` packs them into the [B][T][F] arrays the engine's `pack()` turns into device streams; ragged
  lengths are padded by repeating the last step and reported in `lengths`.
* `rnn_data.pkl` (data_collection/data_conversion_Kalman_to_Training.py:333-336): {k: {'state_INPUT': rows of 60,
  'state_MOCAP': rows of 12, 'state_T265': rows of 12}} -- `save_rnn_data` / `load_rnn_data`.
* `scaling_params.pkl` (gru/gru_train.py:74-132): min/max vectors -- `save_scaling_params` / `load_scaling_params`.

The reference's recorded forces are never used by its pipeline (the MPC's forces are: SURVEY.md appendix 9); a trajectory
dict may therefore carry an extra 'f_list' (externally supplied ground-reaction forces); without it `f` is the quasi-static
share m*g/4 on every stance leg.
"""
import pickle

import numpy as np

MASS, G = 8.8, 9.81


def load_saved_trajectories(path):
    with open(path, "rb") as fh:
        return pickle.load(fh)


def _stack(lst, width):
    return np.stack([np.asarray(v, dtype=np.float64).reshape(-1)[:width] for v in lst]).astype(np.float32)


def trajectories_to_batch(data, keys=None):
    """data: the dict stored in saved_trajectories.pkl.  Returns dict(p, f, dp, imu, accel, contact, body_ref, x0, mocap,
    lengths) with arrays [B][T][F] (T = longest trajectory; shorter ones repeat their last step)."""
    keys = sorted(data.keys()) if keys is None else list(keys)
    per = []
    for k in keys:
        tr = data[k]
        n = min(len(tr["p_list_est"]), len(tr["dp_list"]), len(tr["imu_list"]), len(tr["contact_list"]))
        imu12 = _stack(tr["imu_list"][:n], 12)
        contact = np.stack([np.asarray(c).reshape(-1)[:4] for c in tr["contact_list"][:n]]).astype(np.uint8)
        d = dict(p=_stack(tr["p_list_est"][:n], 12), dp=_stack(tr["dp_list"][:n], 12), imu=imu12[:, 0:6],
                 accel=imu12[:, 6:12], contact=contact)
        if "f_list" in tr:
            d["f"] = _stack(tr["f_list"][:n], 12)
        else:
            f = np.zeros((n, 12), dtype=np.float32)
            stance = np.maximum(contact.sum(1, keepdims=True), 1)
            f[:, 2::3] = contact * (MASS * G) / stance
            d["f"] = f
        d["body_ref"] = _stack(tr["ref_list"][:n], 12) if "ref_list" in tr else np.zeros((n, 12), np.float32)
        d["mocap"] = _stack(tr["mocap_list"][:n], 12) if "mocap_list" in tr else np.zeros((n, 12), np.float32)
        per.append(d)
    lengths = np.array([d["p"].shape[0] for d in per], dtype=np.int64)
    T = int(lengths.max())
    out = {}
    for name in per[0]:
        arrs = []
        for d in per:
            a = d[name]
            if a.shape[0] < T:
                a = np.concatenate([a, np.repeat(a[-1:], T - a.shape[0], axis=0)], axis=0)
            arrs.append(a)
        out[name] = np.stack(arrs)
    out["x0"] = out["mocap"][:, 0, :].copy()          # KF2.x[:] = mocap_list[0]  (Kalman_to_Training.py:137-138)
    out["lengths"] = lengths
    return out


def save_rnn_data(path, rows, mocap, t265=None, lengths=None):
    """rows [B][T][60], mocap [B][T][12] (numpy) -> the reference's rnn_data.pkl layout (keys 1..B, python lists)."""
    rows, mocap = np.asarray(rows), np.asarray(mocap)
    B = rows.shape[0]
    t265 = np.zeros_like(mocap) if t265 is None else np.asarray(t265)
    out = {}
    for b in range(B):
        n = rows.shape[1] if lengths is None else int(lengths[b])
        out[b + 1] = {"state_INPUT": rows[b, :n].astype(np.float64).tolist(),
                      "state_MOCAP": mocap[b, :n].astype(np.float64).tolist(),
                      "state_T265": t265[b, :n].astype(np.float64).tolist()}
    with open(path, "wb") as fh:
        pickle.dump(out, fh)
    return out


def load_rnn_data(path, datasets=None):
    with open(path, "rb") as fh:
        d = pickle.load(fh)
    ks = sorted(d.keys()) if datasets is None else datasets
    kf = np.concatenate([np.asarray(d[k]["state_INPUT"], dtype=np.float64) for k in ks])
    mocap = np.concatenate([np.asarray(d[k]["state_MOCAP"], dtype=np.float64) for k in ks])
    return kf, mocap


def save_scaling_params(path, min_kf, max_kf, min_vic, max_vic):
    p = {"min_vals_KF": np.asarray(min_kf), "max_vals_KF": np.asarray(max_kf),
         "min_vals_VIC": np.asarray(min_vic), "max_vals_VIC": np.asarray(max_vic)}
    with open(path, "wb") as fh:
        pickle.dump(p, fh)
    return p


def load_scaling_params(path):
    with open(path, "rb") as fh:
        return pickle.load(fh)
