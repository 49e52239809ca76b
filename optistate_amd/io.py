"""On-disk formats of the reference pipeline around the hot path, host side only.

* `saved_trajectories.pkl` (data_collection/data_conversion_raw_to_Kalman.py:443-447): {traj_num: {'p_list_est', 'p_list_ref',
  'dp_list', 'imu_list' (12 columns: Euler angles, rates, angular acc, linear acc), 'contact_list', 't265_list', 'mocap_list',
  'ref_list', 'time_list'}}, every entry a per-step list of (k,1) arrays.  `load_saved_trajectories` ->
  `trajectories_to_batch` packs them into the [B][T][F] arrays the engine's `pack()` turns into device streams; ragged
  lengths are padded by repeating the last step and reported in `lengths`.
* `rnn_data.pkl` (data_collection/data_conversion_Kalman_to_Training.py:333-336): {k: {'state_INPUT': rows of 60,
  'state_MOCAP': rows of 12, 'state_T265': rows of 12}} -- `save_rnn_data` / `load_rnn_data`.
* `scaling_params.pkl` (gru/gru_train.py:74-132): min/max vectors -- `save_scaling_params` / `load_scaling_params`.

* raw `.mat` logs (data_collection/data_conversion_raw_to_Kalman.py:43-57): `load_mat_trajectory` restates that script's ETL
  (mocap alignment, finite differences, list windows with their off-by-one) for everything that does not need the absent
  `scaler_kin` leg Jacobian; foot velocities come from a caller-supplied Jacobian or array.  Pinned by golden G13
  (tests/golden/etl_g13.npz, tools/gen_golden_etl.py): the script itself executed unmodified in the build container on
  seeded synthetic logs (cv2.imwrite recorded, scaler_kin's Jacobian a stand-in callable); tests/test_io_formats.py
  requires every list to agree to 1e-12, at reduced and at the shipped cutoffs.

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


DATA_CUTOFF_START, DATA_CUTOFF_END = 430, 4494        # settings.py:15-16


def load_mat_trajectory(path, cutoff=DATA_CUTOFF_START, end_cutoff=DATA_CUTOFF_END, leg_jacobian=None, dp=None,
                        want_depth=False):
    """One raw `.mat` log -> the per-trajectory dict the reference stores in saved_trajectories.pkl
    (data_collection/data_conversion_raw_to_Kalman.py:39-447), ready for `trajectories_to_batch({1: d})`.

    Fields read (:43-57): foot_state_history (N,4,3), footSteps_ref (N,4,3), bodyCM_ref (N,3), bodyR_ref (N,3),
    liftLeg_ref (N,4), body_state_history (N,12), time_history (1,N), imu (N,6: lin-acc, ang-acc), encoder_history (N,4,3),
    depth4 (N,H,W) in [0,1], mocap_history (N,7: mm + xyzw quaternion).  control_history is loaded by the script and never
    used (:50), so it is optional here.

    Rules reproduced from the script:
      * mocap rows cutoff-1 .. N-2 are re-expressed relative to row 0 (rotation initial^-1 * R_i; position through the
        inverse of the initial pose, z + 0.28) IN PLACE and in order (:126-156), so an all-zero row borrows the ALREADY
        transformed previous row, metres divided by 1000 again, exactly as the script does (:133-137);
      * T265 rates/velocities are finite differences written into row i+1 (:158-173);
      * mocap_list / ref_list / t265_list / time_list / imu_list cover i = cutoff-1 .. end_cutoff-2 (end_cutoff - cutoff
        entries; mocap_list entry = row i+1) (:176-266); p_list_*, dp_list, contact_list cover i = cutoff .. end_cutoff-2
        (one entry fewer) (:328-420) -- the consumer indexes both by the same k
        (data_conversion_Kalman_to_Training.py:170-199), so p[k] is one log row ahead of imu[k];
      * imu_list row = [theta(3), omega(3) (T265 finite difference), ang-acc = imu[:,3:6], lin-acc = imu[:,0:3]] (:269);
      * contact = 1 where liftLeg_ref == 0 (:409-418).
    Foot velocities (:388-404) need scaler_kin's leg Jacobian: pass `leg_jacobian(theta (3,), which_leg) -> (3,3)` (units as
    scaler_kin: mm, hence the /1000 of :399) or a precomputed `dp` (n,12); otherwise dp_list is zeros and
    d['dp_available'] is False.  want_depth: d['depth_u8'] (n,H,W) = the 8-bit frames the script writes as PNGs (:427-433).
    """
    import scipy.io
    from scipy.spatial.transform import Rotation as Rot
    data = scipy.io.loadmat(path)
    g = lambda k: np.array(data[k], dtype=np.float64)
    p_est, p_ref = g("foot_state_history"), g("footSteps_ref")
    cm_ref, r_ref, lift = g("bodyCM_ref"), g("bodyR_ref"), g("liftLeg_ref")
    t265, time_h, imu, mocap = g("body_state_history"), g("time_history"), g("imu"), g("mocap_history")
    N = mocap.shape[0]
    end_cutoff = min(int(end_cutoff), N - 1)
    if not (1 <= cutoff < end_cutoff):
        raise ValueError(f"cutoff {cutoff} / end_cutoff {end_cutoff} do not fit a log of {N} rows")
    tt = time_h.reshape(-1)

    # ---- mocap alignment + T265 finite differences, rows cutoff-1 .. N-2 (:121-173), sequential like the script ----
    q0 = mocap[0, 3:].copy()
    rot0 = Rot.from_quat(q0)
    T = np.eye(4)
    T[0:3, 0:3] = rot0.as_matrix()                      # initial_t265_quat is the identity (:112-113)
    T[0:3, 3] = mocap[0, 0:3] / 1000.0
    Tinv = np.linalg.inv(T)
    for i in range(cutoff - 1, N - 1):
        dt = tt[i + 1] - tt[i]
        pos, quat = mocap[i, 0:3] / 1000.0, mocap[i, 3:]
        if np.all(pos == 0) or np.all(quat == 0):
            pos, quat = mocap[i - 1, 0:3] / 1000.0, mocap[i - 1, 3:]          # the previous row as it stands NOW (:135-136)
            if np.all(quat == 0):
                raise ValueError(f"mocap rows {i - 1} and {i} are both invalid (the reference script fails here too)")
        mocap[i, 3:] = (rot0.inv() * Rot.from_quat(quat)).as_quat()
        ph = Tinv @ np.array([pos[0], pos[1], pos[2], 1.0])
        ph[2] += 0.28
        mocap[i, 0:3] = ph[0:3]
        t265[i + 1, 6:9] = (t265[i + 1, 0:3] - t265[i, 0:3]) / dt
        t265[i + 1, 9:12] = (t265[i + 1, 3:6] - t265[i, 3:6]) / dt

    def euler(q_xyzw):                                   # quaternion_2_euler (:72-90): extrinsic x-y-z
        return Rot.from_quat(q_xyzw).as_euler("xyz")

    out = {k: [] for k in ("p_list_est", "p_list_ref", "dp_list", "imu_list", "contact_list", "t265_list", "mocap_list",
                           "ref_list", "time_list")}
    col = lambda v, n: np.asarray(v, dtype=np.float64).reshape(n, 1)
    for i in range(cutoff - 1, end_cutoff - 1):
        dt = tt[i + 1] - tt[i]
        e0, e1 = euler(mocap[i, 3:]), euler(mocap[i + 1, 3:])
        vel = (mocap[i + 1, 0:3] - mocap[i, 0:3]) / dt
        dth = (e1 - e0) / dt
        out["mocap_list"].append(col(np.concatenate([e1, mocap[i + 1, 0:3], dth, vel]), 12))
        out["ref_list"].append(col([r_ref[i, 0], r_ref[i, 1], r_ref[i, 2], cm_ref[i, 0], cm_ref[i, 1], cm_ref[i, 2], 0.0, 0.0, 0.0,
                                    vel[0], vel[1], 0.0], 12))
        out["t265_list"].append(col(t265[i, 0:12], 12))
        out["time_list"].append(tt[i])
        out["imu_list"].append(col(np.concatenate([t265[i, 0:3], t265[i, 6:9], imu[i, 3:6], imu[i, 0:3]]), 12))
    n = end_cutoff - 1 - cutoff
    enc = g("encoder_history") if "encoder_history" in data else None
    if dp is not None:
        dp = np.asarray(dp, dtype=np.float64).reshape(n, 12)
    for k, i in enumerate(range(cutoff, end_cutoff - 1)):
        out["p_list_est"].append(col(p_est[i].reshape(12), 12))
        out["p_list_ref"].append(col(p_ref[i].reshape(12), 12))
        cur = np.zeros((12, 1))
        if dp is not None:
            cur[:, 0] = dp[k]
        elif leg_jacobian is not None:
            dt = tt[i + 1] - tt[i]
            for j in range(4):
                dtheta = (enc[i + 1, j, :] - enc[i, j, :]) / dt
                cur[3 * j:3 * j + 3, 0] = (np.asarray(leg_jacobian(enc[i, j, :], j), dtype=np.float64) @ dtheta) / 1000.0
        out["dp_list"].append(cur)
        out["contact_list"].append(np.array([1 if lift[i, j] == 0 else 0 for j in range(4)]).reshape(4, 1))
    out["dp_available"] = dp is not None or leg_jacobian is not None
    if want_depth:
        depth = np.asarray(data["depth4"])
        out["depth_u8"] = (depth[cutoff:end_cutoff - 1] * 255).astype(np.uint8)
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
