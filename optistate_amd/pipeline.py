"""Host-side mirror of the reference's harness around the hot path, batched on the device.

reference                                                        here
data_conversion_Kalman_to_Training.py:193-254 (KF loop + rows)   kalman_feature_rows() [forces from a log],
                                                                 kalman_feature_rows_mpc() [forces from the MPC, as the script does]
data_conversion_Kalman_to_Training.py:31-104 (Q/R from residuals) fit_noise_covariances()
gru_train.py:59-62 (dataset min/max)                             fit_minmax()
gru_test.py:99-101 (normalise)                                   normalize()
gru_train.py:186-192 / gru_test.py:138-145 (windows + labels)    make_windows()
gru_test.py:174-213 (inference loop, error bands, de-normalise)  predict_windows()

Only data movement and dataset statistics run as torch ops; the filter and the GRU run in the HIP kernels.
"""
import numpy as np
import torch


def kalman_feature_rows(eng, traj, Q, R, x0, P0=None):
    """traj: dict of [B][T][F] arrays (p, f, dp, imu, accel, contact uint8).  Returns (rows [B][T][60] device tensor in
    the reference's column order [x_post | accel | f | p_world | dp | imu], x_hist [B][T][12], status [B])."""
    eng.set_noise(Q, R)
    s = {k: eng.pack(torch.as_tensor(np.asarray(traj[k], dtype=np.float32))) for k in ("p", "f", "dp", "imu")}
    c = eng.pack_contact(torch.as_tensor(np.asarray(traj["contact"], dtype=np.uint8)))
    B = s["p"].shape[2]
    x = torch.as_tensor(np.asarray(x0, dtype=np.float32).reshape(B, 12).T.copy()).to(eng.device)
    P0 = np.tile(np.asarray(Q, dtype=np.float32).reshape(1, 144), (B, 1)) if P0 is None else np.asarray(P0, dtype=np.float32).reshape(B, 144)
    P = torch.as_tensor(P0.T.copy()).to(eng.device)
    r = eng.kf_run(s["p"], s["f"], s["dp"], s["imu"], c, x, P, want_p_rot=True)
    x_hist = eng.unpack(r["x_out"])
    p_world = eng.unpack(r["p_rot"])                       # the rotated p the reference records
    dev = eng.device
    t = lambda k: torch.as_tensor(np.asarray(traj[k], dtype=np.float32)).to(dev)
    rows = torch.cat([x_hist, t("accel"), t("f"), p_world, t("dp"), t("imu")], dim=2)
    return rows, x_hist, r["status"]


def kalman_feature_rows_mpc(eng, traj, Q, R, x0, P0=None):
    """The reference's own loop (data_conversion_Kalman_to_Training.py:136-254): KF2.estimate_state_mpc per step, forces
    from the convex MPC.  traj: dict of [B][T][F] arrays p, dp, imu, accel, ref (12) and contact uint8 [B][T][4].
    Returns (rows [B][T][60] = [x_post | accel | KF2.f[:, 0] | p (world-rotated in place) | dp | imu], x_hist, forces,
    status [B])."""
    eng.set_noise(Q, R)
    s = {k: eng.pack(torch.as_tensor(np.asarray(traj[k], dtype=np.float32))) for k in ("p", "dp", "imu", "ref")}
    c = eng.pack_contact(torch.as_tensor(np.asarray(traj["contact"], dtype=np.uint8)))
    B = s["p"].shape[2]
    x = torch.as_tensor(np.asarray(x0, dtype=np.float32).reshape(B, 12).T.copy()).to(eng.device)
    P0 = np.tile(np.asarray(Q, dtype=np.float32).reshape(1, 144), (B, 1)) if P0 is None else np.asarray(P0, dtype=np.float32).reshape(B, 144)
    P = torch.as_tensor(P0.T.copy()).to(eng.device)
    r = eng.kf_mpc_run(s["p"], s["dp"], s["imu"], c, s["ref"], x, P, want_p_rot=True)
    x_hist, forces, p_world = eng.unpack(r["x_out"]), eng.unpack(r["f"]), eng.unpack(r["p_rot"])
    t = lambda k: torch.as_tensor(np.asarray(traj[k], dtype=np.float32)).to(eng.device)
    rows = torch.cat([x_hist, t("accel"), forces, p_world, t("dp"), t("imu")], dim=2)
    return rows, x_hist, forces, r["status"]


def fit_noise_covariances(eng, p_est, dp, imu, contact, mocap, alias_measurements=True):
    """Q and R from one trajectory's one-step residuals (data_conversion_Kalman_to_Training.py:31-104).

    Every step starts from the ground truth, so the T-1 steps are independent and run as ONE batch: x = mocap[i];
    predict_mpc(p[i], x_ref = mocap[i], contact[i]) -> x_model (:55-63); z from step i+1's p, dp, contact, imu (:65-70);
    Q = diag var(mocap[i+1] - x_model), R = diag var(mocap[i+1][sel] - z) (:80-104, population variance).
    p_est, dp, mocap [T][12], imu [T][>=6], contact [T][4].
    alias_measurements=True reproduces the script as written: `measurement_data.append(KF.z)` (:74) stores the SAME array
    every step, so all entries equal the last z; False uses each step's own z.
    (The script also rebuilds its lists per trajectory, so its Q, R come from the LAST trajectory only: pass that one.)"""
    dev = eng.device
    T = np.asarray(mocap).shape[0]
    n = T - 1
    col = lambda a, lo, hi, w: torch.as_tensor(np.ascontiguousarray(np.asarray(a, dtype=np.float32).reshape(T, -1)[lo:hi, :w].T)).to(dev)
    cpk = lambda lo, hi: torch.as_tensor(np.asarray(contact, dtype=np.uint8).reshape(T, 4)[lo:hi].copy()).to(dev).contiguous().view(torch.int32).reshape(-1)
    gt0, gt1 = col(mocap, 0, n, 12), col(mocap, 1, T, 12)
    p0 = col(p_est, 0, n, 12)
    f = eng.mpc_solve(gt0, gt0, p0, cpk(0, n))["f"]
    x = gt0.clone()
    P = torch.zeros((144, n), dtype=torch.float32, device=dev)
    eng.kf_predict(p0, f, x, P, body_ref=gt0)                       # x is now x_model
    z = eng.kf_odom(col(p_est, 1, T, 12), col(dp, 1, T, 12), cpk(1, T), col(imu, 1, T, 6))
    sel = torch.tensor([0, 1, 2, 5, 6, 7, 8, 9, 10, 11], device=dev)
    res_model = (gt1 - x).double()
    zz = z[:, -1:].expand(-1, n) if alias_measurements else z
    res_meas = (gt1[sel] - zz).double()
    Qd = res_model.var(dim=1, unbiased=False).cpu().numpy()
    Rd = res_meas.var(dim=1, unbiased=False).cpu().numpy()
    return np.diag(Qd), np.diag(Rd)


def fit_minmax(rows):
    """Dataset-wide per-column min and max (gru_train.py:59-62).  rows [..., F] -> (min [F], max [F])."""
    flat = rows.reshape(-1, rows.shape[-1])
    return flat.amin(dim=0), flat.amax(dim=0)


def normalize(rows, mn, mx):
    return (rows - mn) / (mx - mn)


def denormalize(v, mn, mx):
    return v * (mx - mn) + mn


def make_windows(rows_norm, labels_norm, seq_len):
    """rows_norm [N][F] (one trajectory, time-ordered), labels_norm [N][12] -> windows [N-seq+1][seq][F] with
    window i = rows i..i+seq-1 and label i = labels[i+seq-1] (gru_train.py:186-192)."""
    w = rows_norm.unfold(0, seq_len, 1).permute(0, 2, 1).contiguous()
    return w, labels_norm[seq_len - 1:]


def predict_rows(model, rows_norm, seq_len, min_v, max_v):
    """predict_windows for the row stream itself: rows_norm [N][F] (one trajectory, time-ordered, normalised) -> the de-normalised
    (pred, band_above, band_below), each [N - seq_len + 1][12]; entry i belongs to window rows i .. i + seq_len - 1, i.e. to the
    label of row i + seq_len - 1 (gru_test.py:138-140,174-213).  No window tensor is built (RNN.forward_windows)."""
    with torch.no_grad():
        out = model.forward_windows(rows_norm, seq_len)
    return model._engine.gru_bands(out, min_v, max_v)


def predict_windows(model, windows, min_v, max_v):
    """The inference loop of gru_test.py:174-213 for all windows at once: returns de-normalised (pred [N][12],
    band_above [N][12], band_below [N][12])."""
    with torch.no_grad():
        out = model(windows)
    return model._engine.gru_bands(out, min_v, max_v)
