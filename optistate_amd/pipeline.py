"""Host-side mirror of the reference's harness around the hot path, batched on the device.

reference                                                        here
data_conversion_Kalman_to_Training.py:193-254 (KF loop + rows)   kalman_feature_rows()
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


def predict_windows(model, windows, min_v, max_v):
    """The inference loop of gru_test.py:174-213 for all windows at once: returns de-normalised (pred [N][12],
    band_above [N][12], band_below [N][12])."""
    with torch.no_grad():
        out = model(windows)
    pred, err = out[:, 0:12], out[:, 12:24]
    return (denormalize(pred, min_v, max_v), denormalize(pred + err, min_v, max_v), denormalize(pred - err, min_v, max_v))
