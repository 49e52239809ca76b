"""Development probe: fused KF + GRU(60,64,L>1) (the SEQOUT instantiation) against the oracle chain for short T."""
import sys
import numpy as np
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from optistate_amd import Engine, RNN, flatten_state_dict
from optistate_amd.synth import synth_numpy, Q_FITTED, R_FITTED
from oracle import c_oracle as orc
for B in (64, 256, 300):
  for T in (1, 2, 3, 12):
    L = 2
    d = synth_numpy(B, T, seed=21)
    ref = orc.kf_run_batch(d["p"], d["f"], d["dp"], d["imu"], d["contact"], d["x0"], np.tile(Q_FITTED, (B, 1, 1)), Q_FITTED, R_FITTED)
    rows = np.concatenate([ref["x"], d["accel"].astype(np.float64), d["f"].astype(np.float64), ref["p_rot"], d["dp"].astype(np.float64), d["imu"].astype(np.float64)], axis=2)
    mn, mx = rows.reshape(-1, 60).min(0), rows.reshape(-1, 60).max(0) + 1e-3
    norm = (rows - mn) / (mx - mn)
    torch.manual_seed(5)
    m = RNN(60, 64, L, 24, torch.device("cpu"))
    w = orc.flatten_state_dict(m.state_dict(), L)
    ref_out, _, seq = orc.gru_forward(norm, w, 60, 64, L, 24)
    eng = Engine(0); eng.set_noise(Q_FITTED, R_FITTED)
    eng.load_gru(flatten_state_dict(m.state_dict(), L), 60, 64, L, 24)
    s = {k: eng.pack(torch.as_tensor(d[k])) for k in ("p", "f", "dp", "imu", "accel")}
    c = eng.pack_contact(torch.as_tensor(d["contact"]))
    x = torch.as_tensor(d["x0"].T.copy()).cuda()
    P = torch.as_tensor(np.tile(Q_FITTED.astype(np.float32).reshape(144, 1), (1, B))).cuda()
    mm = torch.as_tensor(np.stack([mn, mx]).astype(np.float32)).cuda()
    r = eng.fused_run(s["p"], s["f"], s["dp"], s["imu"], c, s["accel"], mm, x, P, two_kernel=False)
    e = np.abs(r["out"].cpu().numpy() - ref_out)
    print(f"B={B} T={T}: out err max {e.max():.3e}; per-trajectory err (first 8) {e.max(axis=1)[:8]}; worst trajectory {e.max(axis=1).argmax()}; kernels {eng.kernel_name('fused')} {eng.kernel_name('gru_layer')}")
