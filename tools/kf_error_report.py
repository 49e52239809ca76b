import sys; sys.path.insert(0, "/root/repo")
import numpy as np, torch
from optistate_amd import Engine
from optistate_amd.synth import synth_numpy, Q_DEFAULT, R_DEFAULT
from oracle import c_oracle as co
eng = Engine(0); eng.set_noise(Q_DEFAULT, R_DEFAULT)
B, T = 512, 300
d = synth_numpy(B, T, seed=5)
ref = co.kf_run_batch(d["p"], d["f"], d["dp"], d["imu"], d["contact"], d["x0"], np.tile(Q_DEFAULT, (B, 1, 1)), Q_DEFAULT, R_DEFAULT)
s = {k: eng.pack(torch.as_tensor(d[k])) for k in ("p", "f", "dp", "imu")}
c = eng.pack_contact(torch.as_tensor(d["contact"]))
for name, kw in (("sym lanes", dict(lane_per_trajectory=True)), ("rows", dict()), ("full seq", dict(symmetric=False, lane_per_trajectory=True)), ("batch", dict(sequential=False, symmetric=False, lane_per_trajectory=True))):
    x = torch.as_tensor(d["x0"].T.copy()).cuda(); P = torch.as_tensor(np.tile(Q_DEFAULT.astype(np.float32).reshape(144, 1), (1, B))).cuda()
    r = eng.kf_run(s["p"], s["f"], s["dp"], s["imu"], c, x, P, **kw)
    xo = eng.unpack(r["x_out"]).cpu().numpy()
    Pf = P.cpu().numpy().T.reshape(B, 12, 12)
    print(f"{name:10s} state linf {np.abs(xo - ref['x']).max():.2e}   P rel {np.abs(Pf - ref['P_final']).max() / np.abs(ref['P_final']).max():.2e}")
