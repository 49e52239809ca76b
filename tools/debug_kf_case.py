"""GPU (development): one fuzz_kf case under several kernel families: where do the trajectories part from the oracle?
    python tools/debug_kf_case.py B T seed noise hostile"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from optistate_amd import Engine  # noqa: E402
from optistate_amd.synth import synth_numpy, NOISE_SETS  # noqa: E402
from oracle import c_oracle as orc  # noqa: E402

B, T, seed = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
noise, hostile = sys.argv[4], sys.argv[5] == "1"
Q, R = NOISE_SETS[noise]
d = synth_numpy(B, T, seed=seed, hostile=hostile)
ref = orc.kf_run_batch(d["p"], d["f"], d["dp"], d["imu"], d["contact"], d["x0"], np.tile(Q, (B, 1, 1)), Q, R)
print("oracle: status nonzero", int((ref["status"] != 0).sum()), "| max |x|", float(np.abs(ref["x"]).max()), "| max P_trace", float(ref["P_trace"].max()),
      "| median P_trace at T", float(np.median(ref["P_trace"][:, -1])))
eng = Engine(0)
s = {k: eng.pack(torch.as_tensor(np.asarray(d[k], dtype=np.float32))) for k in ("p", "f", "dp", "imu")}
c = eng.pack_contact(torch.as_tensor(np.asarray(d["contact"])))
for name, kw in [("default", {}), ("seq-lanes", dict(sequential=True, symmetric=False, lane_per_trajectory=True)), ("batch", dict(sequential=False, symmetric=False))]:
    eng.set_noise(Q, R)
    x = torch.as_tensor(d["x0"].T.copy()).cuda()
    P = torch.as_tensor(np.tile(np.asarray(Q, dtype=np.float32).reshape(144, 1), (1, B))).cuda()
    r = eng.kf_run(s["p"], s["f"], s["dp"], s["imu"], c, x, P, **kw)
    xo = eng.unpack(r["x_out"]).cpu().numpy()
    st = r["status"].cpu().numpy()
    err = np.abs(xo - ref["x"]).max(axis=2)                 # [B][T]
    worst = err.max(axis=1)
    badb = np.argsort(-worst)[:5]
    print(f"{name} [{eng.kernel_name('kf')}]: status bits {np.unique(st)} | trajectories above 1e-4: {int((worst > 1e-4).sum())} | worst {worst.max():.2e}")
    for b in badb:
        t0 = int(np.argmax(err[b] > 1e-4)) if (err[b] > 1e-4).any() else -1
        print(f"    b={b}: worst {worst[b]:.2e}, first step above 1e-4: {t0}, status {st[b]}, oracle P_trace there {ref['P_trace'][b, max(t0, 0)]:.3e}, "
              f"stance legs at that step {int(d['contact'][b, max(t0, 0)].sum())}, |x| {np.abs(ref['x'][b, max(t0,0)]).max():.2e}")
