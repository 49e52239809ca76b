#!/usr/bin/env python3
"""Development aid: the rows kernel (16 lanes per trajectory) against the lane-per-trajectory kernel on a small batch; prints
the per-component difference of the first steps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from optistate_amd import Engine
from optistate_amd.synth import synth_torch, Q_DEFAULT, R_DEFAULT

B, T = int(sys.argv[1]) if len(sys.argv) > 1 else 6, int(sys.argv[2]) if len(sys.argv) > 2 else 4
eng = Engine(0); eng.set_noise(Q_DEFAULT, R_DEFAULT)
d = synth_torch(B, T, "cuda", seed=5)
cp = eng.contact_soa_to_packed(d["contact"])
out = {}
for name, kw in (("rows", {}), ("lane", dict(lane_per_trajectory=True))):
    x, P = d["x0"].clone(), d["P0"].clone()
    r = eng.kf_run(d["p"], d["f"], d["dp"], d["imu"], cp, x, P, **kw)
    torch.cuda.synchronize()
    out[name] = (r["x_out"].cpu().numpy(), x.cpu().numpy(), P.cpu().numpy(), eng.kernel_name("kf"))
    print(name, out[name][3], "status", r["status"].cpu().numpy()[:8])
np.set_printoptions(precision=5, linewidth=220, suppress=True)
a, b = out["rows"][0], out["lane"][0]
for t in range(min(T, 3)):
    for tr in range(min(B, 2)):
        print(f"t {t} traj {tr}\n  rows {a[t, :, tr]}\n  lane {b[t, :, tr]}\n  imu  {d['imu'][t, :, tr].cpu().numpy()}")
print("max |x_out diff| per component", np.abs(a - b).max(axis=(0, 2)))
print("max |P diff|", np.abs(out["rows"][2] - out["lane"][2]).max())
