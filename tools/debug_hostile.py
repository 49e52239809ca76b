#!/usr/bin/env python3
"""Development aid: hostile inputs + a noise set through the Kalman kernels of the library named by OPTISTATE_HIP_LIB (default: the
in-tree build) against the float64 oracle; prints the worst trajectories and their error history."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from optistate_amd import Engine
from optistate_amd.synth import synth_torch, NOISE_SETS
from oracle import c_oracle as orc

noise = sys.argv[1] if len(sys.argv) > 1 else "fitted"
B, T = 65536, 100
Q, R = NOISE_SETS[noise]
eng = Engine(0); eng.set_noise(Q, R)
d = synth_torch(B, T, "cuda", seed=77, hostile=True)
cp = eng.contact_soa_to_packed(d["contact"])
P0 = torch.tensor(np.asarray(Q, dtype=np.float32).reshape(144, 1), device="cuda").repeat(1, B).contiguous()
x, P = d["x0"].clone(), P0.clone()
r = eng.kf_run(d["p"], d["f"], d["dp"], d["imu"], cp, x, P)
print("lib", os.environ.get("OPTISTATE_HIP_LIB", "in-tree"), "kernel", eng.kernel_name("kf"), "status nonzero", int((r["status"] != 0).sum()))
pick = torch.arange(16384, 20480, device="cuda")                 # the theta_y = pi/2 exact-start block
g = lambda k: d[k][:, :, pick].permute(2, 0, 1).double().cpu().numpy()
n = int(pick.numel())
orc.set_threads(orc.max_threads())
ref = orc.kf_run_batch(g("p"), g("f"), g("dp"), g("imu"), d["contact"][:, :, pick].permute(2, 0, 1).contiguous().cpu().numpy(),
                       d["x0"][:, pick].t().double().cpu().numpy(), np.tile(Q, (n, 1, 1)), Q, R)
xo = r["x_out"][:, :, pick].permute(2, 0, 1).cpu().numpy()
err = np.abs(xo - ref["x"])
print("max err", err.max(), "at", np.unravel_index(int(err.argmax()), err.shape))
worst = np.argsort(-err.reshape(n, -1).max(1))[:5]
imu = g("imu")
for w in worst:
    e = err[w]
    t0 = int(np.argmax(e.max(1) > 1e-4)) if (e.max(1) > 1e-4).any() else -1
    print(f"traj {int(pick[w])}: max {e.max():.3e}, first step above 1e-4: {t0}")
    if t0 >= 0:
        for t in range(max(0, t0 - 1), min(T, t0 + 3)):
            print("   t", t, "err", np.array2string(e[t], precision=2, max_line_width=200))
            print("      gpu x", np.array2string(xo[w, t], precision=6, max_line_width=200))
            print("      ref x", np.array2string(ref["x"][w, t], precision=6, max_line_width=200))
            print("      imu  ", np.array2string(imu[w, t], precision=6, max_line_width=200), "contact", d["contact"][t, :, int(pick[w])].cpu().numpy())
