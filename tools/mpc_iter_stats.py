#!/usr/bin/env python3
"""Active-set iteration statistics of os_kf_mpc_run on the bench's data: per step the mean over the batch and the mean of the maximum
over the four consecutive trajectories that share a wavefront in the sixteen-lanes-per-QP solver (mpc_quad.hip): what lockstep costs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from optistate_amd import Engine
from optistate_amd.synth import synth_torch, Q_DEFAULT, R_DEFAULT
B, T = int(sys.argv[1]) if len(sys.argv) > 1 else 65536, int(sys.argv[2]) if len(sys.argv) > 2 else 20
eng = Engine(0); eng.set_noise(Q_DEFAULT, R_DEFAULT)
d = synth_torch(B, T, "cuda", seed=1000)
c = eng.contact_soa_to_packed(d["contact"])
ref = torch.zeros((T, 12, B), device="cuda"); ref[:, 5] = 0.28; ref[:, 9] = 0.1
x, P = d["x0"].clone(), d["P0"].clone()
r = eng.kf_mpc_run(d["p"], d["dp"], d["imu"], c, ref, x, P, want_iters=True)
it = r["iters"].float()
print("step  mean  mean(max of 4)  max")
for t in range(T):
    print(f"{t:3d} {it[t].mean().item():6.2f} {it[t].reshape(-1, 4).max(1).values.mean().item():6.2f} {int(it[t].max())}")
print("all", it.mean().item(), it.reshape(T, -1, 4).max(2).values.mean().item())
h = torch.bincount(it[1:].flatten().long(), minlength=12)[:16]
print("histogram of warm steps (iterations 0..15):", (h / h.sum()).cpu().numpy().round(3))
# how well does a trajectory's count at step t predict its count at step t + 1?  (the lock-step cost of four rows sharing a wavefront
# disappears if the four have similar counts: sorting a step's problems by the previous step's count)
a, b = it[1:-1].flatten(), it[2:].flatten()
print("correlation of consecutive steps' counts:", float(torch.corrcoef(torch.stack([a, b]))[0, 1]))
for t in (3, 10):
    order = torch.argsort(it[t - 1])
    srt = it[t][order]
    print(f"step {t}: mean {it[t].mean().item():.2f}, mean(max of 4) unsorted {it[t].reshape(-1, 4).max(1).values.mean().item():.2f}, "
          f"sorted by step {t - 1}'s count {srt.reshape(-1, 4).max(1).values.mean().item():.2f}")
