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
# cold starts (a trajectory whose force-carrying legs changed since the previous step: the warm record is not reused) against warm ones
cw = c.reshape(T, -1).cpu().numpy().astype("uint32")
import numpy as np
def ranks(w):
    out = np.zeros_like(w); n = np.zeros(w.shape, dtype=np.uint32)
    for l in range(4):
        byte = (w >> np.uint32(8 * l)) & np.uint32(0xff)
        nz = byte != 0
        out = np.where(nz, out | (byte << (np.uint32(8) * n)), out); n = n + nz.astype(np.uint32)
    return out
rk = ranks(cw)
cold = np.ones((T, cw.shape[1]), dtype=bool); cold[1:] = rk[1:] != rk[:-1]
itn = it.cpu().numpy()
for t in (1, 5, 10, 15):
    cc = cold[t]
    print(f"step {t}: cold {cc.mean():.4f} of the batch; iterations cold mean {itn[t][cc].mean() if cc.any() else 0:.1f} max {itn[t][cc].max() if cc.any() else 0:.0f} | warm mean {itn[t][~cc].mean():.2f} max {itn[t][~cc].max():.0f}; "
          f"of the problems above 12 iterations {((itn[t] > 12) & cc).sum()} cold, {((itn[t] > 12) & ~cc).sum()} warm")
chg = np.zeros_like(cold); chg[1:] = cw[1:] != cw[:-1]
for t in (5, 10, 15):
    cc = chg[t]
    print(f"step {t}: contact word changed for {cc.mean():.4f}; iterations changed mean {itn[t][cc].mean() if cc.any() else 0:.1f} | unchanged mean {itn[t][~cc].mean():.2f} max {itn[t][~cc].max():.0f}; "
          f"above 12 iterations: {((itn[t] > 12) & cc).sum()} changed, {((itn[t] > 12) & ~cc).sum()} unchanged; previous step's count of those above 12: mean {itn[t-1][itn[t] > 12].mean():.1f}; "
          f"P(>12 | prev > 12) = {((itn[t] > 12) & (itn[t-1] > 12)).sum() / max(1, (itn[t-1] > 12).sum()):.3f}, P(>12 | prev <= 3) = {((itn[t] > 12) & (itn[t-1] <= 3)).sum() / max(1, (itn[t-1] <= 3).sum()):.4f}")
