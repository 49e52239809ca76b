#!/usr/bin/env python3
"""Randomised stress of os_mpc_solve against oracle/mpc_oracle.py (development aid; the oracle is test infrastructure).
argv: N seed.  Reports worst |u - u_oracle|, iteration statistics, status flags, oracle failures."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from optistate_amd import Engine
from oracle import mpc_oracle as mo

N = int(sys.argv[1]) if len(sys.argv) > 1 else 500
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
mo.MASS = float(np.float32(8.8)); mo.INERTIA = np.asarray(np.float32([0.05530364, 0.06011944, 0.10530434]), np.float64)
rng = np.random.default_rng(seed)
X, R, P, Cn = [], [], [], []
for t in range(N):
    s = [0.1, 0.5, 1.0, 3.0, 6.0, 10.0][t % 6]
    X.append(np.array([0, 0, 0, 0, 0, 0.28, 0, 0, 0, 0, 0, 0.]) + s * rng.normal(0, [0.05] * 3 + [0.02] * 3 + [0.2] * 3 + [0.1] * 3))
    R.append(np.array([0, 0, 0, 0, 0, 0.28, 0, 0, 0, 0.1, 0, 0.]) + s * rng.normal(0, [0.02] * 3 + [0.01] * 3 + [0.05] * 3 + [0.05] * 3))
    P.append(np.array([0.2, 0.1, -0.28, 0.2, -0.1, -0.28, -0.2, 0.1, -0.28, -0.2, -0.1, -0.28]) + rng.normal(0, 0.02, 12))
    c = rng.integers(0, 2, 4)
    if rng.random() < 0.05: c[rng.integers(0, 4)] = 2
    Cn.append(c)
X, R, P = (np.asarray(a, np.float32) for a in (X, R, P)); Cn = np.asarray(Cn, np.uint8)
eng = Engine(0)
t = lambda a: torch.as_tensor(np.ascontiguousarray(a.T)).cuda()
c = torch.as_tensor(Cn).cuda().contiguous().view(torch.int32).reshape(-1)
r = eng.mpc_solve(t(X), t(R), t(P), c, want_all=True)
u = r["u"].cpu().numpy().T.astype(np.float64); it = r["iters"].cpu().numpy(); st = r["status"].cpu().numpy()
worst, fails, t0 = 0.0, 0, time.time()
for k in range(N):
    try:
        _, uo, info = mo.mpc_forces(X[k].astype(np.float64), R[k].astype(np.float64), P[k].astype(np.float64), Cn[k], dt=float(np.float32(0.01)))
    except RuntimeError as e:
        fails += 1; continue
    d = np.abs(u[k] - uo).max()
    if d > 1e-3: print("  problem", k, "diff", d, "contact", Cn[k], "iters", it[k], "status", st[k])
    worst = max(worst, d)
print(f"N={N} seed={seed}: worst |u-u_oracle| = {worst:.3e} N; iters mean {it.mean():.1f} max {it.max()}; status nonzero {int((st != 0).sum())}; oracle failures {fails}; {time.time()-t0:.0f}s")
