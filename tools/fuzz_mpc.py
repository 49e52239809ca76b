"""GPU: random force-QP problems (every contact pattern, disturbances from 0.1x to 10x the suite's, random references and foot
positions) through os_mpc_solve against the KKT-certified float64 oracle (oracle/mpc_oracle.py).
    python tools/fuzz_mpc.py [n_problems] [seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from optistate_amd import Engine  # noqa: E402
from oracle import mpc_oracle as mo  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    eng = Engine(0)
    mo.MASS = float(np.float32(8.8))
    mo.INERTIA = np.asarray(np.float32([0.05530364, 0.06011944, 0.10530434]), np.float64)
    X, R, P, Cn = [], [], [], []
    for t in range(n):
        s = float(rng.choice([0.1, 0.3, 1.0, 3.0, 6.0, 10.0]))
        X.append(np.array([0, 0, 0, 0, 0, 0.28, 0, 0, 0, 0, 0, 0.]) + s * rng.normal(0, [0.05] * 3 + [0.02] * 3 + [0.2] * 3 + [0.1] * 3))
        R.append(np.array([0, 0, 0, 0, 0, 0.28, 0, 0, 0, 0.1, 0, 0.]) + s * rng.normal(0, [0.02] * 3 + [0.01] * 3 + [0.05] * 3 + [0.05] * 3))
        P.append(np.array([0.2, 0.1, -0.28, 0.2, -0.1, -0.28, -0.2, 0.1, -0.28, -0.2, -0.1, -0.28]) + rng.normal(0, 0.03, 12))
        w = int(rng.integers(0, 16))
        Cn.append([(w >> l) & 1 for l in range(4)])
    f32 = lambda a: np.asarray(a, np.float32)
    X, R, P, Cn = f32(X), f32(R), f32(P), np.asarray(Cn, np.uint8)
    dev = eng.device
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a.T)).to(dev)
    c = torch.as_tensor(Cn).to(dev).contiguous().view(torch.int32).reshape(-1)
    r = eng.mpc_solve(t(X), t(R), t(P), c, want_all=True)
    u = r["u"].cpu().numpy().T.astype(np.float64)
    st = r["status"].cpu().numpy()
    it = r["iters"].cpu().numpy()
    worst, bad = 0.0, 0
    for k in range(n):
        f_o, u_o, info = mo.mpc_forces(X[k].astype(np.float64), R[k].astype(np.float64), P[k].astype(np.float64), Cn[k], dt=float(np.float32(0.01)))
        e = float(np.abs(u[k] - u_o).max())
        worst = max(worst, e)
        if e > 2e-4 or st[k] != 0 or info["stationarity"] > 1e-8:
            bad += 1
            print(f"problem {k}: contact {Cn[k].tolist()} |u - u_oracle| {e:.2e} N, status {st[k]}, iterations {it[k]}, oracle stationarity {info['stationarity']:.1e}   <-- ABOVE THE BAR")
    print(f"{n} problems, {bad} above the bars; worst control distance {worst:.2e} N; iterations max {int(it.max())} mean {it.mean():.1f}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
