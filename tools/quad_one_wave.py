#!/usr/bin/env python3
"""Development: n trot QPs through mpc_solve_quad_kernel (a -DOSQ_TS build prints workgroup 0's cycles per phase).  argv: n [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from optistate_amd import Engine
from test_gpu_mpc import _problems, _solve_gpu
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
os.environ["OS_MPC_QUAD"] = "1"; e = Engine(0)
X, R, P, Cn = _problems(4 * n, seed=int(sys.argv[2]) if len(sys.argv) > 2 else 11)
keep = np.where(Cn.astype(bool).sum(1) == 2)[0][:n]
X, R, P, Cn = X[keep], R[keep], P[keep], Cn[keep]
for _ in range(2):
    r = _solve_gpu(e, X, R, P, Cn); torch.cuda.synchronize()
print("problems", len(keep), "iters", r["iters"].cpu().numpy()[:16])
