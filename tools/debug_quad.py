#!/usr/bin/env python3
"""Development: the sixteen-lanes-per-QP solver (mpc_quad.hip, OS_MPC_QUAD) against the wavefront-per-QP one on the same problems:
per-QP iterations, status and control differences."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from optistate_amd import Engine
from test_gpu_mpc import _problems, _solve_gpu

n = int(sys.argv[1]) if len(sys.argv) > 1 else 96
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 11
os.environ["OS_MPC_QUAD"] = "0"; e0 = Engine(0)
os.environ["OS_MPC_QUAD"] = "1"; e1 = Engine(0)
X, R, P, Cn = _problems(n, seed=seed)
r0 = _solve_gpu(e0, X, R, P, Cn); r1 = _solve_gpu(e1, X, R, P, Cn)
torch.cuda.synchronize()
u0, u1 = r0["u"].cpu().numpy().T, r1["u"].cpu().numpy().T
i0, i1 = r0["iters"].cpu().numpy(), r1["iters"].cpu().numpy()
s0, s1 = r0["status"].cpu().numpy(), r1["status"].cpu().numpy()
nst = Cn.astype(bool).sum(1)
for k in range(n):
    d = np.abs(u0[k] - u1[k]).max()
    if d > 1e-4 or s1[k] != s0[k] or i0[k] != i1[k]:
        print(f"qp {k:4d} grp {k % 4} nst {nst[k]} contact {Cn[k]}  iters {i0[k]} -> {i1[k]}  status {s0[k]} -> {s1[k]}  max |du| {d:.3e}")
print("worst", np.abs(u0 - u1).max(), "iters mean", i0.mean(), i1.mean(), "status", (s0 != 0).sum(), (s1 != 0).sum())
if len(sys.argv) > 3:
    k = int(sys.argv[3])
    k0 = (k // 4) * 4
    dev = e0.device
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a.T)).to(dev)
    c = torch.as_tensor(Cn[k0:k0 + 4]).to(dev).contiguous().view(torch.int32).reshape(-1)
    for mi in range(1, 16):
        a = e0.mpc_solve(t(X[k0:k0 + 4]), t(R[k0:k0 + 4]), t(P[k0:k0 + 4]), c, want_all=True, max_iter=mi)
        b = e1.mpc_solve(t(X[k0:k0 + 4]), t(R[k0:k0 + 4]), t(P[k0:k0 + 4]), c, want_all=True, max_iter=mi)
        ua, ub = a["u"].cpu().numpy().T[k - k0], b["u"].cpu().numpy().T[k - k0]
        print(mi, "iters", int(a["iters"][k - k0]), int(b["iters"][k - k0]), "max|du|", np.abs(ua - ub).max(), "argmax", int(np.abs(ua - ub).argmax()))
        if np.abs(ua - ub).max() > 1e-3:
            print(" wave", np.round(ua.reshape(5, 12)[:, :], 3)); print(" quad", np.round(ub.reshape(5, 12), 3)); break
    # the same QP alone in its wavefront (rows 1..3 idle): divergence between rows out of the picture
    c1 = torch.as_tensor(Cn[k:k + 1]).to(dev).contiguous().view(torch.int32).reshape(-1)
    a = e0.mpc_solve(t(X[k:k + 1]), t(R[k:k + 1]), t(P[k:k + 1]), c1, want_all=True)
    b = e1.mpc_solve(t(X[k:k + 1]), t(R[k:k + 1]), t(P[k:k + 1]), c1, want_all=True)
    print("alone: iters", int(a["iters"][0]), int(b["iters"][0]), "status", int(a["status"][0]), int(b["status"][0]), "max|du|", float((a["u"] - b["u"]).abs().max()))
    if os.environ.get("DBG_ONE"):
        a = e0.mpc_solve(t(X[k:k + 1]), t(R[k:k + 1]), t(P[k:k + 1]), c1, want_all=True, max_iter=8); torch.cuda.synchronize()
        b = e1.mpc_solve(t(X[k:k + 1]), t(R[k:k + 1]), t(P[k:k + 1]), c1, want_all=True, max_iter=8); torch.cuda.synchronize()
