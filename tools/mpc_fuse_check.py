#!/usr/bin/env python3
"""os_kf_mpc_run with the filter step inside the QP launch (default) against the separate launches (OS_MPC_FUSE_KF=0): the same
building blocks in the same order -- x_out, f, P, status and the iteration counts have to be identical.  argv: B T [repeats]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from optistate_amd import Engine
from optistate_amd.synth import synth_torch, Q_DEFAULT, R_DEFAULT

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
T = int(sys.argv[2]) if len(sys.argv) > 2 else 12
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
eng = Engine(0); eng.set_noise(Q_DEFAULT, R_DEFAULT)
d = synth_torch(B, T, "cuda", seed=1000)
c = eng.contact_soa_to_packed(d["contact"])
ref = torch.zeros((T, 12, B), device="cuda"); ref[:, 5] = 0.28; ref[:, 9] = 0.1
def run(fuse):
    os.environ["OS_MPC_FUSE_KF"] = "1" if fuse else "0"
    x, P = d["x0"].clone(), d["P0"].clone()
    torch.cuda.synchronize(); t0 = time.time()
    r = eng.kf_mpc_run(d["p"], d["dp"], d["imu"], c, ref, x, P, want_iters=True)
    torch.cuda.synchronize(); dt = time.time() - t0
    return r, x, P, dt
ok = True
for k in range(reps):
    r0, x0, P0, t0 = run(False)
    r1, x1, P1, t1 = run(True)
    same = {n: bool(torch.equal(r0[n], r1[n])) for n in ("x_out", "f", "iters", "status")}
    same["x"] = bool(torch.equal(x0, x1)); same["P"] = bool(torch.equal(P0, P1))
    ok = ok and all(same.values())
    print(f"B={B} T={T}: separate {t0*1e3:.1f} ms, inside the QP launch {t1*1e3:.1f} ms ({B*T/t1:.3e} steps/s); identical: {same}; status nonzero {int((r1['status'] != 0).sum())}", flush=True)
sys.exit(0 if ok else 1)
