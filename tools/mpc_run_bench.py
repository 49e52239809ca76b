#!/usr/bin/env python3
"""Timing of Engine.kf_mpc_run (estimate_state_mpc over B x T).  argv: B T"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from optistate_amd import Engine
from optistate_amd.synth import synth_torch, Q_DEFAULT, R_DEFAULT

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
T = int(sys.argv[2]) if len(sys.argv) > 2 else 100
eng = Engine(0); eng.set_noise(Q_DEFAULT, R_DEFAULT)
d = synth_torch(B, T, "cuda", seed=1)
c = eng.contact_soa_to_packed(d["contact"])
ref = torch.zeros((T, 12, B), device="cuda"); ref[:, 5] = 0.28; ref[:, 9] = 0.1
tt = torch.arange(T, device="cuda")[:, None] * 0.01
ref[:, 0] = 0.02 * torch.sin(3 * tt); ref[:, 1] = 0.02 * torch.cos(2 * tt)
def run():
    x = d["x0"].clone(); P = d["P0"].clone()
    return eng.kf_mpc_run(d["p"], d["dp"], d["imu"], c, ref, x, P, want_iters=True, sequential=(os.environ.get("SEQ", "0") == "1"))
r = run(); torch.cuda.synchronize()
t0 = time.time(); r = run(); torch.cuda.synchronize(); dt = time.time() - t0
it = r["iters"].float()
print(f"B={B} T={T}: {dt*1e3:.1f} ms -> {B*T/dt:.3e} steps/s; QP iters mean {it.mean():.2f} max {int(it.max())}; status nonzero {int((r['status']!=0).sum())}; |f| max {float(r['f'].abs().max()):.1f}")
