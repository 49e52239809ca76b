#!/usr/bin/env python3
"""Median device time of the fused kernel (library HIP events) at one batch / tile shape: `python tools/time_fused.py [B] [tile] [rounds]`.
With OPTISTATE_HIP_LIB pointing at another build of the library this is one arm of an A/B (tools/ab_libs.sh alternates the arms)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from optistate_amd import Engine, RNN, flatten_state_dict
from optistate_amd.synth import synth_torch, Q_DEFAULT, R_DEFAULT

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
tile = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 12
T = 100
eng = Engine(0); eng.set_noise(Q_DEFAULT, R_DEFAULT); eng.set_fused_tile(tile)
d = synth_torch(B, T, "cuda", seed=1)
c = eng.contact_soa_to_packed(d["contact"])
torch.manual_seed(0)
m = RNN(60, 64, 1, 24, torch.device("cuda"))
eng.load_gru(flatten_state_dict(m.state_dict(), 1, "cuda"), 60, 64, 1, 24)
mm = torch.stack([torch.full((60,), -30.0), torch.full((60,), 30.0)]).cuda()
eng.profile(True)
ms = []
for r in range(rounds + 3):
    x = d["x0"].clone(); P = d["P0"].clone()
    eng.fused_run(d["p"], d["f"], d["dp"], d["imu"], c, d["accel"], mm, x, P)
    torch.cuda.synchronize()
    pr = eng.profile_read()
    if r >= 3:
        ms.append(pr["fused"][0] / max(pr["fused"][1], 1))
v = np.array(ms)
print(f"{os.environ.get('OPTISTATE_HIP_LIB', 'shipped')}: B {B} {eng.kernel_name('fused')}: median {np.median(v):.4f} ms  min {v.min():.4f}  max {v.max():.4f}")
