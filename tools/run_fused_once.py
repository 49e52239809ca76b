#!/usr/bin/env python3
"""Runs the fused path a few times at the bench shape (for rocprofv3 counter passes).  argv: n [split_bf16: 0 | 3 | 2] [B] [tile shape]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from optistate_amd import Engine, RNN, flatten_state_dict
from optistate_amd.synth import synth_torch, Q_DEFAULT, R_DEFAULT
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
split = int(sys.argv[2]) if len(sys.argv) > 2 else 0
B, T = (int(sys.argv[3]) if len(sys.argv) > 3 else 65536), 100
eng = Engine(0); eng.set_noise(Q_DEFAULT, R_DEFAULT)
if len(sys.argv) > 4:
    eng.set_fused_tile(int(sys.argv[4]))
d = synth_torch(B, T, "cuda", seed=1)
c = eng.contact_soa_to_packed(d["contact"])
torch.manual_seed(0)
m = RNN(60, 64, 1, 24, torch.device("cuda"))
eng.load_gru(flatten_state_dict(m.state_dict(), 1, "cuda"), 60, 64, 1, 24)
mm = torch.stack([torch.full((60,), -30.0), torch.full((60,), 30.0)]).cuda()
for _ in range(n):
    x = d["x0"].clone(); P = d["P0"].clone()
    eng.fused_run(d["p"], d["f"], d["dp"], d["imu"], c, d["accel"], mm, x, P, split_bf16=split)
    if split or len(sys.argv) > 3:
        continue
    x = d["x0"].clone(); P = d["P0"].clone()
    eng.kf_run(d["p"], d["f"], d["dp"], d["imu"], c, x, P)
torch.cuda.synchronize()
