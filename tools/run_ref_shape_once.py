#!/usr/bin/env python3
"""Runs the reference's real model shape (GRU(188,128,4,24): 60 Kalman features + 128-d latent) a few times at B = 65,536, T = 100
(for rocprofv3 counter passes: FETCH_SIZE / WRITE_SIZE of gru_layer_kernel and kf_run_sym_kernel<FEAT>)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from optistate_amd import Engine, RNN, flatten_state_dict
from optistate_amd.synth import synth_torch, Q_DEFAULT, R_DEFAULT
B, T, NL = 65536, 100, 128
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
eng = Engine(0); eng.set_noise(Q_DEFAULT, R_DEFAULT)
d = synth_torch(B, T, "cuda", seed=1)
c = eng.contact_soa_to_packed(d["contact"])
torch.manual_seed(0)
m = RNN(60 + NL, 128, 4, 24, torch.device("cuda"))
eng.load_gru(flatten_state_dict(m.state_dict(), 4, "cuda"), 60 + NL, 128, 4, 24)
mm = torch.stack([torch.full((60,), -30.0), torch.full((60,), 30.0)]).cuda()
buf = torch.empty((T, 60 + NL, B), device="cuda"); buf[:, 60:] = torch.rand((T, NL, B), device="cuda")
for _ in range(n):
    x = d["x0"].clone(); P = d["P0"].clone()
    eng.fused_run(d["p"], d["f"], d["dp"], d["imu"], c, d["accel"], mm, x, P, gru_input=buf)
torch.cuda.synchronize()
