#!/usr/bin/env python3
"""Runs the GRU layer stack a few times (for rocprofv3 counter passes).  argv: n H L"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from optistate_amd import Engine, RNN, flatten_state_dict
B, T = 65536, 100
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
H = int(sys.argv[2]) if len(sys.argv) > 2 else 128
L = int(sys.argv[3]) if len(sys.argv) > 3 else 2
eng = Engine(0)
torch.manual_seed(0)
m = RNN(60, H, L, 24, torch.device("cuda"))
eng.load_gru(flatten_state_dict(m.state_dict(), L, "cuda"), 60, H, L, 24)
xs = torch.rand(T, 60, B, device="cuda")
for _ in range(n):
    eng.gru_forward_soa(xs)
torch.cuda.synchronize()
