#!/usr/bin/env python3
"""Runs the small-batch Kalman kernel a few times at BASELINE configs[1] (B = 4096, T = 1000) for rocprofv3 counter passes
(tools/pmc_any.sh <tag> "<counters>" tools/run_rows_once.py)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from optistate_amd import Engine
from optistate_amd.synth import synth_torch, Q_DEFAULT, R_DEFAULT
B, T = 4096, 1000
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
eng = Engine(0); eng.set_noise(Q_DEFAULT, R_DEFAULT)
d = synth_torch(B, T, "cuda", seed=404)
c = eng.contact_soa_to_packed(d["contact"])
for _ in range(n):
    x = d["x0"].clone(); P = d["P0"].clone()
    eng.kf_run(d["p"], d["f"], d["dp"], d["imu"], c, x, P)
torch.cuda.synchronize()
