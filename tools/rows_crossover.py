#!/usr/bin/env python3
"""Development aid: Kalman-only throughput of the 16-lanes-per-trajectory kernel against the lane-per-trajectory kernel over the
batch sizes around their crossover (OS_KF_ROWS_BELOW sets the switch; the default lives in capi.hip)."""
import os, sys, time
os.environ["OS_KF_ROWS_BELOW"] = "1000000"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from optistate_amd import Engine
from optistate_amd.synth import synth_torch, Q_DEFAULT, R_DEFAULT

T = 400
eng = Engine(0); eng.set_noise(Q_DEFAULT, R_DEFAULT)
print("B      rows steps/s   lane steps/s   (T = %d)" % T)
for B in (4096, 6144, 8192, 10240, 12288, 14336, 16384, 20480, 24576, 32768):
    d = synth_torch(B, T, "cuda", seed=3)
    cp = eng.contact_soa_to_packed(d["contact"])
    res = []
    for kw in ({}, dict(lane_per_trajectory=True)):
        best = 1e9
        for rep in range(4):
            x, P = d["x0"].clone(), d["P0"].clone()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            eng.kf_run(d["p"], d["f"], d["dp"], d["imu"], cp, x, P, **kw)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        res.append((B * T / best, eng.kernel_name("kf")))
    print(f"{B:6d} {res[0][0]:.3e} {res[1][0]:.3e}   {res[0][1]} | {res[1][1]}")
