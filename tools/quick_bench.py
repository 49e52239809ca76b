#!/usr/bin/env python3
"""Quick device timing of the individual entry points (development aid; bench.py is the judged harness)."""
import argparse
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from optistate_amd import Engine, RNN, flatten_state_dict
from optistate_amd.synth import synth_torch, Q_DEFAULT, R_DEFAULT

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=65536)
ap.add_argument("--T", type=int, default=100)
ap.add_argument("--H", type=int, default=64)
ap.add_argument("--L", type=int, default=1)
ap.add_argument("--iters", type=int, default=3)
a = ap.parse_args()
eng = Engine(0)
d = synth_torch(a.B, a.T, "cuda", seed=1)
c = eng.contact_soa_to_packed(d["contact"])
eng.set_noise(Q_DEFAULT, R_DEFAULT)


def timeit(fn, n=a.iters):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


steps = a.B * a.T
for seq, sym, lanes in ((True, True, False), (True, True, True), (True, False, True), (False, False, True)):
    def kf():
        x = d["x0"].clone(); P = d["P0"].clone()
        eng.kf_run(d["p"], d["f"], d["dp"], d["imu"], c, x, P, sequential=seq, symmetric=sym, lane_per_trajectory=lanes)
    ms = timeit(kf)
    print(f"kf_run seq={seq} sym={sym} force_lanes={lanes}: {ms:.3f} ms  {steps / ms * 1e3:.3e} steps/s  {steps * 220 / ms * 1e3 / 1e9:.1f} GB/s algorithmic")

torch.manual_seed(0)
m = RNN(60, a.H, a.L, 24, torch.device("cuda"))
eng.load_gru(flatten_state_dict(m.state_dict(), a.L, "cuda"), 60, a.H, a.L, 24)
xs = torch.rand(a.T, 60, a.B, device="cuda")
ms = timeit(lambda: eng.gru_forward_soa(xs))
fl = sum(2 * 3 * a.H * ((60 if l == 0 else a.H) + a.H) for l in range(a.L))
print(f"gru_forward_soa H={a.H} L={a.L}: {ms:.3f} ms  {steps / ms * 1e3:.3e} steps/s  {steps * fl / ms * 1e3 / 1e12:.1f} TFLOP/s")
mm = torch.stack([torch.full((60,), -30.0), torch.full((60,), 30.0)]).cuda()


for tk in (False, True):
    def fused():
        x = d["x0"].clone(); P = d["P0"].clone()
        eng.fused_run(d["p"], d["f"], d["dp"], d["imu"], c, d["accel"], mm, x, P, two_kernel=tk)
    ms = timeit(fused)
    print(f"fused_run two_kernel={tk}: {ms:.3f} ms  {steps / ms * 1e3:.3e} steps/s")
