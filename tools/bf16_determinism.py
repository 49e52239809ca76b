#!/usr/bin/env python3
"""Runs the fused kernels repeatedly on the same inputs and reports whether the outputs are bit-identical (development aid: a
race in hand-scheduled inline-asm MFMA code shows up as run-to-run differences)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from optistate_amd import Engine, RNN, flatten_state_dict
from optistate_amd.synth import synth_torch, Q_DEFAULT, R_DEFAULT
B, T = 65536, 100
d = synth_torch(B, T, "cuda", seed=1)
c = Engine.contact_soa_to_packed(d["contact"])
torch.manual_seed(0)
m = RNN(60, 64, 1, 24, torch.device("cpu"))
flat = flatten_state_dict(m.state_dict(), 1, "cuda")
mm = torch.stack([torch.full((60,), -30.0), torch.full((60,), 30.0)]).cuda()
for name, env, kw in (("v2", {}, {}), ("bf3", {}, {"split_bf16": True}), ("bf2", {}, {"split_bf16": 2})):
    os.environ.pop("OS_BF16_TERMS", None); os.environ.update(env)
    e = Engine(0); e.set_noise(Q_DEFAULT, R_DEFAULT); e.load_gru(flat, 60, 64, 1, 24)
    outs = []
    for r in range(6):
        x = d["x0"].clone(); P = d["P0"].clone()
        o = e.fused_run(d["p"], d["f"], d["dp"], d["imu"], c, d["accel"], mm, x, P, two_kernel=False, **kw)
        torch.cuda.synchronize()
        outs.append(o["out"].clone())
    diffs = [int((outs[i] != outs[0]).any(dim=1).sum()) for i in range(1, 6)]
    nan = [int(torch.isnan(o).any(dim=1).sum()) for o in outs]
    print(name, "trajectories differing from run 0:", diffs, "with NaN:", nan)
