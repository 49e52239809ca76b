#!/usr/bin/env python3
"""gru_wide_kernel (four CUs per (layer, tile)) against gru_stack_kernel (one) on small batches of the reference's model:
device time per forward / per training forward + backward, and the distance of both to the float64 oracle.
usage: python3 tools/wide_probe.py [reps]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optistate_amd import Engine, RNN, flatten_state_dict          # noqa: E402
from oracle import c_oracle as orc                                  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
I, H, L, C = 188, 128, 4, 24
torch.manual_seed(0)
m = RNN(I, H, L, C, torch.device("cpu"))
flat = flatten_state_dict(m.state_dict(), L, "cuda")
wref = orc.flatten_state_dict(m.state_dict(), L)
print("| B | T | kernel | forward, us | train forward, us | l-inf vs float64 |\n|---|---|---|---|---|---|")
for B, T in ((8, 10), (64, 10), (128, 10), (256, 10), (512, 10), (64, 100)):
    x = torch.rand(B, T, I)
    ref, _, _ = orc.gru_forward(x[:64].numpy(), wref, I, H, L, C)
    for wide in ("1", "0"):
        os.environ["OS_GRU_WIDE"] = wide
        e = Engine(0)
        e.load_gru(flat, I, H, L, C)
        xg = x.cuda()
        out = e.gru_forward(xg)
        torch.cuda.synchronize()
        name = e.kernel_name("gru_layer")
        err = float(np.abs(out.cpu().numpy()[:64] - ref).max())
        t0 = time.perf_counter()
        for _ in range(reps):
            e.gru_forward(xg)
        torch.cuda.synchronize()
        fwd = (time.perf_counter() - t0) / reps * 1e6
        o = e.gru_forward_train(xg)
        torch.cuda.synchronize()
        errt = float(np.abs(o.cpu().numpy()[:64] - ref).max())
        t0 = time.perf_counter()
        for _ in range(reps):
            e.gru_forward_train(xg)
        torch.cuda.synchronize()
        trn = (time.perf_counter() - t0) / reps * 1e6
        print(f"| {B} | {T} | {name} | {fwd:.1f} | {trn:.1f} ({e.kernel_name('gru_layer')}) | {err:.1e} / {errt:.1e} |", flush=True)
        e.close()
