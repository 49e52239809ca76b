#!/usr/bin/env python3
"""K_gain (np.trace(K), kalman_filter.py:174) of every Kalman kernel family against G3, both noise sets: max abs error (GPU)."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from test_gpu_kf import VARIANTS, VIDS, run, load_golden
from optistate_amd import Engine
eng = Engine(0); g = load_golden("kf_g3_traj.npz")
for s in (0, 1):
    for v, n in zip(VARIANTS, VIDS):
        r = run(eng, g, g[f"Q{s}"], g[f"R{s}"], 2, want_p_rot=True, want_trace=True, want_gain=True, **v)
        e = max(np.abs(r["K_gain"].cpu().numpy()[:, b] - g[f"s{s}_b{b}_K_gain"]).max() for b in range(2))
        print(s, n, f"{e:.2e}", float(np.abs(g[f"s{s}_b0_K_gain"]).max()))
