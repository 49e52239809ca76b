"""Development probe: os_kf_odom -> os_kf_predict -> os_kf_update (row-layout single pieces) against one step of os_kf_run."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optistate_amd import Engine
from optistate_amd.synth import synth_torch, Q_FITTED, R_FITTED
np.set_printoptions(linewidth=220, precision=4)
eng = Engine(0); eng.set_noise(Q_FITTED, R_FITTED)
B = 5
d = synth_torch(B, 1, torch.device("cuda"), seed=3)
c = eng.contact_soa_to_packed(d["contact"])
for dense in (False, True):
    for seq in (True, False):
        x1, P1 = d["x0"].clone(), d["P0"].clone()
        br = torch.zeros((1, 12, B), device="cuda"); br[0, 0:3] = d["imu"][0, 0:3] + 0.01
        r1 = eng.kf_run(d["p"], d["f"], d["dp"], d["imu"], c, x1, P1, body_ref=br if dense else None, dense_fd=dense, sequential=seq, symmetric=False,
                        lane_per_trajectory=True, want_trace=True, want_gain=True)
        x2, P2 = d["x0"].clone(), d["P0"].clone()
        p2 = d["p"][0].clone()
        z = eng.kf_odom(p2, d["dp"][0].contiguous(), c[0].contiguous(), d["imu"][0].contiguous())
        eng.kf_predict(p2, d["f"][0].contiguous(), x2, P2, body_ref=br[0].contiguous() if dense else None)
        Pp = P2.clone(); xp = x2.clone()
        r2 = eng.kf_update(z, x2, P2, sequential=seq, want_K=True)
        torch.cuda.synchronize()
        print(f"dense={dense} seq={seq}: x diff {float((x1 - x2).abs().max()):.3e}  P diff {float((P1 - P2).abs().max()):.3e} (|P| {float(P1.abs().max()):.3e})  "
              f"ptrace {float((r1['P_trace'][0] - r2['P_trace']).abs().max()):.3e}  kgain {float((r1['K_gain'][0] - r2['K_gain']).abs().max()):.3e}  status {r2['status'].cpu().numpy()}")
        if float((P1 - P2).abs().max()) > 1e-3 * float(P1.abs().max()):
            print("  P after predict, trajectory 0, rows 0..2:\n", Pp[:, 0].cpu().numpy().reshape(12, 12)[:3])
            print("  final P kf_run rows 0..2:\n", P1[:, 0].cpu().numpy().reshape(12, 12)[:3], "\n  pieces:\n", P2[:, 0].cpu().numpy().reshape(12, 12)[:3])
