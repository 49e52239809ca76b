"""Development probe: kf_dense_rows_kernel against the C oracle's predict_mpc path, step by step (x_out per step, final P)."""
import sys
import numpy as np
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from optistate_amd import Engine
from optistate_amd.synth import synth_numpy, Q_FITTED, R_FITTED, Q_DEFAULT, R_DEFAULT
from oracle import c_oracle as orc

eng = Engine(0)
for (Q, R, tag) in ((Q_DEFAULT, R_DEFAULT, "default"), (Q_FITTED, R_FITTED, "fitted")):
    for T in (1, 2, 5, 40):
        B = 3
        d = synth_numpy(B, T, seed=7)
        d["body_ref"] = np.zeros((B, T, 12), dtype=np.float32); d["body_ref"][..., 0:3] = d["imu"][..., 0:3] + 0.01
        ref = orc.kf_run_batch(d["p"], d["f"], d["dp"], d["imu"], d["contact"], d["x0"], np.tile(Q, (B, 1, 1)), Q, R, body_ref=d["body_ref"], mode=1)
        s = {k: eng.pack(torch.as_tensor(d[k])) for k in ("p", "f", "dp", "imu", "body_ref")}
        c = eng.pack_contact(torch.as_tensor(d["contact"]))
        eng.set_noise(Q, R)
        for seq in (False, True):
            x = torch.as_tensor(d["x0"].T.copy()).cuda()
            P = torch.as_tensor(np.tile(Q.astype(np.float32).reshape(144, 1), (1, B))).cuda()
            r = eng.kf_run(s["p"], s["f"], s["dp"], s["imu"], c, x, P, body_ref=s["body_ref"], dense_fd=True, sequential=seq, want_trace=True, want_gain=True)
            xo = eng.unpack(r["x_out"]).cpu().numpy()
            Pf = P.cpu().numpy().T.reshape(B, 12, 12)
            ex = np.abs(xo - ref["x"]).max(axis=(0, 2))
            print(f"{tag} T={T} {'seq' if seq else 'batch'}: x err per step (first 5) {ex[:5]}, max {ex.max():.2e}; P_final rel err "
                  f"{np.abs(Pf - ref['P_final']).max() / np.abs(ref['P_final']).max():.2e}; ptrace rel {np.abs(r['P_trace'].cpu().numpy().T / ref['P_trace'] - 1).max():.2e}; "
                  f"kgain abs {np.abs(r['K_gain'].cpu().numpy().T - ref['K_gain']).max():.2e}; status {r['status'].cpu().numpy()}")
            if T == 1 and not seq:
                print("  P_final[0] row0 gpu", Pf[0, 0, :6], "\n  P_final[0] row0 ref", ref["P_final"][0, 0, :6])
                print("  P_final[0] row7 gpu", Pf[0, 7, 4:10], "\n  P_final[0] row7 ref", ref["P_final"][0, 7, 4:10])
