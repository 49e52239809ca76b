#!/usr/bin/env python3
"""End-to-end walk through the reference's pipeline with the drop-in pieces, on synthetic trajectories (the reference's
recorded `.mat` data is not redistributable):

  1. data_conversion_Kalman_to_Training.py : Kalman filter over every trajectory -> 60-feature rows      (one batched launch)
  2. gru_train.py                          : min-max scaling, windows of 10, Adam on the self-referential target
  3. gru_test.py                           : sliding-window inference, de-normalised predictions with error bands

Run on an MI355X:  python examples/pipeline_demo.py [--traj 64 --steps 400 --epochs 3] [--mpc]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optistate_amd import Engine, RNN                      # noqa: E402
from optistate_amd import pipeline as pl                   # noqa: E402
from optistate_amd.synth import synth_numpy, Q_FITTED, R_FITTED   # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--traj", type=int, default=64)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--hidden", type=int, default=128)
    ap.add_argument("--layers", type=int, default=4)
    ap.add_argument("--mpc", action="store_true",
                    help="ground-reaction forces from the convex MPC (KF2.estimate_state_mpc, as the reference's script runs it) "
                         "instead of a force log")
    a = ap.parse_args(argv)
    dev = torch.device("cuda")
    eng = Engine(0)

    # 1. Kalman filter over all trajectories at once; "mocap" ground truth = the clean synthetic state + noise-free copy
    d = synth_numpy(a.traj, a.steps, seed=0)
    if a.mpc:
        # data_conversion_Kalman_to_Training.py:194-199: x = KF2.estimate_state_mpc(imu, p, dp, x_ref, contact_ref)
        ref = np.zeros((a.traj, a.steps, 12), np.float32); ref[..., 5] = 0.28; ref[..., 9] = 0.1
        d["ref"] = ref
        rows, x_hist, forces, status = pl.kalman_feature_rows_mpc(eng, d, Q_FITTED, R_FITTED, d["x0"])
        print(f"MPC forces: |f| max {float(forces.abs().max()):.1f} N")
    else:
        rows, x_hist, status = pl.kalman_feature_rows(eng, d, Q_FITTED, R_FITTED, d["x0"])
    assert int(eng.failed(status).sum()) == 0          # bits 0-3; bit 4 (truncation knife edge) is informational
    mocap = x_hist + 0.01 * torch.randn_like(x_hist)          # stand-in labels with the same 12-state layout

    # 2. scaling + windows (every trajectory contributes its own windows), training with the reference's loop
    mn, mx = pl.fit_minmax(rows)
    mn_v, mx_v = pl.fit_minmax(mocap)
    wins, labs = [], []
    for b in range(a.traj):
        w, l = pl.make_windows(pl.normalize(rows[b], mn, mx), pl.normalize(mocap[b], mn_v, mx_v), 10)
        wins.append(w); labs.append(l)
    wins, labs = torch.cat(wins), torch.cat(labs)
    torch.manual_seed(1)
    model = RNN(60, a.hidden, a.layers, 24, dev).to(dev)
    criterion = torch.nn.MSELoss()
    optimizer = torch.optim.Adam(model.parameters(), lr=1e-3)
    n = wins.shape[0]
    losses = []
    for epoch in range(a.epochs):
        perm = torch.randperm(n, device=dev)
        for i in range(0, n, 4096):
            idx = perm[i:i + 4096]
            outputs = model(wins[idx])                                       # HIP forward (training path)
            target = torch.cat([labs[idx], (outputs[:, :12].detach() - labs[idx]).abs()], dim=1)   # gru_train.py:237-244
            loss = criterion(outputs, target)
            optimizer.zero_grad(); loss.backward(); optimizer.step()         # HIP backward, torch Adam
        losses.append(float(loss.item()))
        print(f"epoch {epoch + 1}: loss {losses[-1]:.6f}")

    # 3. evaluation on the first trajectory
    model.eval()
    # (the row stream itself: no window tensor, the first layer's input projection once per row -- os_gru_forward_windows)
    pred, above, below = pl.predict_rows(model, pl.normalize(rows[0], mn, mx), 10, mn_v, mx_v)
    truth = mocap[0][9:]                                         # window i <-> the label of row i + 9 (gru_train.py:186-192)
    mae = (pred - truth).abs().mean(dim=0)
    print("MAE per state:", np.round(mae.cpu().numpy(), 4))
    return losses, float(mae.mean().item())


if __name__ == "__main__":
    main()
