#!/usr/bin/env python3
"""The reference's two caller loops, written out as a USER of the drop-in classes would have them after the one-line import switch
(INTEGRATION.md section 1), and timed -- the number a user who flips one import gets.

(a) gru/gru_train.py:232-249 -- `model(inputs)` -> the host-side target loop over the batch (numpy, one row at a time) ->
    `criterion(outputs, ground_truth_tensor)` -> `optimizer.zero_grad(); loss.backward(); optimizer.step()` with
    `torch.optim.Adam(model.parameters(), lr=1e-4)`, `RNN(188, 128, 4, 24)`, batch 64, a shuffling DataLoader over a TensorDataset.
    Only `RNN` comes from optistate_amd; everything else is torch / numpy exactly as the reference's script has it.  Beside it:
    `DataParallelTrainer.step` on the same batches (device-side target + fused Adam) and the same loop on torch's own CPU modules
    (= the reference itself on this host).
(b) data_collection/data_conversion_Kalman_to_Training.py:136-144,193-199 -- per trajectory a fresh `Kalman_Filter()`, the Q / R /
    P assignments, then `x = KF2.estimate_state_mpc(imu, p, dp, x_ref, contact_ref)` per time step, P_trace / K_gain appended:
    10 trajectories x 4,063 steps (settings.py:15-16: 4494 - 430 - 1).  Microseconds per step split into the QP call
    and the filter step (`_step` -> os_kf_step_mpc: both launches behind one call since round 6) and the Python around it.  Beside it: the same
    trajectories as ONE batched call (Engine.kf_mpc_run: the persistent kernel).

This file is the builder's caller code (a restatement of the loops' shape around the drop-in classes); no reference file travels.
    python tools/dropin_loops.py [--train-steps 300] [--traj 10] [--T 4063] [--out profiles/r06_dropin_loops.json]
"""
import argparse
import copy
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn as nn
from torch.utils.data import DataLoader, TensorDataset

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optistate_amd import RNN, Kalman_Filter, Engine                     # noqa: E402  (the import switch)
from optistate_amd.synth import synth_numpy, Q_FITTED, R_FITTED            # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--train-steps", type=int, default=300)
ap.add_argument("--traj", type=int, default=10)
ap.add_argument("--T", type=int, default=4063)
ap.add_argument("--out", default="")
a = ap.parse_args()
res = {}
device = torch.device("cuda")

# ----------------------------------------------------------------------------------------------------------------------------
# (a) the training loop of gru/gru_train.py:217-249
# ----------------------------------------------------------------------------------------------------------------------------
input_size, hidden_size, num_layers, num_outputs, batch_size, learning_rate = 188, 128, 4, 24, 64, 1e-4      # gru_train.py:30-37
n_windows = batch_size * 100
g = torch.Generator().manual_seed(1)
state_KF_tensor = torch.rand(n_windows, 10, input_size, generator=g)         # windows of 10 normalised rows (gru_train.py:186-192)
state_VICON_tensor = torch.rand(n_windows, 12, generator=g)
dataset = TensorDataset(state_KF_tensor, state_VICON_tensor)


def reference_loop(model, dev, steps, phases=None):
    """gru_train.py:232-249 as written there (the deepcopy, the per-row numpy loop and the .cpu() of the loss included)."""
    criterion = nn.MSELoss()
    optimizer = torch.optim.Adam(model.parameters(), lr=learning_rate)
    done, loss_list_training = 0, []
    t_start = None
    while done < steps + 10:
        train_loader = DataLoader(dataset, batch_size=batch_size, shuffle=True)
        for i, (inputs, labels) in enumerate(train_loader):
            if done == 10:                                   # ten warm-up steps, then the clock
                if dev.type == "cuda":
                    torch.cuda.synchronize()
                t_start = time.perf_counter()
            t0 = time.perf_counter()
            inputs = inputs.to(dev)
            labels = labels.to(dev)
            outputs = model(inputs)
            outputs_array = copy.deepcopy(outputs.cpu().detach().numpy())
            t1 = time.perf_counter()
            labels_array = labels.cpu().numpy()
            ground_truth_array = np.zeros((outputs_array.shape[0], 24))
            for l in range(outputs_array.shape[0]):
                error_array = np.abs(outputs_array[l, 0:12].reshape(12, 1) - labels_array[l, :].reshape(12, 1))
                ground_truth_array[l, 0:12] = labels_array[l, 0:12]
                ground_truth_array[l, 12:] = error_array.reshape(12,)
            ground_truth_tensor = torch.from_numpy(ground_truth_array).to(dev, dtype=torch.float32)
            loss = criterion(outputs, ground_truth_tensor)
            t2 = time.perf_counter()
            optimizer.zero_grad()
            loss.backward()
            optimizer.step()
            loss_list_training.append(loss.cpu().detach().numpy())
            t3 = time.perf_counter()
            if phases is not None and done >= 10:
                phases["to_device_forward_and_readback"] += t1 - t0
                phases["host_target_loop_and_loss"] += t2 - t1
                phases["backward_adam_and_loss_readback"] += t3 - t2
            done += 1
            if done >= steps + 10:
                break
    return (time.perf_counter() - t_start) / steps * 1e3, float(loss_list_training[-1])


torch.manual_seed(1)
model = RNN(input_size, hidden_size, num_layers, num_outputs, device).to(device)
ph = {"to_device_forward_and_readback": 0.0, "host_target_loop_and_loss": 0.0, "backward_adam_and_loss_readback": 0.0}
ms, last = reference_loop(model, device, a.train_steps, ph)
res["train_loop_verbatim_on_dropin_RNN"] = {
    "ms_per_step": ms, "last_loss": last, "steps": a.train_steps, "batch": batch_size,
    "ms_per_phase": {k: v / a.train_steps * 1e3 for k, v in ph.items()},
    "what": "gru/gru_train.py:232-249 as written (DataLoader, model(inputs), deepcopy + per-row numpy target loop, MSELoss, "
            "loss.backward(), torch.optim.Adam.step(), loss.cpu()) with only `RNN` imported from optistate_amd"}
ph = res["train_loop_verbatim_on_dropin_RNN"]["ms_per_phase"]
res["train_loop_verbatim_on_dropin_RNN"]["largest_python_side_term"] = max(ph, key=ph.get)

# the same batches through the library's own step (device-side target, fused Adam, no readback)
from optistate_amd.train import DataParallelTrainer                       # noqa: E402
torch.manual_seed(1)
model2 = RNN(input_size, hidden_size, num_layers, num_outputs, device).to(device)
tr = DataParallelTrainer(model2, lr=learning_rate)
xs, ys = state_KF_tensor.to(device), state_VICON_tensor.to(device)
for i in range(10):
    tr.step(xs[i * 64:(i + 1) * 64], ys[i * 64:(i + 1) * 64])
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(a.train_steps):
    j = (i % 100) * 64
    tr.step(xs[j:j + 64], ys[j:j + 64])
torch.cuda.synchronize()
res["train_DataParallelTrainer_step"] = {"ms_per_step": (time.perf_counter() - t0) / a.train_steps * 1e3,
                                         "what": "DataParallelTrainer.step on device-resident batches of 64 (target + MSE gradient on the device, fused Adam)"}

# the reference itself on this host: torch's own GRU / Linear on the CPU through the same loop
class _CpuRNN(nn.Module):                                                 # gru/gru_model.py:7-49 in torch (a library, not reference code)
    def __init__(self):
        super().__init__()
        self.gru = nn.GRU(input_size, hidden_size, num_layers, batch_first=True); self.fc = nn.Linear(hidden_size, num_outputs)

    def forward(self, x):
        return torch.sigmoid(self.fc(self.gru(x)[0][:, -1, :]))


torch.set_num_threads(min(8, os.cpu_count() or 1))     # the build container's eight cores (18.9 ms there); all 128 of the GPU box oversubscribe a batch of 64: 760 ms
torch.manual_seed(1)
ms_cpu, _ = reference_loop(_CpuRNN(), torch.device("cpu"), min(a.train_steps, 60))
res["train_loop_torch_cpu"] = {"ms_per_step": ms_cpu, "threads": torch.get_num_threads(),
                               "what": "the same loop on torch.nn.GRU + Linear + sigmoid on the host cores (the reference's own arithmetic)"}

# ----------------------------------------------------------------------------------------------------------------------------
# (b) the estimate_state_mpc caller loop of data_conversion_Kalman_to_Training.py:136-144,193-199
# ----------------------------------------------------------------------------------------------------------------------------
NT, T = a.traj, a.T
d = synth_numpy(NT, T, seed=5)
tt = np.arange(T) * 0.01
ref_all = np.zeros((NT, T, 12)); ref_all[:, :, 5] = 0.28; ref_all[:, :, 9] = 0.1
ref_all[:, :, 0] = 0.02 * np.sin(3 * tt); ref_all[:, :, 1] = 0.02 * np.cos(2 * tt)
Q, R = Q_FITTED.copy(), R_FITTED.copy()

timers = {"solve_mpc": 0.0, "_step": 0.0}
for name in list(timers):
    orig = getattr(Kalman_Filter, name)

    def wrap(self, *args, __orig=orig, __name=name, **kw):
        t0 = time.perf_counter()
        try:
            return __orig(self, *args, **kw)
        finally:
            timers[__name] += time.perf_counter() - t0
    setattr(Kalman_Filter, name, wrap)

t_loop = 0.0
for n in range(NT + 1):                                      # trajectory 0 runs twice: the first pass warms up context and code objects
    k = max(n - 1, 0)
    p_list_est, dp_list, imu_list = d["p"][k].astype(np.float64), d["dp"][k].astype(np.float64), d["imu"][k].astype(np.float64)
    contact_list, ref_list = d["contact"][k], ref_all[k]
    traj_length = T if n else min(T, 200)
    KF2 = Kalman_Filter()
    x_start = d["x0"][k].astype(np.float64).reshape(12, 1)
    KF2.x[:] = x_start
    KF2.Q = Q
    KF2.R = R
    KF2.R[0, 0] = 0.0001
    KF2.R[1, 1] = 0.0001
    KF2.R[2, 2] = 0.0001
    KF2.P = copy.deepcopy(Q)
    p_trace, K_gain = [], []
    if n == 1:
        timers = {kk: 0.0 for kk in timers}; t_loop = 0.0
    t0 = time.perf_counter()
    for i in range(0, traj_length):
        p = p_list_est[i].reshape(12, 1)
        dp = dp_list[i].reshape(12, 1)
        imu = imu_list[i][0:6].reshape(6, 1)
        contact_ref = contact_list[i].reshape(4, 1)
        x_ref = ref_list[i].reshape(12, 1)
        x = KF2.estimate_state_mpc(imu, p, dp, x_ref, contact_ref)
        p_trace.append(KF2.P_trace)
        K_gain.append(KF2.K_gain)
    t_loop += time.perf_counter() - t0
steps = NT * T
res["estimate_state_mpc_loop_on_dropin_Kalman_Filter"] = {
    "us_per_step": t_loop / steps * 1e6, "trajectories": NT, "steps_each": T,
    "us_per_step_split": {"os_kf_step_mpc_call": timers["_step"] / steps * 1e6,
                          "python_around_it": (t_loop - timers["solve_mpc"] - timers["_step"]) / steps * 1e6},
    "finite": bool(np.isfinite(x).all()),
    "what": "data_conversion_Kalman_to_Training.py:136-144,193-199: a fresh Kalman_Filter per trajectory, estimate_state_mpc per step "
            "(ONE os_kf_step_mpc call: the QP launch + the float64 step kernel + one stream synchronise), sequentially over the trajectories",
    "round_5_form": "separate solve_mpc call through torch tensors + os_kf_step: 194 us per step (157 in the QP call)"}
# the explicit two-call form a caller may still use (kf.solve_mpc(...) then predict_mpc(f=...)): its QP call alone
kfq = Kalman_Filter()
pq, rq, cq = d["p"][0, 0].astype(np.float64).reshape(12, 1), ref_all[0, 0].reshape(12, 1), d["contact"][0, 0].reshape(4, 1)
for _ in range(20):
    kfq.solve_mpc(pq, rq, cq)
t0 = time.perf_counter()
for _ in range(200):
    kfq.solve_mpc(pq, rq, cq)
res["estimate_state_mpc_loop_on_dropin_Kalman_Filter"]["separate_solve_mpc_call_us"] = (time.perf_counter() - t0) / 200 * 1e6
sp = res["estimate_state_mpc_loop_on_dropin_Kalman_Filter"]["us_per_step_split"]
res["estimate_state_mpc_loop_on_dropin_Kalman_Filter"]["largest_term"] = max(sp, key=sp.get)

# the same trajectories as ONE batched call (what replaces the loop: INTEGRATION.md section 2)
eng = Engine(0)
eng.set_noise(Q, R)
sd = {k: torch.as_tensor(np.ascontiguousarray(d[k].transpose(1, 2, 0)), dtype=torch.float32, device=device) for k in ("p", "dp", "imu")}
cp = eng.contact_soa_to_packed(torch.as_tensor(np.ascontiguousarray(d["contact"].transpose(1, 2, 0)), device=device))
reft = torch.as_tensor(np.ascontiguousarray(ref_all.transpose(1, 2, 0)), dtype=torch.float32, device=device)
x0 = torch.as_tensor(d["x0"].T.copy(), dtype=torch.float32, device=device)
P0 = torch.as_tensor(np.tile(Q.reshape(144, 1), (1, NT)), dtype=torch.float32, device=device)


def batched():
    xx, PP = x0.clone(), P0.clone()
    return eng.kf_mpc_run(sd["p"], sd["dp"], sd["imu"], cp, reft, xx, PP)


batched(); torch.cuda.synchronize(); t0 = time.perf_counter()
batched(); torch.cuda.synchronize()
el = time.perf_counter() - t0
res["estimate_state_mpc_batched_kf_mpc_run"] = {"us_per_step_of_a_trajectory": el / T * 1e6, "us_per_trajectory_step": el / steps * 1e6,
                                                "what": f"Engine.kf_mpc_run over the same {NT} x {T} (one persistent kernel, float64 filter + QP)"}
print(json.dumps(res, indent=1))
if a.out:
    with open(a.out, "w") as fh:
        json.dump(res, fh, indent=1)
