#!/usr/bin/env python3
"""A/B of fused-kernel variants in ONE process, interleaved rounds (cdna_hip_programming.md rule 24): the round-1 kernel
(OS_FUSED_V1=1: h tile in LDS) against the current one, same inputs; prints per-variant kernel ms (library HIP events),
the difference between their outputs and both against the float64 oracle on a sample.  Development aid."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from optistate_amd import Engine, RNN, flatten_state_dict
from optistate_amd.synth import synth_torch, Q_DEFAULT, R_DEFAULT

B, T = int(os.environ.get("AB_B", 65536)), int(os.environ.get("AB_T", 100))
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
variants, kw = {}, {}
for name, env, kws in (("v2", {}, {}), ("bf3", {}, {"split_bf16": True}),
                       ("bf2", {}, {"split_bf16": 2})):
    for e in ("OS_FUSED_V1", "OS_BF16_TERMS"):
        os.environ.pop(e, None)
    os.environ.update(env)
    variants[name] = Engine(0)
    kw[name] = kws
for e in ("OS_FUSED_V1", "OS_BF16_TERMS"):
    os.environ.pop(e, None)
d = synth_torch(B, T, "cuda", seed=1)
c = Engine.contact_soa_to_packed(d["contact"])
torch.manual_seed(0)
m = RNN(60, 64, 1, 24, torch.device("cpu"))
flat = flatten_state_dict(m.state_dict(), 1, "cuda")
mm = torch.stack([torch.full((60,), -30.0), torch.full((60,), 30.0)]).cuda()
res, ms = {}, {k: [] for k in variants}
for e in variants.values():
    e.set_noise(Q_DEFAULT, R_DEFAULT); e.load_gru(flat, 60, 64, 1, 24); e.profile(True)
for r in range(rounds + 1):
    for name, e in variants.items():
        x = d["x0"].clone(); P = d["P0"].clone()
        res[name] = e.fused_run(d["p"], d["f"], d["dp"], d["imu"], c, d["accel"], mm, x, P, two_kernel=False, **kw[name])
        torch.cuda.synchronize()
        pr = e.profile_read()
        if r:
            ms[name].append(pr["fused"][0] / max(pr["fused"][1], 1))
for name in variants:
    v = np.array(ms[name])
    print(f"{name}: kernel {variants[name].kernel_name('fused')}: median {np.median(v):.4f} ms  min {v.min():.4f}  max {v.max():.4f}  "
          f"-> {47616 * B * T / np.median(v) / 1e9:.1f} TFLOP/s = {47616 * B * T / np.median(v) / 1e9 / 157.3:.3f} of fp32 MFMA peak")
for name in variants:
    if name != "v2":
        print("%s vs v2: state max abs diff %.3e, head max abs diff %.3e" % (name, float((res[name]["x_out"] - res["v2"]["x_out"]).abs().max()),
                                                                            float((res[name]["out"] - res["v2"]["out"]).abs().max())))
from oracle import c_oracle as orc
idx = torch.randperm(B, generator=torch.Generator().manual_seed(3))[:128].cuda()
g = lambda k: d[k][:, :, idx].permute(2, 0, 1).double().cpu().numpy()
ref = orc.kf_run_batch(g("p"), g("f"), g("dp"), g("imu"), d["contact"][:, :, idx].permute(2, 0, 1).cpu().numpy(),
                       d["x0"][:, idx].t().double().cpu().numpy(), np.tile(Q_DEFAULT, (128, 1, 1)), Q_DEFAULT, R_DEFAULT)
rows = np.concatenate([ref["x"], g("accel"), g("f"), ref["p_rot"], g("dp"), g("imu")], axis=2)
ro, _, _ = orc.gru_forward((rows + 30.0) / 60.0, orc.flatten_state_dict(m.state_dict(), 1), 60, 64, 1, 24)
for name in variants:
    xo = res[name]["x_out"][:, :, idx].permute(2, 0, 1).cpu().numpy()
    print(f"{name} vs oracle: state linf {np.abs(xo - ref['x']).max():.3e}  head linf {np.abs(res[name]['out'][idx].cpu().numpy() - ro).max():.3e}  "
          f"status nonzero {int((res[name]['status'] != 0).sum())}")
