#!/usr/bin/env python3
"""Probe: S independent Engine.kf_mpc_run calls of B / S trajectories each on S streams against one call of B.  argv: B T S [S ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from optistate_amd import Engine
from optistate_amd.synth import synth_torch, Q_DEFAULT, R_DEFAULT

B = int(sys.argv[1]); T = int(sys.argv[2]); SS = [int(v) for v in sys.argv[3:]] or [1, 2, 4]
d = synth_torch(B, T, "cuda", seed=1)
ref = torch.zeros((T, 12, B), device="cuda"); ref[:, 5] = 0.28; ref[:, 9] = 0.1
tt = torch.arange(T, device="cuda")[:, None] * 0.01
ref[:, 0] = 0.02 * torch.sin(3 * tt); ref[:, 1] = 0.02 * torch.cos(2 * tt)
for S in SS:
    n = B // S
    engs = [Engine(0) for _ in range(S)]
    for e in engs: e.set_noise(Q_DEFAULT, R_DEFAULT)
    prio = os.environ.get("PRIO", "0") == "1"
    streams = [torch.cuda.Stream(priority=(-1 if (prio and i == 0) else 0)) for i in range(S)]
    sh = []
    for i in range(S):
        sl = slice(i * n, (i + 1) * n)
        sh.append({k: d[k][..., sl].contiguous() for k in ("p", "dp", "imu", "x0", "P0")})
        sh[-1]["c"] = engs[i].contact_soa_to_packed(d["contact"][..., sl].contiguous())
        sh[-1]["ref"] = ref[..., sl].contiguous()
    def run():
        out = []
        for i in range(S):
            with torch.cuda.stream(streams[i]):
                s = sh[i]
                out.append(engs[i].kf_mpc_run(s["p"], s["dp"], s["imu"], s["c"], s["ref"], s["x0"].clone(), s["P0"].clone(), want_iters=True))
        return out
    torch.cuda.synchronize(); run(); torch.cuda.synchronize()
    t0 = time.time(); r = run(); torch.cuda.synchronize(); dt = time.time() - t0
    it = torch.cat([o["iters"].float().flatten() for o in r])
    print(f"S={S} x {n}: {dt*1e3:.1f} ms -> {B*T/dt:.3e} steps/s ({dt/T*1e3:.3f} ms per step); iters mean {it.mean():.2f} max {int(it.max())}", flush=True)
