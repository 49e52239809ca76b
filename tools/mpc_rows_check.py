#!/usr/bin/env python3
"""kf_mpc_rows_kernel (a 16-lane row per trajectory for all T steps; OS_MPC_ROWS=lo:hi) against the launch sequence and the persistent kernel:
outputs to the bars of the form-against-form tests (state 1e-4, forces 5e-3 N; the record of a QP is built by the same source inlined
into another kernel: last-bit differences), times.  argv: B T [--force: the rows form whatever the batch size]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from optistate_amd import Engine
from optistate_amd.synth import synth_torch, Q_DEFAULT, R_DEFAULT

args = [v for v in sys.argv[1:] if not v.startswith("--")]
B = int(args[0]) if len(args) > 0 else 8192
T = int(args[1]) if len(args) > 1 else 20
d = synth_torch(B, T, "cuda", seed=1000)
ref = torch.zeros((T, 12, B), device="cuda"); ref[:, 5] = 0.28; ref[:, 9] = 0.1
out = {}
for name, env in (("sequence", {"OS_MPC_PERSISTENT": "0"}), ("persistent", {"OS_MPC_PERSISTENT": "2"}), ("rows", {"OS_MPC_ROWS": "64:100000000"} if "--force" in sys.argv else {})):
    for k in ("OS_MPC_PERSISTENT", "OS_MPC_ROWS"):
        os.environ.pop(k, None)
    os.environ.update(env)
    eng = Engine(0); eng.set_noise(Q_DEFAULT, R_DEFAULT)
    c = eng.contact_soa_to_packed(d["contact"])
    def run():
        x, P = d["x0"].clone(), d["P0"].clone()
        r = eng.kf_mpc_run(d["p"], d["dp"], d["imu"], c, ref, x, P, want_iters=True)
        torch.cuda.synchronize()
        return r, x, P
    run()
    t0 = time.time(); r, x, P = run(); dt = time.time() - t0
    out[name] = (r, x, P, dt, eng.kernel_name("mpc"))
    print(f"{name:10s} {dt*1e3:8.2f} ms  {B*T/dt:.3e} steps/s  [{eng.kernel_name('mpc')[:60]}]  iters mean {r['iters'].float().mean():.3f} status nonzero {int((r['status']!=0).sum())}", flush=True)
rs, xs, Ps = out["sequence"][:3]
for name in ("persistent", "rows"):
    r, x, P = out[name][:3]
    print(f"{name} vs sequence: x_out {float((r['x_out']-rs['x_out']).abs().max()):.2e}  f {float((r['f']-rs['f']).abs().max()):.2e} N  P {float((P-Ps).abs().max()):.2e}  "
          f"iters equal {bool(torch.equal(r['iters'], rs['iters']))} (differing {int((r['iters']!=rs['iters']).sum())})")
