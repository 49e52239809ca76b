#!/bin/bash
# Development build of the library with in-kernel timestamps in kf_run_rows2_kernel (-DOS_ROWS_TS) and a run at BASELINE
# configs[1] (B = 4096, T = 1000): prints cycles per step and phase.  usage (GPU box): bash tools/rows_ts.sh
R=${GRAFT_REPO_ROOT:-/root/repo}; D=$R/build_ab/ts
bash $R/tools/ts_lib.sh ts liboptistate_ts.so kf_kernels,kf_rows_kernel -DOS_ROWS_TS > /dev/null || exit 1
cd $R
OPTISTATE_HIP_LIB=$D/liboptistate_ts.so python3 - <<'PY'
import ctypes as C, numpy as np, torch, sys
sys.path.insert(0, ".")
from optistate_amd import Engine
from optistate_amd.engine import _ptr
from optistate_amd.synth import synth_torch, Q_DEFAULT, R_DEFAULT
B, T = 4096, 1000
eng = Engine(0); eng.set_noise(Q_DEFAULT, R_DEFAULT)
d = synth_torch(B, T, "cuda", seed=404)
cp = eng.contact_soa_to_packed(d["contact"])
ts = torch.zeros(16, dtype=torch.int64, device="cuda")
for rep in range(2):
    x, P = d["x0"].clone(), d["P0"].clone()
    x_out = torch.empty((T, 12, B), device="cuda"); st = torch.empty((B,), dtype=torch.int32, device="cuda")
    rc = eng.lib.os_kf_run(eng._h, B, T, _ptr(d["p"]), _ptr(d["f"]), _ptr(d["dp"]), _ptr(d["imu"]), _ptr(cp), None, _ptr(x), _ptr(P),
                           _ptr(x_out), None, None, C.c_void_p(ts.data_ptr()), _ptr(st), 1 | 4, eng._stream())
    assert rc == 0, eng.lib.os_last_error(eng._h)
    torch.cuda.synchronize()
v = ts.cpu().numpy()[:8].astype(np.float64) / T
names = ["-", "wait + LDS reads + next DMA request", "attitude broadcast + rotations + odometry + torque/force sums", "covariance predict + own component of next_state",
         "optional outputs", "ten measurement updates", "(unused)", "(unused)"]
print("kernel", eng.kernel_name("kf"), " cycles per step by phase (lane 0 of workgroup 0, mean over", T, "steps; the x_out store and loop overhead are in phase 1 of the next step):")
for i in range(1, 6):
    print(f"  {names[i]:55s} {v[i]:8.0f}")
print(f"  {'sum':55s} {v[1:6].sum():8.0f}")
PY
