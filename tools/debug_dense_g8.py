"""Development probe: kf_dense_rows_kernel on the G8 golden inputs, per-step error against the golden and the oracle."""
import sys
import numpy as np
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from optistate_amd import Engine
from oracle import c_oracle as orc
np.set_printoptions(linewidth=200, precision=3)
g = np.load("tests/golden/kf_g8_mpc.npz")
eng = Engine(0)
B, T = 2, 60
ref = orc.kf_run_batch(g["p"], g["f"], g["dp"], g["imu"], g["contact"], g["x0"], np.tile(g["Q"], (B, 1, 1)), g["Q"], g["R"], body_ref=g["body_ref"], mode=1)
print("oracle vs golden", max(np.abs(ref["x"][b] - g[f"b{b}_x"]).max() for b in range(2)))
s = {k: eng.pack(torch.as_tensor(np.asarray(g[k], dtype=np.float32))) for k in ("p", "f", "dp", "imu", "body_ref")}
c = eng.pack_contact(torch.as_tensor(np.asarray(g["contact"])))
eng.set_noise(g["Q"], g["R"])
for seq in (False, True):
    x = torch.as_tensor(np.asarray(g["x0"], dtype=np.float32).T.copy()).cuda()
    P = torch.as_tensor(np.tile(np.asarray(g["Q"], dtype=np.float32).reshape(144, 1), (1, B))).cuda()
    r = eng.kf_run(s["p"], s["f"], s["dp"], s["imu"], c, x, P, body_ref=s["body_ref"], dense_fd=True, sequential=seq, want_trace=bool(int(__import__("os").environ.get("WT", "1"))))
    xo = eng.unpack(r["x_out"]).cpu().numpy()
    for b in range(2):
        e = np.abs(xo[b] - g[f"b{b}_x"])
        print("seq" if seq else "batch", "b", b, "max", e.max(), "at step", e.max(axis=1).argmax(), "component", e.max(axis=0).argmax())
        print("  per-step max:", e.max(axis=1)[:12], "...", e.max(axis=1)[-4:])
        print("  per-component max:", e.max(axis=0))
        print("  status", r["status"].cpu().numpy())
