"""GPU: random-shape / random-variant parity sweep of os_kf_run (every kernel family, both update forms, predict(p,f) and the
predict_mpc covariance, nominal and hostile inputs, both noise sets) against the float64 C oracle.
    python tools/fuzz_kf.py [n_cases] [seed]
Exits non-zero at the first state distance above the suite's bar (1e-4; P / traces 1e-3 relative)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from optistate_amd import Engine  # noqa: E402
from optistate_amd.synth import synth_numpy, NOISE_SETS  # noqa: E402
from oracle import c_oracle as orc  # noqa: E402

VARIANTS = {"batch": dict(sequential=False, symmetric=False), "seq-lanes": dict(sequential=True, symmetric=False, lane_per_trajectory=True),
            "sym-lanes": dict(sequential=True, symmetric=True, lane_per_trajectory=True), "default": dict(),
            "dense-batch": dict(dense_fd=True, sequential=False), "dense-seq": dict(dense_fd=True, sequential=True)}


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    eng = Engine(0)
    bad = 0
    for case in range(n):
        B = int(rng.choice([1, 2, 3, 15, 16, 17, 63, 64, 65, 255, 257, 1000, 4096, 8191, 8193, 9000]))
        T = int(rng.choice([1, 2, 5, 17, 40, 100]))
        vname = str(rng.choice(list(VARIANTS)))
        noise = str(rng.choice(["default", "fitted"]))
        hostile = bool(rng.integers(0, 2))
        aux = bool(rng.integers(0, 2))
        Q, R = NOISE_SETS[noise]
        d = synth_numpy(B, T, seed=1000 + case, hostile=hostile)
        kw = dict(VARIANTS[vname])
        dense = kw.get("dense_fd", False)
        br = None
        if dense:
            d["body_ref"] = np.zeros((B, T, 12), dtype=np.float32)
            d["body_ref"][..., 0:3] = d["imu"][..., 0:3] + rng.normal(0, 0.01, (B, T, 3)).astype(np.float32)
            br = eng.pack(torch.as_tensor(d["body_ref"]))
        ref = orc.kf_run_batch(d["p"], d["f"], d["dp"], d["imu"], d["contact"], d["x0"], np.tile(Q, (B, 1, 1)), Q, R,
                               body_ref=d.get("body_ref"), mode=1 if dense else 0)
        s = {k: eng.pack(torch.as_tensor(np.asarray(d[k], dtype=np.float32))) for k in ("p", "f", "dp", "imu")}
        c = eng.pack_contact(torch.as_tensor(np.asarray(d["contact"])))
        eng.set_noise(Q, R)
        x = torch.as_tensor(d["x0"].T.copy()).cuda()
        P = torch.as_tensor(np.tile(np.asarray(Q, dtype=np.float32).reshape(144, 1), (1, B))).cuda()
        r = eng.kf_run(s["p"], s["f"], s["dp"], s["imu"], c, x, P, body_ref=br, want_p_rot=aux, want_trace=aux, want_gain=aux, **kw)
        torch.cuda.synchronize()
        okrows = ref["status"] == 0                       # (the oracle flags what the reference would raise on)
        fail = eng.failed(r["status"]).cpu().numpy().astype(bool)
        xo = eng.unpack(r["x_out"]).cpu().numpy()
        both = okrows & ~fail
        e_x = float(np.abs(xo[both] - ref["x"][both]).max()) if both.any() else 0.0
        Pf = P.cpu().numpy().T.reshape(B, 12, 12)
        e_P = float((np.abs(Pf[both] - ref["P_final"][both]).max() / np.abs(ref["P_final"][both]).max())) if both.any() else 0.0
        msg = f"case {case}: B={B} T={T} {vname} noise={noise} hostile={hostile} aux={aux} [{eng.kernel_name('kf')}] x {e_x:.1e} P {e_P:.1e}"
        ok = e_x < 1e-4 and e_P < 1e-3 and int((fail != ~okrows).sum()) <= 0.001 * B + 1
        if aux and both.any():
            e_t = float(np.abs(r["P_trace"].cpu().numpy().T[both] / ref["P_trace"][both] - 1).max())
            e_r = float(np.abs(eng.unpack(r["p_rot"]).cpu().numpy()[both] - ref["p_rot"][both]).max())
            msg += f" trace {e_t:.1e} p_rot {e_r:.1e}"
            ok = ok and e_t < 2e-3 and e_r < 1e-4
        print(msg + f" status-mismatch {int((fail != ~okrows).sum())}" + ("" if ok else "   <-- ABOVE THE BAR"), flush=True)
        bad += 0 if ok else 1
    print(f"{n} cases, {bad} above the bars")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
