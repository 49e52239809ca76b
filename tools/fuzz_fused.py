"""GPU: random-shape parity sweep of os_fused_run (Kalman -> 60-feature row -> min-max -> GRU [-> latent stream]) against the
float64 C oracle chain: hidden 64 / 128 / 32, 1-4 layers, latent streams, single-kernel / two-kernel paths, both noise sets,
nominal and hostile inputs.     python tools/fuzz_fused.py [n_cases] [seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from optistate_amd import Engine, RNN, flatten_state_dict  # noqa: E402
from optistate_amd.synth import synth_numpy, NOISE_SETS  # noqa: E402
from oracle import c_oracle as orc  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    eng = Engine(0)
    bad = 0
    for case in range(n):
        H = int(rng.choice([64, 64, 128, 32]))
        L = int(rng.integers(1, 5))
        NL = int(rng.choice([0, 0, 0, 4, 128, 132]))
        B = int(rng.choice([1, 2, 31, 64, 65, 200, 333, 1000, 2100, 21000, 33000]))
        T = int(rng.choice([1, 2, 5, 12, 30]))
        if B > 5000:
            T = min(T, 5)
        noise = str(rng.choice(["default", "fitted"]))
        hostile = bool(rng.integers(0, 2))
        two = rng.choice([None, None, True, False])
        two = None if two is None else bool(two)
        if two is False and not (H == 64 and L >= 1 and NL == 0):
            two = None
        Q, R = NOISE_SETS[noise]
        d = synth_numpy(B, T, seed=2000 + case, hostile=hostile)
        ref = orc.kf_run_batch(d["p"], d["f"], d["dp"], d["imu"], d["contact"], d["x0"], np.tile(Q, (B, 1, 1)), Q, R)
        rows = np.concatenate([ref["x"], d["accel"].astype(np.float64), d["f"].astype(np.float64), ref["p_rot"],
                               d["dp"].astype(np.float64), d["imu"].astype(np.float64)], axis=2)
        mn, mx = rows.reshape(-1, 60).min(0), rows.reshape(-1, 60).max(0)
        mx = np.where(mx - mn < 1e-6, mn + 1.0, mx)
        norm = (rows - mn) / (mx - mn)
        lat = rng.random((B, T, NL)).astype(np.float32) if NL else None
        gin = norm if lat is None else np.concatenate([norm, lat.astype(np.float64)], axis=2)
        torch.manual_seed(case)
        m = RNN(60 + NL, H, L, 24, torch.device("cpu"))
        pick = np.unique(np.r_[0:min(B, 40), max(B - 40, 0):B])
        ref_out, _, _ = orc.gru_forward(gin[pick], orc.flatten_state_dict(m.state_dict(), L), 60 + NL, H, L, 24)
        eng.set_noise(Q, R)
        eng.load_gru(flatten_state_dict(m.state_dict(), L), 60 + NL, H, L, 24)
        s = {k: eng.pack(torch.as_tensor(np.asarray(d[k], dtype=np.float32))) for k in ("p", "f", "dp", "imu", "accel")}
        c = eng.pack_contact(torch.as_tensor(np.asarray(d["contact"])))
        x = torch.as_tensor(d["x0"].T.copy()).cuda()
        P = torch.as_tensor(np.tile(np.asarray(Q, dtype=np.float32).reshape(144, 1), (1, B))).cuda()
        mm = torch.as_tensor(np.stack([mn, mx]).astype(np.float32)).cuda()
        latent = None if lat is None else torch.as_tensor(lat).permute(1, 2, 0).contiguous().cuda()      # [T][NL][B]
        try:
            r = eng.fused_run(s["p"], s["f"], s["dp"], s["imu"], c, s["accel"], mm, x, P, two_kernel=two, latent=latent)
        except RuntimeError as e:
            if "(-4)" in str(e):
                print(f"case {case}: H={H} L={L} NL={NL} B={B} T={T} two_kernel={two}: shape not taken (-4): {str(e)[:90]}")
                continue
            raise
        torch.cuda.synchronize()
        okrows = ref["status"] == 0
        fail = eng.failed(r["status"]).cpu().numpy().astype(bool)
        both = okrows & ~fail
        xo = eng.unpack(r["x_out"]).cpu().numpy()
        e_x = float(np.abs(xo[both] - ref["x"][both]).max())
        e_o = float(np.abs(r["out"].cpu().numpy()[pick][both[pick]] - ref_out[both[pick]]).max())
        ok = e_x < 1e-4 and e_o < 1e-4 and int((fail != ~okrows).sum()) <= 0.001 * B + 1
        print(f"case {case}: H={H} L={L} NL={NL} B={B} T={T} noise={noise} hostile={hostile} two_kernel={two} "
              f"[{eng.kernel_name('fused') or '-'} / {eng.kernel_name('kf') or '-'} / {eng.kernel_name('gru_layer') or '-'}] state {e_x:.1e} gru {e_o:.1e} "
              f"status-mismatch {int((fail != ~okrows).sum())}" + ("" if ok else "   <-- ABOVE THE BAR"), flush=True)
        bad += 0 if ok else 1
    print(f"{n} cases, {bad} above the bars")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
