"""GPU: the single-step Kalman pieces of the C-ABI (os_kf_update batch / sequential) on RANDOM covariances -- well- and
ill-conditioned, exactly symmetric and slightly asymmetric (what rounding leaves behind, scaled up), non-diagonal R -- against the
reference's formulas in float64 numpy (kalman_filter.py:164-174: K = P H^T inv(S) with S as it is, P <- P - K H P).
    python tools/fuzz_pieces.py [n_cases] [seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from optistate_amd import Engine  # noqa: E402
from optistate_amd.synth import Q_DEFAULT, R_DEFAULT, R_FITTED  # noqa: E402

SEL = [0, 1, 2, 5, 6, 7, 8, 9, 10, 11]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    eng = Engine(0)
    bad = 0
    for case in range(n):
        B = int(rng.choice([1, 3, 16, 17, 64, 333, 1024, 5000]))
        sequential = bool(rng.integers(0, 2))
        kind = str(rng.choice(["default", "fitted", "full"]))
        if kind == "default":
            R = R_DEFAULT.copy()
        elif kind == "fitted":
            R = R_FITTED.copy()
        else:
            A = rng.normal(0, 0.1, (10, 10)); R = A @ A.T + np.diag(rng.uniform(1e-3, 1.0, 10)); sequential = False
        asym = 0.0 if sequential else float(rng.choice([0.0, 1e-7, 1e-5]))
        scale = 10.0 ** rng.uniform(-3, 1.5, (B, 12))
        A = rng.normal(0, 1, (B, 12, 12))
        P = np.einsum("bij,bkj->bik", A * scale[:, :, None], A * scale[:, :, None]) / 12 + np.eye(12) * 1e-6
        P = P * (1.0 + asym * rng.normal(0, 1, (B, 12, 12)))      # a not exactly symmetric P: every entry off by a relative asym
        P = P.astype(np.float32).astype(np.float64)
        x = rng.normal(0, 1, (B, 12)).astype(np.float32).astype(np.float64)
        z = rng.normal(0, 1, (B, 10)).astype(np.float32).astype(np.float64)
        eng.set_noise(Q_DEFAULT, R)
        R32 = np.asarray(R, np.float32).astype(np.float64)
        # the reference's update, float64
        S = P[:, SEL][:, :, SEL] + R32
        K = P[:, :, SEL] @ np.linalg.inv(S)
        xr = x + np.einsum("bia,ba->bi", K, z - x[:, SEL])
        Pr = P - K @ P[:, SEL, :]
        xt = torch.as_tensor(x.T.astype(np.float32).copy()).cuda()
        Pt = torch.as_tensor(P.reshape(B, 144).T.astype(np.float32).copy()).cuda()
        zt = torch.as_tensor(z.T.astype(np.float32).copy()).cuda()
        r = eng.kf_update(zt, xt, Pt, sequential=sequential, want_K=True)
        torch.cuda.synchronize()
        st = r["status"].cpu().numpy()
        xo = xt.cpu().numpy().T.astype(np.float64)
        Po = Pt.cpu().numpy().T.reshape(B, 12, 12).astype(np.float64)
        Ko = r["K"].cpu().numpy().T.reshape(B, 12, 10).astype(np.float64)
        cond = np.linalg.cond(S)
        tol = 1e-5 * np.maximum(1.0, cond / 1e6)[:, None]                 # float32 storage of P and x on both sides of the call
        e_x = np.abs(xo - xr).max(axis=1) / np.maximum(1.0, np.abs(xr).max(axis=1))
        e_P = np.abs(Po - Pr).max(axis=(1, 2)) / np.abs(P).max(axis=(1, 2))
        e_K = np.abs(Ko - K).max(axis=(1, 2)) / np.maximum(1.0, np.abs(K).max(axis=(1, 2)))
        good = (st & 15) == 0              # (a flagged trajectory -- non-positive pivot -- is a reported failure, not a wrong answer)
        e_x, e_P, e_K, tl = e_x[good], e_P[good], e_K[good], tol[good, 0] * 10
        ok = bool(good.mean() > 0.999 and (e_x < tl).all() and (e_P < tl).all() and (e_K < tl).all())
        print(f"case {case}: B={B} {'seq' if sequential else 'batch'} R={kind} asym={asym:g} cond(S) median {np.median(cond):.1e} max {cond.max():.1e}: "
              f"x {e_x.max():.1e} P {e_P.max():.1e} K {e_K.max():.1e} flagged {int((~good).sum())}" + ("" if ok else "   <-- ABOVE THE BAR"), flush=True)
        bad += 0 if ok else 1
    print(f"{n} cases, {bad} above the bars")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
