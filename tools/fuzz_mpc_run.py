"""GPU: os_kf_mpc_run (estimate_state_mpc over B x T, kalman_filter.py:176-182) -- the persistent kernel against the per-step launch
sequence on random batch sizes (around the forms' crossovers), horizons, nominal and hostile inputs (every contact pattern), both
update forms; a few short trajectories of each case also against the two oracles (QP: mpc_oracle, filter step: the C oracle).  Round 6:
at 64 trajectories and more the launch sequence is also run in its plain form (no P_trace output) with the filter step inside the QP
launch and, at 8,192 and more, in two concurrent parts -- against OS_MPC_FUSE_KF=0: every output has to be IDENTICAL.
    python tools/fuzz_mpc_run.py [n_cases] [seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from optistate_amd import Engine  # noqa: E402
from optistate_amd.synth import synth_numpy, NOISE_SETS  # noqa: E402
from oracle import mpc_oracle as mo  # noqa: E402
from oracle import c_oracle as co  # noqa: E402


def oracle_run(d, Q, R, T, rows):
    xs = np.zeros((len(rows), T, 12)); fs = np.zeros((len(rows), T, 12))
    for i, b in enumerate(rows):
        x = d["x0"][b].astype(np.float64).copy(); P = Q.copy()
        for t in range(T):
            x32 = x.astype(np.float32).astype(np.float64)
            f, _, info = mo.mpc_forces(x32, d["body_ref"][b, t].astype(np.float64), d["p"][b, t].astype(np.float64), d["contact"][b, t],
                                       dt=float(np.float32(0.01)))
            f32 = f.astype(np.float32).astype(np.float64)
            r = co.kf_run_batch(d["p"][b:b + 1, t:t + 1], f32.reshape(1, 1, 12), d["dp"][b:b + 1, t:t + 1], d["imu"][b:b + 1, t:t + 1],
                                d["contact"][b:b + 1, t:t + 1], x.reshape(1, 12), P.reshape(1, 144), Q, R,
                                body_ref=d["body_ref"][b:b + 1, t:t + 1], mode=1)
            x = r["x_final"][0].copy(); P = r["P_final"][0].copy()
            xs[i, t] = x; fs[i, t] = f
    return xs, fs


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    mo.MASS = float(np.float32(8.8))
    mo.INERTIA = np.asarray(np.float32([0.05530364, 0.06011944, 0.10530434]), np.float64)
    engs = {}
    for mode in ("2", "0", "1"):             # forced wavefront-per-trajectory kernel / forced launch sequence / the library's own pick
        os.environ["OS_MPC_PERSISTENT"] = mode
        engs[mode] = Engine(0)
    bad = 0
    for case in range(n):
        B = int(rng.choice([1, 2, 7, 8, 63, 65, 256, 1000, 2047, 2049, 2560 + int(rng.integers(0, 3000)), 6144, 8192 + int(rng.integers(0, 3000)), 32768 + 16 * int(rng.integers(0, 40)) + int(rng.integers(0, 16))]))
        T = int(rng.choice([1, 2, 7, 20, 40]))
        if B > 1000:
            T = min(T, 7)
        if B > 10000:
            T = min(T, 3)
        hostile = bool(rng.integers(0, 2))
        if hostile:
            # the QP closes a loop around the filter: on hostile stretches (fast yaw, one or two stance legs) the closed loop is unstable
            # and a 1e-7 rounding difference grows ~3x per step (seen: 1e-6 at step 12 -> 1e-2 at step 26, both GPU forms together,
            # away from the float64 chain) -- keep hostile horizons short enough (8 steps) for a 1e-4 comparison to mean something
            T = min(T, 8)
        sequential = bool(rng.integers(0, 2))
        noise = str(rng.choice(["default", "fitted"]))
        Q, R = NOISE_SETS[noise]
        d = synth_numpy(B, T, seed=3000 + case, hostile=hostile)
        tt = np.arange(T) * 0.01
        ref = np.zeros((B, T, 12), np.float32)
        ref[:, :, 0] = 0.02 * np.sin(3 * tt); ref[:, :, 1] = 0.02 * np.cos(2 * tt); ref[:, :, 5] = 0.28; ref[:, :, 9] = 0.1
        ref += rng.normal(0, 0.005, ref.shape).astype(np.float32)
        d["body_ref"] = ref
        out = {}
        for mode, eng in engs.items():
            if mode == "1":
                continue
            eng.set_noise(Q, R)
            s = {k: eng.pack(torch.as_tensor(np.asarray(d[k], dtype=np.float32))) for k in ("p", "dp", "imu", "body_ref")}
            c = eng.pack_contact(torch.as_tensor(np.asarray(d["contact"])))
            x = torch.as_tensor(d["x0"].T.copy()).cuda()
            P = torch.as_tensor(np.tile(np.asarray(Q, dtype=np.float32).reshape(144, 1), (1, B))).cuda()
            r = eng.kf_mpc_run(s["p"], s["dp"], s["imu"], c, s["body_ref"], x, P, sequential=sequential, want_trace=True)
            torch.cuda.synchronize()
            out[mode] = (eng.unpack(r["x_out"]).cpu().numpy(), eng.unpack(r["f"]).cpu().numpy(), eng.failed(r["status"]).cpu().numpy().astype(bool),
                         eng.kernel_name("mpc"))
        # the plain launch sequence, filter step inside the QP launch (and two parts at >= 8,192) against the separate launches: identical
        same = True
        if B >= 64:
            eng = engs["0"]
            res = {}
            for fuse in ("0", "1"):
                os.environ["OS_MPC_FUSE_KF"] = fuse
                x = torch.as_tensor(d["x0"].T.copy()).cuda()
                P = torch.as_tensor(np.tile(np.asarray(Q, dtype=np.float32).reshape(144, 1), (1, B))).cuda()
                r = eng.kf_mpc_run(s["p"], s["dp"], s["imu"], c, s["body_ref"], x, P, sequential=sequential, want_iters=True, want_p_rot=True)
                torch.cuda.synchronize()
                res[fuse] = (r, x, P)
            os.environ.pop("OS_MPC_FUSE_KF")
            same = all(torch.equal(res["0"][0][k], res["1"][0][k]) for k in ("x_out", "f", "iters", "status", "p_rot")) and \
                torch.equal(res["0"][1], res["1"][1]) and torch.equal(res["0"][2], res["1"][2])
        # the library's own pick in its plain form: in the rows form's range (a 16-lane row per trajectory for all steps, where every step
        # qualifies) against the launch sequence: same iteration counts, the form-against-form bars
        rows_note = ""
        if B >= 64:
            eng = engs["1"]; eng.set_noise(Q, R)
            x = torch.as_tensor(d["x0"].T.copy()).cuda()
            P = torch.as_tensor(np.tile(np.asarray(Q, dtype=np.float32).reshape(144, 1), (1, B))).cuda()
            r = eng.kf_mpc_run(s["p"], s["dp"], s["imu"], c, s["body_ref"], x, P, sequential=sequential, want_iters=True, want_p_rot=True)
            torch.cuda.synchronize()
            if eng.kernel_name("mpc").startswith("kf_mpc_rows"):
                r0 = res["0"][0]
                e_rx = float((r["x_out"] - r0["x_out"]).abs().max()); e_rf = float((r["f"] - r0["f"]).abs().max())
                it_same = bool(torch.equal(r["iters"], r0["iters"])) and bool(torch.equal(r["status"], r0["status"]))
                same = same and e_rx < 1e-4 and e_rf < max(2e-2, 600.0 * e_rx) and it_same
                rows_note = f" | rows form vs sequence: x {e_rx:.1e} f {e_rf:.1e} N, iterations {'equal' if it_same else 'DIFFERENT'}"
        xp, fp, sp, kp = out["2"]; xs, fs, ss, ks = out["0"]
        good = ~sp & ~ss
        e_x = float(np.abs(xp[good] - xs[good]).max()) if good.any() else 0.0
        e_f = float(np.abs(fp[good] - fs[good]).max()) if good.any() else 0.0
        rows = [b for b in np.unique(np.r_[0, B // 2, B - 1]) if good[b]][:2]
        To = min(T, 8)
        xo, fo = oracle_run(d, Q, R, To, rows) if rows else (np.zeros((0, To, 12)), np.zeros((0, To, 12)))
        # EACH form against the float64 oracle chain, separately, with the FLAT bars (state 1e-4, forces 2e-2 N): ADVICE r5
        e_xo = float(max(np.abs(xp[rows][:, :To] - xo).max(), np.abs(xs[rows][:, :To] - xo).max())) if rows else 0.0
        e_fo = float(max(np.abs(fp[rows][:, :To] - fo).max(), np.abs(fs[rows][:, :To] - fo).max())) if rows else 0.0
        # forces: the QP's solution moves by ~100-500 N per unit of state near a face change, the states agree to ~3e-5: 2e-2 N (of up to
        # 150) -- and in proportion where the two forms' STATES are further apart (still inside their own 1e-4 bar): nominal inputs too are a
        # closed loop, a rounding difference can grow ~1.4x per step over a 20-step stretch before a face change resets it (seen: seed 142,
        # B = 1,000, T = 40: states 7e-5 apart at the worst step, forces 3.7e-2 N, both forms within 3e-2 N of the float64 chain)
        # (only THIS bar, form against form, scales with their state distance; restated after seed 142 of the round-5 sweep, recorded above)
        f_bar = max(2e-2, 600.0 * e_x)
        f_ok = e_f < f_bar
        self_note = ""
        if not f_ok and e_x < 1e-4 and good.any():
            # Round 6 (seed 11, B = 1,000, T = 40: an unstable stretch grows a 1e-7 rounding difference 1.4x per step from step 24 on; at
            # step 39 the forms are 8.9e-5 apart in state and 7.2e-2 N in force, 809 N per unit of state).  Rather than move the factor
            # again: where the forms' forces are furthest apart, EACH form's force must be the QP of ITS OWN state (the float64 oracle
            # solved at the state that form handed its solver) to the flat 2e-2 N -- the solver is right, the closed loop amplified.
            bw_f = int(np.argmax(np.where(good, np.abs(fp - fs).max(axis=(1, 2)), -1.0)))
            tw = int(np.argmax(np.abs(fp[bw_f] - fs[bw_f]).max(axis=1)))
            e_self = 0.0
            for xf, ff in ((xp, fp), (xs, fs)):
                x_prev = (d["x0"][bw_f] if tw == 0 else xf[bw_f, tw - 1]).astype(np.float32).astype(np.float64)
                f_or, _, _ = mo.mpc_forces(x_prev, d["body_ref"][bw_f, tw].astype(np.float64), d["p"][bw_f, tw].astype(np.float64), d["contact"][bw_f, tw],
                                           dt=float(np.float32(0.01)))
                e_self = max(e_self, float(np.abs(ff[bw_f, tw] - f_or).max()))
            f_ok = e_self < 2e-2
            self_note = f" | forces {e_f:.1e} N apart at trajectory {bw_f} step {tw}: each form against the QP of its own state {e_self:.1e} N"
        ok = e_x < 1e-4 and f_ok and e_xo < 1e-4 and e_fo < 2e-2 and int((sp != ss).sum()) == 0 and int(sp.sum()) <= 0.01 * B and same
        diag = ""
        if not ok and good.any():
            # which form left the oracle chain?  the trajectory where the two forms are furthest apart, over the whole horizon
            bw = int(np.argmax(np.where(good, np.abs(xp - xs).max(axis=(1, 2)), -1.0)))
            xo2, fo2 = oracle_run(d, Q, R, T, [bw])
            diag = (f"\n     trajectory {bw}: persistent vs oracle x {np.abs(xp[bw] - xo2[0]).max():.1e} f {np.abs(fp[bw] - fo2[0]).max():.1e} | "
                    f"sequence vs oracle x {np.abs(xs[bw] - xo2[0]).max():.1e} f {np.abs(fs[bw] - fo2[0]).max():.1e} | first step the forms differ by > 1e-5: "
                    f"{int(np.argmax(np.abs(xp[bw] - xs[bw]).max(axis=1) > 1e-5))}")
            ex = np.abs(xp[bw] - xo2[0]).max(axis=1); ef = np.abs(fp[bw] - fo2[0]).max(axis=1)
            diag += "\n     persistent vs oracle per step (x | f): " + " ".join(f"{t}:{ex[t]:.0e}|{ef[t]:.0e}" for t in range(T))
            diag += "\n     stance legs per step: " + " ".join(str(int(d["contact"][bw, t].sum())) for t in range(T))
        print(f"case {case}: B={B} T={T} hostile={hostile} {'seq' if sequential else 'batch'} noise={noise} [{kp} | {ks}] persistent vs sequence: x {e_x:.1e} f {e_f:.1e} N | "
              f"vs oracles ({len(rows)} trajectories x {To}): x {e_xo:.1e} f {e_fo:.1e} N | flagged {int(sp.sum())}/{int(ss.sum())}"
              + ("" if B < 64 else " | filter step inside the QP launch: identical" if same else " | filter step inside the QP launch or rows form: DIFFERENT") + rows_note + self_note
              + ("" if ok else "   <-- ABOVE THE BAR") + diag, flush=True)
        bad += 0 if ok else 1
    print(f"{n} cases, {bad} above the bars")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
