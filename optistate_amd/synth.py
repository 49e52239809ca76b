"""Synthetic input streams shaped like the reference's `.mat`-derived pipeline.

The reference's sample trajectories are not in the repository (Google-Drive link, reference
README.md:24), so tests and bench use seeded synthetic streams with the schema that
`data_collection/data_conversion_raw_to_Kalman.py:443-447` produces and
`data_collection/data_conversion_Kalman_to_Training.py:193-199,245-254` consumes:
per step p (12, leg-major xyz foot positions, body frame), f (12, ground-reaction forces),
dp (12, foot velocities), imu (6: Euler angles, angular rates), contact (4, 0/1),
accel (6: the `imu_list[i][6:12]` columns that only enter the feature row).
Distributions follow SURVEY.md section 8(d).
"""
import numpy as np

NOMINAL_P = np.array([0.2, 0.1, -0.28, 0.2, -0.1, -0.28, -0.2, 0.1, -0.28, -0.2, -0.1, -0.28])
MASS, G = 8.8, 9.81
X0 = np.array([0, 0, 0, 0, 0, 0.28, 0, 0, 0, 0, 0, 0], dtype=np.float64)       # settings.py:25
Q_DEFAULT = np.diag([0.01, 0.01, 0.01, 0.01, 0.0001, 0.01, 0.01, 0.01, 0.01, 0.01, 0.01, 0.0001])  # settings.py:28
R_DEFAULT = np.diag([0.01] * 10)                                                # settings.py:30
# data_collection/trajectories/Q_R.pkl (the reference's fitted set; values copied as data), with
# R[0:3] forced to 1e-4 as data_collection/data_conversion_Kalman_to_Training.py:139-144 does.
Q_FITTED = np.diag([8.00759796e-05, 8.78192928e-05, 1.84720027e-02, 1.08296176e-05, 1.46891961e-05,
                    6.07544836e-06, 2.10808463e-01, 2.03398616e-01, 1.15387378e+02, 3.67406764e-02,
                    5.26113437e-02, 2.12345892e-02])
R_FITTED = np.diag([1e-4, 1e-4, 1e-4, 7.81921020e-02, 2.42114010e-01, 2.60983908e-01, 5.75257015e+01,
                    7.05195918e-02, 1.10483579e-01, 3.73801504e-02])


NOISE_SETS = {"default": (Q_DEFAULT, R_DEFAULT), "fitted": (Q_FITTED, R_FITTED)}

# ---- "hostile" streams: everything the nominal distributions never exercise (contact patterns, unwrapped yaw) ----
# contacts: random 4-bit words held for 16-step segments (0, 1, 2, 3 or 4 stance legs: standing and flight included, a
# quarter of the trajectories stand on all four legs for the whole first half); attitude: yaw = yaw0 + rate t with |rate| up
# to 8 rad/s (passes +-pi and keeps going: the Cody-Waite branch of the kernels' sincos, every quadrant), roll / pitch
# sinusoids up to 1 rad; the first 5/16 of the batch start at exact attitudes (0, and theta_z / theta_y = float32(pi/2), +-pi)
# with the IMU agreeing at t = 0, so the int64-truncation predicate (misc/force_controller.py:248-251,271) sees entries of R
# at or next to +-1.  The pitch = pi/2 block starts with roll 0.3 and yaw -0.4 by default: with roll = yaw = 0 there (gimbal lock:
# R depends on yaw - roll only) the float64 R[1][1] = cz cx + sz sy sx lands within one rounding of 1.0 for several steps, and
# whether the reference's int64 A picks up a 1 then depends on the last bit of its own float64 state -- a coin toss no float32
# filter can reproduce (measured: one trajectory in 65,536 x 100 steps flipped, theta_y off by 5e-3).  gimbal_lock=True puts the
# block AT roll = yaw = 0: the kernels report those decisions in status bit 4 (include/optistate_hip.h), and
# tests/test_gpu_fullsize.py::test_gimbal_lock_knife_edge_is_flagged checks that nothing else exceeds the state bar.
HOSTILE_SEG = 16


def _hostile_exact_starts(B, gimbal_lock=False):
    """[(first, last, (thx, thy, thz))]: index ranges of the batch that start at exact attitudes."""
    n = max(1, B // 16)
    pi2, pi = float(np.float32(np.pi / 2)), float(np.float32(np.pi))
    return [(0, n, (0.0, 0.0, 0.0)), (n, 2 * n, (0.0, 0.0, pi2)), (2 * n, 3 * n, (0.0, 0.0, pi)),
            (3 * n, 4 * n, (0.0, 0.0, -pi)), (4 * n, 5 * n, (0.0, pi2, 0.0) if gimbal_lock else (0.3, pi2, -0.4))]


def synth_numpy(B, T, seed=0, theta0_noise=True, dtype=np.float32, hostile=False):
    """Returns dict of [B][T][field] arrays (float32-representable) + x0 [B][12], P0 [B][12][12]."""
    if hostile:
        import torch
        d = synth_torch(B, T, "cpu", seed=seed, soa=False, hostile=True)
        out = {k: v.numpy().astype(dtype) for k, v in d.items() if k not in ("contact", "P0")}
        out["contact"] = d["contact"].numpy().astype(np.uint8)
        out["P0"] = np.tile(Q_DEFAULT, (B, 1, 1)).astype(dtype)
        return out
    rng = np.random.default_rng(seed)
    t = np.arange(T)[None, :, None] * 0.01
    p = NOMINAL_P[None, None, :] + rng.normal(0, 0.01, (B, T, 12))
    f = np.tile(np.array([0, 0, MASS * G / 4]), 4)[None, None, :] + rng.normal(0, 3.0, (B, T, 12))
    dp = rng.normal(0, 0.1, (B, T, 12))
    amp = rng.uniform(0.05, 0.1, (B, 1, 6))
    frq = rng.uniform(1.0, 3.0, (B, 1, 6))
    ph = rng.uniform(0, 2 * np.pi, (B, 1, 6))
    imu = amp * np.sin(frq * t + ph) + rng.normal(0, 0.005, (B, T, 6))
    # trot: diagonal pairs alternate every 25 steps (always 2 stance legs)
    phase = ((np.arange(T)[None, :] // 25) + rng.integers(0, 2, (B, 1))) % 2
    contact = np.zeros((B, T, 4), dtype=np.uint8)
    contact[..., 0] = phase == 0
    contact[..., 3] = phase == 0
    contact[..., 1] = phase == 1
    contact[..., 2] = phase == 1
    accel = rng.normal(0, 1.0, (B, T, 6))
    x0 = np.tile(X0, (B, 1))
    if theta0_noise:
        half = B // 2
        x0[half:, 0:3] += rng.normal(0, 0.02, (B - half, 3))
    P0 = np.tile(Q_DEFAULT, (B, 1, 1))
    out = dict(p=p, f=f, dp=dp, imu=imu, accel=accel, x0=x0, P0=P0)
    out = {k: v.astype(dtype) for k, v in out.items()}
    out["contact"] = contact
    return out


def synth_torch(B, T, device, seed=0, soa=True, hostile=False, gimbal_lock=False):
    """Same distributions generated on `device` with torch, directly in the kernel's SoA layout
    [T][field][B] (float32) when soa=True.  Used by bench.py at BASELINE sizes.  hostile=True: see HOSTILE_SEG above."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)

    def rn(*shape, std=1.0):
        return torch.randn(*shape, device=device, generator=g, dtype=torch.float32) * std

    def ru(*shape, lo=0.0, hi=1.0):
        return torch.rand(*shape, device=device, generator=g, dtype=torch.float32) * (hi - lo) + lo

    nom = torch.tensor(NOMINAL_P, dtype=torch.float32, device=device)
    fz = torch.tensor(np.tile(np.array([0, 0, MASS * G / 4]), 4), dtype=torch.float32, device=device)
    p = nom[None, :, None] + rn(T, 12, B, std=0.01)
    f = fz[None, :, None] + rn(T, 12, B, std=3.0)
    dp = rn(T, 12, B, std=0.1)
    tt = torch.arange(T, device=device, dtype=torch.float32)[:, None, None] * 0.01
    imu = ru(1, 6, B, lo=0.05, hi=0.1) * torch.sin(ru(1, 6, B, lo=1.0, hi=3.0) * tt + ru(1, 6, B, hi=6.2831853)) \
        + rn(T, 6, B, std=0.005)
    phase = ((torch.arange(T, device=device)[:, None] // 25) + torch.randint(0, 2, (1, B), device=device, generator=g)) % 2
    contact = torch.zeros(T, 4, B, dtype=torch.uint8, device=device)
    contact[:, 0] = phase == 0
    contact[:, 3] = phase == 0
    contact[:, 1] = phase == 1
    contact[:, 2] = phase == 1
    accel = rn(T, 6, B)
    x0 = torch.tensor(X0, dtype=torch.float32, device=device)[:, None].repeat(1, B).contiguous()
    if hostile:
        nseg = (T + HOSTILE_SEG - 1) // HOSTILE_SEG + 1
        word = torch.randint(0, 16, (nseg, B), device=device, generator=g)
        kind = ru(nseg, B)
        word = torch.where(kind < 0.2, torch.full_like(word, 15), word)             # standing
        word = torch.where(kind > 0.92, torch.zeros_like(word), word)               # flight
        off = torch.randint(0, HOSTILE_SEG, (1, B), device=device, generator=g)
        seg = (torch.arange(T, device=device)[:, None] + off) // HOSTILE_SEG        # [T][B]
        wt = torch.gather(word, 0, seg)
        wt[: T // 2, (B // 2): (B // 2 + B // 4)] = 15                               # long all-stance stretch
        contact = torch.stack([(wt >> l) & 1 for l in range(4)], dim=1).to(torch.uint8).contiguous()
        rate = ru(1, 1, B, lo=-8.0, hi=8.0)
        yaw = ru(1, 1, B, lo=-3.0, hi=3.0) + rate * tt + 0.05 * torch.sin(ru(1, 1, B, lo=1.0, hi=3.0) * tt)
        rp = ru(1, 2, B, lo=0.3, hi=1.0) * torch.sin(ru(1, 2, B, lo=1.0, hi=3.0) * tt + ru(1, 2, B, hi=6.2831853))
        imu = torch.cat([rp, yaw, imu[:, 3:6] + torch.cat([torch.zeros(1, 2, B, device=device), rate], dim=1)], dim=1) \
            + torch.cat([rn(T, 3, B, std=0.005), torch.zeros(T, 3, B, device=device)], dim=1)
        x0[0:3] = imu[0, 0:3]                                                        # the filter starts on the first IMU attitude
        for lo, hi, th in _hostile_exact_starts(B, gimbal_lock):
            for c in range(3):
                x0[c, lo:hi] = th[c]
                imu[0, c, lo:hi] = th[c]
    else:
        x0[0:3, B // 2:] += rn(3, B - B // 2, std=0.02)
    P0 = torch.tensor(Q_DEFAULT, dtype=torch.float32, device=device).reshape(144, 1).repeat(1, B).contiguous()
    d = dict(p=p, f=f, dp=dp, imu=imu, contact=contact, accel=accel, x0=x0, P0=P0)
    if not soa:
        d = {k: (v.permute(2, 0, 1).contiguous() if v.dim() == 3 else v.t().contiguous()) for k, v in d.items()}
    return d
