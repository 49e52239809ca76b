"""Seeded input recipes shared by tools/gen_golden.py (which runs the reference on them in the build container) and by the
tests (which rebuild the same inputs on the GPU box): numpy Generators are reproducible across machines, so the fixture only
has to store the reference's OUTPUTS."""
import numpy as np

G11_SEED, G11_FRAMES = 20261002, 8
VIT_DIMS = dict(img_size=224, patch_size=16, in_chans=1, embed_dim=128, depth=3, num_heads=4, mlp_hidden=512)


def g11_encoder_state(seed=G11_SEED):
    """Encoder weights with the reference's state_dict keys and shapes (transformer/transformer_model.py:20-29 with the
    constructor's defaults).  NOT an initialisation the reference would produce: every bias and LayerNorm parameter is
    non-trivial on purpose, so that a dropped bias, a swapped LayerNorm or a wrong position row shows up in the latent."""
    rng = np.random.default_rng(seed)
    D, Hm, depth, P = 128, 512, 3, 16
    u = lambda shape, fan_in, fan_out: rng.uniform(-1, 1, shape) * np.sqrt(6.0 / (fan_in + fan_out))
    sd = {"patch_embed.proj.weight": u((D, 1, P, P), P * P, D), "patch_embed.proj.bias": rng.normal(0, 0.05, D),
          "cls_token": rng.normal(0, 0.2, (1, 1, D))}
    for i in range(depth):
        b = f"blocks.{i}."
        sd[b + "norm1.weight"] = 1 + rng.normal(0, 0.1, D); sd[b + "norm1.bias"] = rng.normal(0, 0.1, D)
        sd[b + "attn.qkv.weight"] = 2.0 * u((3 * D, D), D, 3 * D); sd[b + "attn.qkv.bias"] = rng.normal(0, 0.1, 3 * D)
        sd[b + "attn.proj.weight"] = u((D, D), D, D); sd[b + "attn.proj.bias"] = rng.normal(0, 0.05, D)
        sd[b + "norm2.weight"] = 1 + rng.normal(0, 0.1, D); sd[b + "norm2.bias"] = rng.normal(0, 0.1, D)
        sd[b + "mlp.fc1.weight"] = u((Hm, D), D, Hm); sd[b + "mlp.fc1.bias"] = rng.normal(0, 0.1, Hm)
        sd[b + "mlp.fc2.weight"] = u((D, Hm), Hm, D); sd[b + "mlp.fc2.bias"] = rng.normal(0, 0.05, D)
    sd["norm.weight"] = 1 + rng.normal(0, 0.1, D); sd["norm.bias"] = rng.normal(0, 0.1, D)
    return {k: v.astype(np.float32) for k, v in sd.items()}


def g11_frames(seed=G11_SEED, n=G11_FRAMES):
    """Depth-like frames in [0, 1] (gru/gru_test.py:49-53): smooth blobs + noise, one all-zero and one all-one frame."""
    rng = np.random.default_rng(seed + 1)
    yy, xx = np.mgrid[0:224, 0:224] / 224.0
    fr = []
    for i in range(n):
        cx, cy, s = rng.uniform(0.2, 0.8), rng.uniform(0.2, 0.8), rng.uniform(0.05, 0.3)
        img = np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * s * s)) * rng.uniform(0.3, 1.0) + rng.uniform(0, 0.2, (224, 224))
        fr.append(np.clip(img, 0.0, 1.0))
    fr[n - 2][:] = 0.0
    fr[n - 1][:] = 1.0
    return np.stack(fr).astype(np.float32)


# ---- G13: a raw `.mat` log with the schema data_collection/data_conversion_raw_to_Kalman.py:43-57 reads ----------------------
G13_SEED, G13_ROWS, G13_CUTOFF, G13_END = 1313, 700, 430, 640


def _qmul(a, b):
    """Hamilton product of xyzw quaternions (arithmetic only: bit-reproducible on any machine)."""
    ax, ay, az, aw = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    bx, by, bz, bw = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz], axis=-1)


def g13_raw(seed=G13_SEED, N=G13_ROWS, drop_rows=(450,)):
    """Synthetic quadruped log, built from Generator draws, +, -, *, / and sqrt only (no libm): smooth random walks for the
    body, nominal stance + noise for the feet, lift patterns with 1..3 legs in the air (get_odom's runnable cases,
    kalman_filter/kalman_filter.py:83-98), a dropped mocap frame (all zeros: the script's quirk at :133-137)."""
    rng = np.random.default_rng(seed)
    walk = lambda shape, s: np.cumsum(rng.normal(0, s, shape), axis=0)
    t = np.cumsum(rng.uniform(0.009, 0.011, N)).reshape(1, N)
    th = walk((N, 3), 0.0015)                                      # T265 Euler angles, a few 1e-2 rad
    pos = walk((N, 3), 0.0008) + np.array([0.0, 0.0, 0.28]); pos[:, 0] += 0.001 * np.arange(N)
    t265 = np.concatenate([th, pos, rng.normal(0, 0.1, (N, 6))], axis=1)          # columns 6..11 are overwritten by the script
    # mocap: millimetres + xyzw quaternion, in a frame yawed and shifted against the T265's
    small = np.concatenate([0.5 * (th + rng.normal(0, 0.0005, (N, 3))), np.ones((N, 1))], axis=1)
    small /= np.sqrt((small * small).sum(1, keepdims=True))
    q0 = np.array([0.02, -0.01, 0.15, 1.0]); q0 /= np.sqrt((q0 * q0).sum())
    quat = _qmul(np.broadcast_to(q0, (N, 4)), small)
    mpos = (pos - np.array([0.0, 0.0, 0.28]) + rng.normal(0, 0.0002, (N, 3))) * 1000.0 + np.array([1200.0, -400.0, 310.0])
    mocap = np.concatenate([mpos, quat], axis=1)
    for r in drop_rows:
        mocap[r] = 0.0
    nominal = np.array([[0.2, 0.1, -0.28], [0.2, -0.1, -0.28], [-0.2, 0.1, -0.28], [-0.2, -0.1, -0.28]])
    lift_pats = np.array([[1, 0, 0, 1], [0, 1, 1, 0], [1, 0, 0, 0], [0, 1, 1, 1], [0, 0, 1, 0], [1, 1, 0, 1]], dtype=np.float64)
    lift = lift_pats[(np.arange(N) // 25) % len(lift_pats)]
    return dict(foot_state_history=nominal + rng.normal(0, 0.01, (N, 4, 3)), footSteps_ref=nominal + rng.normal(0, 0.005, (N, 4, 3)),
                bodyCM_ref=pos + rng.normal(0, 0.002, (N, 3)), bodyR_ref=th + rng.normal(0, 0.002, (N, 3)),
                control_history=rng.normal(0, 1, (N, 12)), liftLeg_ref=lift, body_state_history=t265, time_history=t,
                imu=rng.normal(0, 1, (N, 6)), encoder_history=walk((N, 4, 3), 0.004) + np.array([0.1, 0.8, -1.4]),
                depth4=rng.random((N, 4, 4)), mocap_history=mocap)


def g13_leg_jacobian(theta, which_leg=0):
    """Deterministic stand-in for scaler_kin's Leg.leg_jacobian_3DoF(theta, which_leg=j) (millimetres per radian;
    data_conversion_raw_to_Kalman.py:397): polynomial in the joint angles, arithmetic only."""
    a, b, c = (float(v) for v in np.asarray(theta, dtype=np.float64).reshape(3))
    s = 1.0 if which_leg in (0, 2) else -1.0
    return np.array([[120.0 * (1 - 0.5 * a * a), -40.0 * b, 15.0 * c + 3.0 * which_leg],
                     [s * 60.0 * a, 150.0 * (1 - 0.5 * b * b), -25.0 * c],
                     [10.0 * a * b, -90.0 * b + 5.0 * which_leg, 110.0 * (1 - 0.5 * c * c)]])
