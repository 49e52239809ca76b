#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE in this container.

Run:  python tools/gen_golden.py        (needs /root/reference; never runs on the GPU box)

What is executed is the reference's own, unmodified code imported from /root/reference
(tools/ref_import.py explains the inert `casadi` stand-in that only lets the import succeed):
  Kalman_Filter.get_odom / set_measurements / predict / update / rotation_matrix_body_world
  (kalman_filter/kalman_filter.py), next_state (misc/force_controller.py), RNN (gru/gru_model.py).
The files written are DATA (inputs + the reference's outputs), float64 unless noted.

  kf_g1_odom.npz      G1  rotation / odometry / measurement unit vectors
  kf_g2_next_state.npz G2 next_state unit vectors incl. the int64-truncation edge cases
  kf_g3_traj.npz      G3  predict+update trajectories, T=200, two Q/R sets
  kf_g4_batch.npz     G4  32 independent trajectories x T=100 (batched-layout tests)
  kf_g7_feature.npz   G7  60-feature rows from G3 (order + rotated-p rule)
  kf_g8_mpc.npz       G8  estimate_state_mpc with externally supplied forces ("next" row f.1)
  gru_g5_small.npz    G5  RNN(60,64,1,24)  seed 1, weights + x (8,100,60) -> out
  gru_g5_ref.npz      G5  RNN(188,128,4,24) seed 1, weights + x (8,10,188) -> out
  gru_g6_train.npz    G6  one Adam step (loss, target, post-step fc.bias) with gru_train.py's loop body
  vit_g10_pos_embed.npz G10 the fixed 2-D sin-cos position table of the ViT encoder: transformer/pos_embed.py imported
                          unmodified (it needs only numpy/torch), called as transformer_model.py:66 does
  vit_g11_glue.npz    G11 "blocks = stand-in, glue = reference": the reference's own Transformer_Autoencoder
                          (transformer/transformer_model.py, imported unmodified) with tools/timm_standin.py providing the two
                          timm 0.3.2 classes it imports: forward_encoder latents of 8 seeded frames under seeded weights
                          (tests/golden_recipes.py rebuilds both from the seed), float32 as the reference runs and float64;
                          plus what initialize_weights (:54-82) leaves in a freshly constructed instance (seed 0):
                          position table, bias / LayerNorm constants, Xavier bounds, cls-token statistics
"""
import copy
import os
import pickle
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

from ref_import import import_reference, prime_forces          # noqa: E402
from optistate_amd.synth import synth_numpy, Q_DEFAULT, R_DEFAULT   # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)

KF_cls, next_state, fc, IP, RNN = import_reference()


def fresh_kf(x0, P0, Q, R):
    """A Kalman_Filter with private copies of every array (the reference aliases class-level
    constants, kalman_filter/kalman_filter.py:10,27-29 -- do not let runs pollute each other)."""
    kf = KF_cls()
    kf.x = np.array(x0, dtype=np.float64).reshape(12, 1).copy()
    kf.x_model = kf.x.copy()
    kf.P = np.array(P0, dtype=np.float64).copy()
    kf.Q = np.array(Q, dtype=np.float64).copy()
    kf.R = np.array(R, dtype=np.float64).copy()
    kf.z = np.zeros((10, 1))
    return kf


def col(v):
    return np.array(v, dtype=np.float64).reshape(-1, 1).copy()


def g1():
    rng = np.random.default_rng(101)
    n = 64
    th = rng.uniform(-1.0, 1.0, (n, 3))
    th[0] = 0.0
    th[1] = [np.pi / 2, 0, 0]
    th[2] = [0, 0, np.pi]
    p = rng.normal(0, 0.3, (n, 12)); dp = rng.normal(0, 0.2, (n, 12))
    imu = np.concatenate([th, rng.normal(0, 0.5, (n, 3))], axis=1)
    contact = np.zeros((n, 4), dtype=np.uint8)
    for i in range(n):
        k = 1 + (i % 3)                      # 1..3 stance legs (the cases the reference can run)
        contact[i, rng.permutation(4)[:k]] = 1
    kf = fresh_kf(IP.STARTING_STATE, IP.P, IP.Q, IP.R)
    R = np.zeros((n, 3, 3)); odom = np.zeros((n, 4)); z = np.zeros((n, 10))
    for i in range(n):
        R[i] = kf.rotation_matrix_body_world(th[i, 0], th[i, 1], th[i, 2])
        od = kf.get_odom(col(p[i]), col(dp[i]), contact[i].reshape(4, 1), col(imu[i]))
        kf.set_measurements(col(imu[i]), od)
        odom[i] = od.ravel(); z[i] = kf.z.ravel()
    np.savez(os.path.join(OUT, "kf_g1_odom.npz"), th=th, p=p, dp=dp, imu=imu, contact=contact, R=R, odom=odom, z=z)


def g2():
    rng = np.random.default_rng(202)
    specials = [[0, 0, 0]]
    for e in (1e-9, 1.1e-8, 1e-6, 1e-5, 2e-4, 3.4e-4, 5e-4):
        specials += [[e, 0, 0], [0, e, 0], [0, 0, e], [e, e, e], [-e, e, -e]]
    specials += [[0, 0, np.pi / 2], [0, 0, np.pi], [0, 0, -np.pi / 2], [np.pi, 0, 0], [0, np.pi, 0],
                 [np.pi / 2, 0, 0], [0, np.pi / 2, 0], [0, 0, float(np.float32(np.pi / 2))],
                 [0, 0, float(np.float32(np.pi))], [np.pi, np.pi, np.pi], [np.pi / 2, 0, np.pi / 2]]
    gen = rng.uniform(-1.2, 1.2, (40, 3))
    th = np.array(specials + gen.tolist())
    n = th.shape[0]
    x = rng.normal(0, 0.5, (n, 12)); x[:, 0:3] = th
    p = rng.normal(0, 0.3, (n, 12)); f = rng.normal(0, 10, (n, 12)) + np.tile([0, 0, 21.6], 4)
    xn = np.zeros((n, 12)); prot = np.zeros((n, 12)); Ablk = np.zeros((n, 3, 3))
    for i in range(n):
        pc = col(p[i])
        xn[i] = next_state(col(x[i]), pc, col(f[i]), 0.01).ravel()
        prot[i] = pc.ravel()                          # mutated in place by the reference
        Ablk[i] = fc.A[0:3, 6:9]
    np.savez(os.path.join(OUT, "kf_g2_next_state.npz"), x=x, p=p, f=f, dt=0.01, x_next=xn, p_rot=prot, A_block=Ablk)


def run_traj(d, b, Q, R, T):
    """One trajectory through the reference: get_odom -> set_measurements -> predict -> update."""
    kf = fresh_kf(d["x0"][b].astype(np.float64), d["P0"][b].astype(np.float64) if "P0f" not in d else d["P0f"], Q, R)
    kf.P = np.array(Q, dtype=np.float64).copy()       # P = copy(Q), Kalman_to_Training.py:144
    xs = np.zeros((T, 12)); xprior = np.zeros((T, 12)); prot = np.zeros((T, 12)); zs = np.zeros((T, 10))
    ptr = np.zeros(T); kg = np.zeros(T); Ks = {}
    for t in range(T):
        p = col(d["p"][b, t]); dp = col(d["dp"][b, t]); imu = col(d["imu"][b, t]); f = col(d["f"][b, t])
        c = d["contact"][b, t].reshape(4, 1)
        od = kf.get_odom(p, dp, c, imu)
        kf.set_measurements(imu, od)
        kf.predict(p, f)
        xprior[t] = kf.x_model.ravel()
        kf.update()
        xs[t] = kf.x.ravel(); prot[t] = p.ravel(); zs[t] = kf.z.ravel()
        ptr[t] = kf.P_trace; kg[t] = kf.K_gain
        if t in (0, 1, 50, T - 1):
            Ks[t] = kf.K.copy()
    return dict(x=xs, x_prior=xprior, p_rot=prot, z=zs, P_trace=ptr, K_gain=kg, P_final=kf.P.copy(), K=Ks)


def fitted_QR():
    with open("/root/reference/data_collection/trajectories/Q_R.pkl", "rb") as fh:
        Q, R = pickle.load(fh)
    R = R.copy()
    R[0, 0] = R[1, 1] = R[2, 2] = 0.0001              # Kalman_to_Training.py:141-143
    return Q.copy(), R


def g3_g7():
    T = 200
    d = synth_numpy(2, T, seed=303)
    Qf, Rf = fitted_QR()
    sets = [(Q_DEFAULT, R_DEFAULT), (Qf, Rf)]
    out = {k: d[k] for k in ("p", "f", "dp", "imu", "contact", "accel", "x0")}
    for s, (Q, R) in enumerate(sets):
        out[f"Q{s}"] = Q; out[f"R{s}"] = R
        for b in range(2):
            r = run_traj(d, b, Q, R, T)
            for k in ("x", "x_prior", "p_rot", "z", "P_trace", "K_gain", "P_final"):
                out[f"s{s}_b{b}_{k}"] = r[k]
            for t, K in r["K"].items():
                out[f"s{s}_b{b}_K{t}"] = K
    np.savez(os.path.join(OUT, "kf_g3_traj.npz"), **out)
    # G7: the 60-feature rows exactly as data_conversion_Kalman_to_Training.py:245-254 lays them out
    # (x_post | imu_list[6:12] | f | p AFTER next_state rotated it in place | dp | imu[0:6]).
    rows = np.zeros((T, 60))
    r = run_traj(d, 0, *sets[0], T)
    for t in range(T):
        rows[t] = np.concatenate([r["x"][t], d["accel"][0, t].astype(np.float64), d["f"][0, t].astype(np.float64),
                                  r["p_rot"][t], d["dp"][0, t].astype(np.float64), d["imu"][0, t].astype(np.float64)])
    mn, mx = rows.min(axis=0), rows.max(axis=0)                    # gru_train.py:59-62
    norm = (rows - mn) / (mx - mn)
    np.savez(os.path.join(OUT, "kf_g7_feature.npz"), rows=rows, min_vals=mn, max_vals=mx, normalized=norm)


def g4():
    B, T = 32, 100
    d = synth_numpy(B, T, seed=404)
    Qf, Rf = fitted_QR()
    out = {k: d[k] for k in ("p", "f", "dp", "imu", "contact", "accel", "x0")}
    for s, (Q, R) in enumerate([(Q_DEFAULT, R_DEFAULT), (Qf, Rf)]):
        X = np.zeros((B, T, 12)); PT = np.zeros((B, T)); KG = np.zeros((B, T)); PF = np.zeros((B, 12, 12))
        PR = np.zeros((B, T, 12))
        for b in range(B):
            r = run_traj(d, b, Q, R, T)
            X[b], PT[b], KG[b], PF[b], PR[b] = r["x"], r["P_trace"], r["K_gain"], r["P_final"], r["p_rot"]
        out.update({f"s{s}_x": X, f"s{s}_P_trace": PT, f"s{s}_K_gain": KG, f"s{s}_P_final": PF, f"s{s}_p_rot": PR,
                    f"Q{s}": Q, f"R{s}": R})
    np.savez(os.path.join(OUT, "kf_g4_batch.npz"), **out)


def g8():
    """estimate_state_mpc end-to-end as unmodified reference code, the QP 'solution' primed with
    externally supplied forces (column 0 is what predict_mpc uses, kalman_filter.py:152,161)."""
    T = 60
    d = synth_numpy(2, T, seed=808)
    rng = np.random.default_rng(809)
    body_ref = np.zeros((2, T, 12), dtype=np.float32)
    body_ref[..., 0:3] = (d["imu"][..., 0:3] + rng.normal(0, 0.01, (2, T, 3))).astype(np.float32)
    body_ref[..., 5] = 0.28
    out = {k: d[k] for k in ("p", "f", "dp", "imu", "contact", "x0")}
    out["body_ref"] = body_ref
    Qf, Rf = fitted_QR()
    for b in range(2):
        kf = fresh_kf(d["x0"][b], Qf, Qf, Rf)
        xs = np.zeros((T, 12)); ptr = np.zeros(T); prot = np.zeros((T, 12)); Fd_minmax = np.zeros((T, 2))
        for t in range(T):
            fm = np.zeros((12, 5)); fm[:, 0] = d["f"][b, t]
            prime_forces(fm)
            p = col(d["p"][b, t])
            x = kf.estimate_state_mpc(col(d["imu"][b, t]), p, col(d["dp"][b, t]), col(body_ref[b, t]),
                                      d["contact"][b, t].reshape(4, 1).astype(np.float64))
            xs[t] = x.ravel(); ptr[t] = kf.P_trace; prot[t] = p.ravel()
            Fd_minmax[t] = [kf.F_d.min(), kf.F_d.max()]
        out[f"b{b}_x"] = xs; out[f"b{b}_P_trace"] = ptr; out[f"b{b}_p_rot"] = prot; out[f"b{b}_P_final"] = kf.P.copy()
        out[f"b{b}_Fd_minmax"] = Fd_minmax
    out["Q"] = Qf; out["R"] = Rf
    np.savez(os.path.join(OUT, "kf_g8_mpc.npz"), **out)


def g5_g6():
    import torch
    torch.set_num_threads(1)
    dev = torch.device("cpu")
    for name, (I, H, L, Cc, B, T) in {"small": (60, 64, 1, 24, 8, 100), "ref": (188, 128, 4, 24, 8, 10)}.items():
        torch.manual_seed(1)
        m = RNN(I, H, L, Cc, dev)
        m.eval()
        g = torch.Generator().manual_seed(7)
        x = torch.rand(B, T, I, generator=g)
        with torch.no_grad():
            out = m(x)
            seq, hl = m.gru(x, torch.zeros(L, B, H))
        sd = {k: v.detach().numpy().copy() for k, v in m.state_dict().items()}
        np.savez(os.path.join(OUT, f"gru_g5_{name}.npz"), x=x.numpy(), out=out.numpy(), h_last=hl.numpy(),
                 seq_top=seq.numpy(), dims=np.array([I, H, L, Cc]), **{"w:" + k: v for k, v in sd.items()})
    # G6: one optimisation step with the loop body of gru/gru_train.py:232-249 (reference's RNN class,
    # nn.MSELoss, Adam lr 1e-4), small config so the fixture stays small.
    I, H, L, Cc, B, T = 60, 64, 1, 24, 16, 10
    torch.manual_seed(1)
    model = RNN(I, H, L, Cc, dev)
    criterion = torch.nn.MSELoss()
    optimizer = torch.optim.Adam(model.parameters(), lr=0.0001)
    g = torch.Generator().manual_seed(11)
    inputs = torch.rand(B, T, I, generator=g); labels = torch.rand(B, 12, generator=g)
    w0 = {k: v.detach().numpy().copy() for k, v in model.state_dict().items()}
    outputs = model(inputs)
    outputs_array = copy.deepcopy(outputs.cpu().detach().numpy())
    labels_array = labels.cpu().numpy()
    ground_truth_array = np.zeros((outputs_array.shape[0], 24))
    for l in range(outputs_array.shape[0]):
        error_array = np.abs(outputs_array[l, 0:12].reshape(12, 1) - labels_array[l, :].reshape(12, 1))
        ground_truth_array[l, 0:12] = labels_array[l, 0:12]
        ground_truth_array[l, 12:] = error_array.reshape(12,)
    ground_truth_tensor = torch.from_numpy(ground_truth_array).to(dev, dtype=torch.float32)
    loss = criterion(outputs, ground_truth_tensor)
    optimizer.zero_grad()
    loss.backward()
    grads = {k: p.grad.detach().numpy().copy() for k, p in model.named_parameters()}
    optimizer.step()
    w1 = {k: v.detach().numpy().copy() for k, v in model.state_dict().items()}
    np.savez(os.path.join(OUT, "gru_g6_train.npz"), inputs=inputs.numpy(), labels=labels.numpy(),
             outputs=outputs_array, target=ground_truth_array, loss=float(loss.item()),
             dims=np.array([I, H, L, Cc]), **{"w0:" + k: v for k, v in w0.items()},
             **{"w1:" + k: v for k, v in w1.items()}, **{"g:" + k: v for k, v in grads.items()})


def g9():
    """Convex-MPC force QP (misc/force_controller.py:70-162).  NOT reference-generated: casadi/qpOASES are absent, so this
    fixture holds float32-representable inputs and the KKT-certified solutions of oracle/mpc_oracle.py (parity unpinned)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import mpc_oracle as mo
    rng = np.random.default_rng(2024)
    pats = [(1, 0, 0, 1), (0, 1, 1, 0), (1, 1, 1, 1), (1, 1, 0, 1), (0, 0, 1, 0), (0, 0, 0, 0), (1, 0, 1, 1), (2, 1, 1, 0)]
    X, R, P, Cn, U = [], [], [], [], []
    for t in range(40):
        s = [0.3, 1.0, 3.0, 6.0][t % 4]
        x = np.array([0, 0, 0, 0, 0, 0.28, 0, 0, 0, 0, 0, 0.]) + s * rng.normal(0, [0.05] * 3 + [0.02] * 3 + [0.2] * 3 + [0.1] * 3)
        ref = np.array([0, 0, 0, 0, 0, 0.28, 0, 0, 0, 0.1, 0, 0.]) + s * rng.normal(0, [0.02] * 3 + [0.01] * 3 + [0.05] * 3 + [0.05] * 3)
        p = np.array([0.2, 0.1, -0.28, 0.2, -0.1, -0.28, -0.2, 0.1, -0.28, -0.2, -0.1, -0.28]) + rng.normal(0, 0.01, 12)
        c = np.array(pats[t % len(pats)], dtype=np.uint8)
        x, ref, p = (a.astype(np.float32).astype(np.float64) for a in (x, ref, p))
        f, u, info = mo.mpc_forces(x, ref, p, c)
        X.append(x); R.append(ref); P.append(p); Cn.append(c); U.append(u)
    np.savez_compressed(os.path.join(OUT, "mpc_g9_oracle.npz"), x=np.array(X), body_ref=np.array(R), p=np.array(P),
                        contact=np.array(Cn), u=np.array(U))


def g10():
    """ViT position table: the reference's own transformer/pos_embed.py (imports numpy + torch only, so it runs here although
    timm is absent), called exactly as Transformer_Autoencoder.initialize_weights does (transformer/transformer_model.py:64-67:
    get_2d_sincos_pos_embed(embed_dim, int(num_patches ** .5), cls_token=True)) for the encoder (128, 14x14) and, for a second
    shape, the decoder table (64)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_pos_embed", "/root/reference/transformer/pos_embed.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    enc = mod.get_2d_sincos_pos_embed(128, 14, cls_token=True)
    dec = mod.get_2d_sincos_pos_embed(64, 14, cls_token=True)
    small = mod.get_2d_sincos_pos_embed(32, 5, cls_token=True)
    np.savez_compressed(os.path.join(OUT, "vit_g10_pos_embed.npz"), enc_128_14=enc, dec_64_14=dec, small_32_5=small)


def g11():
    """The reference-owned glue of the optional ViT row (A10), pinned through the reference's own class."""
    import torch
    import timm_standin
    standin = timm_standin.install()
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from golden_recipes import g11_encoder_state, g11_frames, G11_SEED
    if "/root/reference" not in sys.path:
        sys.path.insert(0, "/root/reference")
    from transformer.transformer_model import Transformer_Autoencoder as RefViT
    # (a) initialisation as the reference leaves it
    torch.manual_seed(0)
    m = RefViT()
    sd0 = {k: v.detach().double().numpy() for k, v in m.state_dict().items()}
    enc_keys = [k for k in sd0 if not k.startswith("decoder")]
    lin_w = [k for k in enc_keys if k.endswith(".weight") and sd0[k].ndim == 2]
    init = dict(
        init_pos_embed=sd0["pos_embed"][0],
        init_bias_absmax=np.array([max(np.abs(sd0[k]).max() for k in enc_keys if k.endswith(".bias") and not k.startswith("patch_embed"))]),
        # _init_weights (:71-82) touches nn.Linear and nn.LayerNorm only: the Conv2d bias keeps torch's default U(-1/sqrt(fan_in), ..)
        init_patch_bias_absmax_over_bound=np.array([np.abs(sd0["patch_embed.proj.bias"]).max() * 16.0]),
        init_ln_weight_dev=np.array([max(np.abs(sd0[k] - 1).max() for k in enc_keys if "norm" in k and k.endswith("weight"))]),
        init_linear_absmax_over_bound=np.array([np.abs(sd0[k]).max() / np.sqrt(6.0 / sum(sd0[k].shape)) for k in lin_w]),
        init_linear_keys=np.array(lin_w),
        init_patch_absmax_over_bound=np.array([np.abs(sd0["patch_embed.proj.weight"]).max() / np.sqrt(6.0 / (256 + 128))]),
        init_cls_std=np.array([sd0["cls_token"].std()]),
        init_encoder_keys=np.array(sorted(enc_keys)),
        init_encoder_shapes=np.array([str(tuple(sd0[k].shape)) for k in sorted(enc_keys)]))
    # (b) forward_encoder under seeded weights, float32 (as the reference runs it) and float64
    w = g11_encoder_state()
    missing = m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=False)
    assert all(k.startswith("decoder") or k == "pos_embed" for k in missing.missing_keys), missing.missing_keys
    assert not missing.unexpected_keys
    fr = torch.from_numpy(g11_frames()).unsqueeze(1)
    m.eval()
    with torch.no_grad():
        lat32 = m.forward_encoder(fr).numpy()
        lat64 = m.double().forward_encoder(fr.double()).numpy()
    assert lat32.shape == (8, 1, 128)
    np.savez_compressed(os.path.join(OUT, "vit_g11_glue.npz"), latent_f32=lat32[:, 0], latent_f64=lat64[:, 0], seed=np.array([G11_SEED]),
                        blocks=np.array(["tools/timm_standin.py (timm 0.3.2 as published)" if standin else "timm " + __import__("timm").__version__]),
                        **init)
    print("G11: |f32 - f64| max", np.abs(lat32 - lat64).max(), "latent range", lat64.min(), lat64.max())


if __name__ == "__main__":
    if "--g9-only" in sys.argv:
        g9()
    elif "--g11-only" in sys.argv:
        g11()
    elif "--g10-only" in sys.argv:
        g10()
    else:
        g1(); g2(); g3_g7(); g4(); g8(); g5_g6(); g9(); g10(); g11()
    for fn in sorted(os.listdir(OUT)):
        print(f"{fn:28s} {os.path.getsize(os.path.join(OUT, fn)) / 1024:9.1f} KiB")
