"""GPU: regression tests for the round-1 advisor findings (shared per-device engine state, split predict_mpc -> update in
float32, silent symmetric-storage assumptions, kernel names in reports) and for per-trajectory noise."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _ref_out(m, x):
    from oracle import c_oracle as orc
    out, _, _ = orc.gru_forward(x.cpu().numpy(), orc.flatten_state_dict(m.state_dict(), m.num_layers), m.input_size,
                                m.hidden_size, m.num_layers, m.num_classes)
    return out


def test_two_rnn_instances_on_one_gpu_do_not_share_weights():
    """gru_train.py:205-217 trains num_models networks; two modules on one device share the per-device engine, whose single
    loaded model must follow whoever calls forward."""
    from optistate_amd import RNN
    torch.manual_seed(0)
    a = RNN(60, 64, 1, 24, torch.device("cuda")).to("cuda")
    b = RNN(60, 64, 1, 24, torch.device("cuda")).to("cuda")
    c = RNN(188, 128, 2, 24, torch.device("cuda")).to("cuda")          # different shape on the same engine
    xa, xc = torch.rand(9, 10, 60).cuda(), torch.rand(5, 10, 188).cuda()
    with torch.no_grad():
        for _ in range(2):                                             # interleave: A, B, C, A, B, C
            oa, ob, oc = a(xa), b(xa), c(xc)
            assert np.abs(oa.cpu().numpy() - _ref_out(a, xa)).max() < 1e-5
            assert np.abs(ob.cpu().numpy() - _ref_out(b, xa)).max() < 1e-5
            assert np.abs(oc.cpu().numpy() - _ref_out(c, xc)).max() < 1e-5
    assert (oa - ob).abs().max().item() > 1e-3                         # the two models really differ


def test_two_forwards_before_backward_and_an_eval_forward_in_between():
    """loss(model(a)) + loss(model(b)) with a no-grad forward of ANOTHER model in between: every graph keeps its own saved
    activations and weights (they used to live in one engine-global buffer)."""
    from optistate_amd import RNN
    from test_gpu_train import torch_reference_grads
    dims = (60, 64, 2, 24)
    torch.manual_seed(4)
    m = RNN(*dims, torch.device("cuda")).to("cuda")
    other = RNN(60, 32, 1, 24, torch.device("cuda")).to("cuda")
    xa, xb = torch.rand(40, 6, 60), torch.rand(23, 6, 60)              # different batch sizes on purpose
    ya, yb = torch.rand(40, 12), torch.rand(23, 12)
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    _, _, la, ga, _ = torch_reference_grads(sd, dims, xa, ya)
    _, _, lb, gb, _ = torch_reference_grads(sd, dims, xb, yb)

    def loss_of(x, y):
        out = m(x.cuda())
        tgt = torch.cat([y.cuda(), (out[:, :12].detach() - y.cuda()).abs()], dim=1)
        return torch.nn.functional.mse_loss(out, tgt)
    l1 = loss_of(xa, ya)
    with torch.no_grad():
        other(torch.rand(7, 6, 60).cuda())                             # evicts m's weights from the shared engine
    l2 = loss_of(xb, yb)
    (l1 + l2).backward()
    assert abs(l1.item() - la) < 1e-6 and abs(l2.item() - lb) < 1e-6
    for k, p in m.named_parameters():
        r = ga[k] + gb[k]
        scale = max(r.abs().max().item(), 1e-8)
        assert (p.grad.cpu().double() - r).abs().max().item() < 2e-4 * scale + 1e-9, k


def test_trainer_step_then_model_forward_uses_the_updated_weights():
    """DataParallelTrainer's fused Adam writes the flat bucket in place (no torch version bump): an evaluation forward
    between steps (gru_train.py:253-261) must run on the post-step weights, GRU and head alike."""
    from optistate_amd import RNN
    from optistate_amd.train import DataParallelTrainer
    torch.manual_seed(1)
    m = RNN(60, 64, 1, 24, torch.device("cuda")).to("cuda")
    tr = DataParallelTrainer(m, lr=1e-2)                               # large step so stale weights are obvious
    x, y = torch.rand(64, 10, 60).cuda(), torch.rand(64, 12).cuda()
    with torch.no_grad():
        before = m(x).clone()
    for _ in range(3):
        tr.step(x, y)
    with torch.no_grad():
        after = m(x)
    assert np.abs(after.cpu().numpy() - _ref_out(m, x)).max() < 1e-5    # state_dict (views of the bucket) == what ran
    assert (after - before).abs().max().item() > 1e-3


def test_split_predict_mpc_then_update_meets_the_bar_g8():
    """The sequence estimate_state_mpc is made of, called piecewise on the drop-in class (get_odom, set_measurements,
    predict_mpc with logged forces, update): float64 throughout (os_kf_step), so the split sequence meets the reference to
    rounding level although predict_mpc's element-wise exp(dt F) leaves an ill-conditioned covariance between the calls."""
    from optistate_amd import Kalman_Filter
    g = load_golden("kf_g8_mpc.npz")
    for b in range(2):
        kf = Kalman_Filter()
        kf.x[:] = g["x0"][b].reshape(12, 1)
        kf.Q = g["Q"].copy(); kf.R = g["R"].copy(); kf.P = g["Q"].copy()
        for t in range(30):
            p = g["p"][b, t].astype(np.float64).reshape(12, 1)
            imu = g["imu"][b, t].reshape(6, 1)
            odom = kf.get_odom(p, g["dp"][b, t].reshape(12, 1), g["contact"][b, t].reshape(4, 1), imu)
            kf.set_measurements(imu, odom)
            kf.predict_mpc(p, g["body_ref"][b, t].reshape(12, 1), g["contact"][b, t].reshape(4, 1), f=g["f"][b, t])
            assert kf.P.dtype == np.float64
            kf.update()
            assert np.abs(kf.x.ravel() - g[f"b{b}_x"][t]).max() < 1e-7, (b, t)
            assert np.abs(p.ravel() - g[f"b{b}_p_rot"][t]).max() < 1e-8
            assert abs(kf.P_trace / g[f"b{b}_P_trace"][t] - 1) < 1e-7
        if b == 0:
            assert kf.K.shape == (12, 10)


def test_asymmetric_p0_is_flagged_by_the_symmetric_storage_kernel():
    from optistate_amd import Engine
    from optistate_amd.synth import synth_numpy, Q_DEFAULT, R_DEFAULT
    eng = Engine(0)
    eng.set_noise(Q_DEFAULT, R_DEFAULT)
    B, T = 12288, 3                                                    # above the small-batch kernel's range
    d = synth_numpy(B, T, seed=5)
    s = {k: eng.pack(torch.as_tensor(d[k])) for k in ("p", "f", "dp", "imu")}
    c = eng.pack_contact(torch.as_tensor(d["contact"]))
    x = torch.as_tensor(d["x0"].T.copy()).cuda()
    P = torch.as_tensor(np.tile(Q_DEFAULT.astype(np.float32).reshape(144, 1), (1, B))).cuda()
    P[1 * 12 + 7, 100] = 3e-3                                          # P[1][7] != P[7][1] for trajectory 100 only
    r = eng.kf_run(s["p"], s["f"], s["dp"], s["imu"], c, x.clone(), P.clone())
    assert eng.kernel_name("kf") == "kf_run_sym_kernel"
    st = r["status"].cpu().numpy()
    assert st[100] & 8 and (np.delete(st, 100) == 0).all()
    r2 = eng.kf_run(s["p"], s["f"], s["dp"], s["imu"], c, x.clone(), P.clone(), symmetric=False)     # the reference's full P
    assert eng.kernel_name("kf") == "kf_dense_rows_kernel<SEQ,predict(p,f)>" and (r2["status"].cpu().numpy() == 0).all()
    # a non-symmetric Q switches the default to the full-P kernels
    Qa = Q_DEFAULT.copy(); Qa[0, 5] = 1e-3
    eng.set_noise(Qa, R_DEFAULT)
    eng.kf_run(s["p"], s["f"], s["dp"], s["imu"], c, x.clone(), P.clone())
    assert eng.kernel_name("kf") == "kf_dense_rows_kernel<SEQ,predict(p,f)>"
    eng.set_noise(Q_DEFAULT, R_DEFAULT)


def test_kernel_name_reports_the_variant_that_ran():
    from optistate_amd import Engine
    from optistate_amd.synth import synth_numpy, Q_DEFAULT, R_DEFAULT
    eng = Engine(0)
    eng.set_noise(Q_DEFAULT, R_DEFAULT)
    d = synth_numpy(256, 2, seed=1)
    s = {k: eng.pack(torch.as_tensor(d[k])) for k in ("p", "f", "dp", "imu")}
    c = eng.pack_contact(torch.as_tensor(d["contact"]))
    x = torch.as_tensor(d["x0"].T.copy()).cuda()
    P = torch.as_tensor(np.tile(Q_DEFAULT.astype(np.float32).reshape(144, 1), (1, 256))).cuda()
    eng.profile(True)
    eng.kf_run(s["p"], s["f"], s["dp"], s["imu"], c, x.clone(), P.clone())
    assert eng.kernel_name("kf") == "kf_run_rows2_kernel"               # B = 256 <= 8,192: 16 lanes per trajectory
    eng.kf_run(s["p"], s["f"], s["dp"], s["imu"], c, x.clone(), P.clone(), sequential=False)
    assert eng.kernel_name("kf") == "kf_dense_rows_kernel<BATCH,predict(p,f)>"      # the batch form: float64, 16 lanes per trajectory
    pr = eng.profile_read()
    assert pr["kf"][1] == 2 and pr["kf"][0] > 0 and set(pr) == set(__import__("optistate_amd._capi", fromlist=["x"]).PHASE_NAMES)
    eng.profile(False)


@pytest.mark.parametrize("B,trace", [(192, True), (20000, True), (192, False), (8200, False)])
def test_per_trajectory_noise_matches_the_oracle_looped_per_filter(B, trace):
    """os_kf_run_noise: every trajectory carries its own diagonal Q and R (each reference filter instance does:
    data_conversion_Kalman_to_Training.py:138-144); the oracle is run once per noise set."""
    from optistate_amd import Engine
    from optistate_amd.synth import synth_numpy, Q_DEFAULT, R_DEFAULT, Q_FITTED, R_FITTED
    from oracle import c_oracle as orc
    eng = Engine(0)
    eng.set_noise(Q_DEFAULT, R_DEFAULT)                                # context-wide values must NOT be used
    T = 40
    d = synth_numpy(B, T, seed=21)
    rng = np.random.default_rng(0)
    sets = [(Q_DEFAULT, R_DEFAULT), (Q_FITTED, R_FITTED), (Q_DEFAULT * 3.0, R_DEFAULT * 0.2)]
    which = rng.integers(0, 3, B)
    qd = np.stack([np.diag(sets[w][0]) for w in which]).astype(np.float32)     # [B][12]
    rd = np.stack([np.diag(sets[w][1]) for w in which]).astype(np.float32)
    P0 = np.stack([np.diag(q) for q in qd]).astype(np.float32)                 # P = Q.copy() per filter (:144)
    s = {k: eng.pack(torch.as_tensor(d[k])) for k in ("p", "f", "dp", "imu")}
    c = eng.pack_contact(torch.as_tensor(d["contact"]))
    x = torch.as_tensor(d["x0"].T.copy()).cuda()
    P = torch.as_tensor(P0.reshape(B, 144).T.copy()).cuda()
    r = eng.kf_run_noise(s["p"], s["f"], s["dp"], s["imu"], c, x, P, torch.as_tensor(qd.T.copy()).cuda(),
                         torch.as_tensor(rd.T.copy()).cuda(), want_trace=trace)       # (trace / no trace: two instantiations)
    assert eng.kernel_name("kf") == "kf_run_sym_noise_kernel"
    assert (r["status"].cpu().numpy() == 0).all()
    xo = eng.unpack(r["x_out"]).cpu().numpy()
    sample = np.arange(B) if B <= 256 else rng.choice(B, 96, replace=False)
    for w in range(3):
        idx = sample[which[sample] == w]
        if idx.size == 0:
            continue
        Qw, Rw = np.diag(np.diag(sets[w][0]).astype(np.float32)).astype(np.float64), np.diag(np.diag(sets[w][1]).astype(np.float32)).astype(np.float64)
        ref = orc.kf_run_batch(d["p"][idx], d["f"][idx], d["dp"][idx], d["imu"][idx], d["contact"][idx], d["x0"][idx],
                               np.tile(Qw, (idx.size, 1, 1)), Qw, Rw)
        assert np.abs(xo[idx] - ref["x"]).max() < 1e-4
        if trace:
            ptr = r["P_trace"].cpu().numpy()[:, idx].T
            assert np.abs(ptr / ref["P_trace"] - 1).max() < 1e-3


# ------------------------------------------------------------------------------------------------------------------
# round-2 advisor findings
# ------------------------------------------------------------------------------------------------------------------
def test_non_finite_values_in_unselected_leg_entries_never_reach_the_measurement():
    """get_odom only reads dp_x, dp_y, p_z of STANCE legs and dp_z of SWING legs (kalman_filter.py:83-90 `if contact_cur[i]
    == 1 / == 0`): a NaN / Inf in the entries it skips must not reach z (a 0/1 weight would: 0 * NaN = NaN).  Through
    os_kf_odom and through every batched filter kernel family (the helper is shared)."""
    from optistate_amd import Engine
    from optistate_amd.engine import _ptr
    from optistate_amd.synth import synth_numpy, Q_DEFAULT, R_DEFAULT
    from oracle import c_oracle as orc
    eng = Engine(0)
    eng.set_noise(Q_DEFAULT, R_DEFAULT)
    B, T = 96, 12
    d = synth_numpy(B, T, seed=31)
    rng = np.random.default_rng(0)
    c = np.zeros((B, T, 4), dtype=np.uint8)
    for b in range(B):
        for t in range(T):
            c[b, t, rng.permutation(4)[:b % 5]] = 1                    # 0..4 stance legs
    d["contact"] = c
    clean = {k: d[k].copy() for k in ("p", "dp")}
    poison = [np.nan, np.inf, -np.inf]
    for b in range(B):
        for t in range(T):
            for l in range(4):
                v = poison[(b + t + l) % 3]
                if c[b, t, l] == 0:                                    # swing: odometry must not look at dp_x, dp_y (p_z feeds next_state: left clean)
                    d["dp"][b, t, 3 * l] = v; d["dp"][b, t, 3 * l + 1] = v
                else:                                                  # stance: dp_z is skipped
                    d["dp"][b, t, 3 * l + 2] = v
    # (1) the odometry kernel alone, every (b, t) as one batch entry; here p_z of swing legs is poisoned as well
    n = B * T
    p1, dp1 = d["p"].reshape(n, 12).copy(), d["dp"].reshape(n, 12).copy()
    c1, imu1 = c.reshape(n, 4), d["imu"].reshape(n, 6)
    for i in range(n):
        for l in range(4):
            if c1[i, l] == 0:
                p1[i, 3 * l + 2] = np.nan
    up = lambda a: torch.as_tensor(np.ascontiguousarray(a.T)).cuda()
    z = torch.empty((10, n), dtype=torch.float32, device="cuda")
    pk = torch.as_tensor(c1.copy().view(np.int32).reshape(n).copy()).cuda()
    pt, dpt, it = up(p1), up(dp1), up(imu1)
    eng._check(eng.lib.os_kf_odom(eng._h, n, _ptr(pt), _ptr(dpt), _ptr(pk), _ptr(it), _ptr(z), eng._stream()), "os_kf_odom")
    zz = z.cpu().numpy().T
    assert np.isfinite(zz).all()
    pc, dpc = clean["p"].reshape(n, 12), clean["dp"].reshape(n, 12)
    for i in range(0, n, 7):
        od = orc.get_odom(pc[i].astype(np.float64), dpc[i].astype(np.float64), c1[i], imu1[i].astype(np.float64))
        assert abs(zz[i, 3] - od[0]) < 2e-6 and np.abs(zz[i, 7:10] - od[1:]).max() < 2e-6
    # (2) whole filter runs: poisoned dp against the oracle on the clean copy (dp enters the filter through get_odom only)
    ref = orc.kf_run_batch(d["p"], d["f"], clean["dp"], d["imu"], c, d["x0"], np.tile(Q_DEFAULT, (B, 1, 1)), Q_DEFAULT, R_DEFAULT)
    s = {k: eng.pack(torch.as_tensor(d[k])) for k in ("p", "f", "dp", "imu")}
    cp = eng.pack_contact(torch.as_tensor(c))
    for kw in (dict(sequential=False, symmetric=False), dict(sequential=True, symmetric=False, lane_per_trajectory=True),
               dict(sequential=True, symmetric=True, lane_per_trajectory=True), dict(sequential=True, symmetric=True)):
        x = torch.as_tensor(d["x0"].T.copy()).cuda()
        P = torch.as_tensor(np.tile(Q_DEFAULT.astype(np.float32).reshape(144, 1), (1, B))).cuda()
        r = eng.kf_run(s["p"], s["f"], s["dp"], s["imu"], cp, x, P, **kw)
        assert int(r["status"].abs().sum()) == 0, kw
        assert np.abs(eng.unpack(r["x_out"]).cpu().numpy() - ref["x"]).max() < 1e-4, kw


def test_backward_refuses_sizes_beyond_its_32_bit_buffer_offsets():
    """bwd_sweep_kernel addresses saved activations / gate derivatives through buffer descriptors with 32-bit byte
    offsets: T*B*4*H*4 >= 4 GiB (H = 128: T*B >= 2.1 M rows, e.g. the 65,536 x 100 bench shape) must be an error, not
    silently wrong gradients.  The guard fires before any buffer is touched, so dummy pointers are enough here."""
    import ctypes as C
    from optistate_amd import Engine, _capi
    eng = Engine(0)
    d = _capi.OsGruDims(60, 128, 2, 24, 1)
    dummy = torch.zeros(64, device="cuda")
    ptr = C.c_void_p(dummy.data_ptr())
    rc = eng.lib.os_gru_backward_ws(eng._h, C.byref(d), ptr, 65536, 100, ptr, ptr, ptr, ptr, ptr, None, eng._stream())
    assert rc == -2
    assert b"32-bit" in eng.lib.os_last_error(eng._h)
    # just below the limit the guard stays quiet (T * B * 4 * H * 4 bytes = 4 GiB - 2 KiB): nothing is launched with these
    # dummy pointers either -- a null x is refused first
    rc = eng.lib.os_gru_backward_ws(eng._h, C.byref(d), ptr, 2 * 1024 * 1024 - 1, 1, None, ptr, ptr, ptr, ptr, None, eng._stream())
    assert rc == -2 and b"32-bit" not in eng.lib.os_last_error(eng._h)


def test_alternating_models_reselect_their_packed_image_instead_of_repacking():
    """gru_train.py:205-217 trains / evaluates `num_models` networks: models alternating on one context are re-selected from
    the library's LRU of packed images (os_gru_load_keyed), a changed model is packed again, a fifth model evicts the least
    recently used one -- and every forward still runs on the right weights."""
    from optistate_amd import RNN
    torch.manual_seed(3)
    ms = [RNN(60, 64, 1, 24, torch.device("cuda")).to("cuda") for _ in range(5)]
    x = torch.rand(7, 10, 60).cuda()
    eng = None
    with torch.no_grad():
        outs = [m(x) for m in ms[:3]]
        eng = ms[0]._engine
        g0 = eng.gru_generation()
        for _ in range(3):                                             # A, B, C, A, B, C, ...: no pack at all
            for m, o in zip(ms[:3], outs):
                assert torch.equal(m(x), o)
        assert eng.gru_generation() == g0
        ms[1].fc.bias.add_(0.5)                                        # torch bumps the version: new key, one pack
        o1 = ms[1](x)
        assert eng.gru_generation() == g0 + 1 and (o1 - outs[1]).abs().max().item() > 1e-3
        assert np.abs(o1.cpu().numpy() - _ref_out(ms[1], x)).max() < 1e-5
        assert torch.equal(ms[0](x), outs[0]) and torch.equal(ms[2](x), outs[2]) and eng.gru_generation() == g0 + 1
        for m in ms:                                                   # five models through four slots: still correct
            assert np.abs(m(x).cpu().numpy() - _ref_out(m, x)).max() < 1e-5
        assert eng.gru_generation() > g0 + 1


def test_latent_in_place_equals_the_copying_path():
    """fused_run(gru_input=...) -- the ViT latent packed straight into rows 60.. of the GRU input buffer, Kalman features written
    in place -- against fused_run(latent=...) (feature scratch + a copy of the latent) and against the oracle."""
    from optistate_amd import Engine, RNN, flatten_state_dict
    from optistate_amd.synth import synth_numpy, Q_DEFAULT, R_DEFAULT
    from oracle import c_oracle as orc
    B, T, NL = 96, 12, 128
    eng = Engine(0); eng.set_noise(Q_DEFAULT, R_DEFAULT)
    d = synth_numpy(B, T, seed=21)
    torch.manual_seed(5)
    m = RNN(60 + NL, 128, 2, 24, torch.device("cpu"))
    eng.load_gru(flatten_state_dict(m.state_dict(), 2), 60 + NL, 128, 2, 24)
    lat = torch.rand(B, T, NL)
    s = {k: eng.pack(torch.as_tensor(d[k])) for k in ("p", "f", "dp", "imu", "accel")}
    c = eng.pack_contact(torch.as_tensor(d["contact"]))
    mm = torch.stack([torch.full((60,), -30.0), torch.full((60,), 30.0)]).cuda()
    st = lambda: (torch.as_tensor(d["x0"].T.copy()).cuda(), torch.as_tensor(np.tile(Q_DEFAULT.astype(np.float32).reshape(144, 1), (1, B))).cuda())
    x, P = st()
    a = eng.fused_run(s["p"], s["f"], s["dp"], s["imu"], c, s["accel"], mm, x, P, latent=eng.pack(lat))
    buf = eng.gru_input_with_latent(lat.cuda())
    assert buf.shape == (T, 60 + NL, B) and torch.equal(buf[:, 60:], eng.pack(lat))
    x, P = st()
    b = eng.fused_run(s["p"], s["f"], s["dp"], s["imu"], c, s["accel"], mm, x, P, gru_input=buf)
    assert torch.equal(a["out"], b["out"]) and torch.equal(a["x_out"], b["x_out"])
    ref = orc.kf_run_batch(d["p"], d["f"], d["dp"], d["imu"], d["contact"], d["x0"], np.tile(Q_DEFAULT, (B, 1, 1)), Q_DEFAULT, R_DEFAULT)
    rows = np.concatenate([(np.concatenate([ref["x"], d["accel"], d["f"], ref["p_rot"], d["dp"], d["imu"]], axis=2) + 30.0) / 60.0,
                           lat.numpy().astype(np.float64)], axis=2)
    ro, _, _ = orc.gru_forward(rows, orc.flatten_state_dict(m.state_dict(), 2), 60 + NL, 128, 2, 24)
    assert np.abs(b["out"].cpu().numpy() - ro).max() < 1e-5
    assert np.abs(buf[:, :60].permute(2, 0, 1).cpu().numpy() - rows[:, :, :60]).max() < 1e-5       # the features landed in rows 0..59
