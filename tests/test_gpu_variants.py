"""GPU: the kernel instantiations that no other test of the suite launched (found by tools/kernel_coverage.py: the -m gpu suite under
rocprofv3 --kernel-trace against the library's code objects, round 5): general-Q forms of the fused and lane kernels, outputs that
switch a template parameter at large batches, float64-P single-step pieces of the C-ABI, the 64-row gi tile of the window stream,
the training forward on the eight-wave split kernel, the 188-wide unfused weight-gradient kernel, the ViT tuning knobs' other arms.
Every case is a parity test against the float64 oracle (or the exact default path), not a smoke run."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

STATE_TOL = 1e-4


def _full_q(seed=31):
    from optistate_amd.synth import Q_DEFAULT
    A = np.random.default_rng(seed).normal(0, 1, (12, 12))
    return Q_DEFAULT + 1e-4 * (A @ A.T) / 12            # symmetric, not diagonal: the interface takes it, the reference's are diagonal


def _kf_ref(d, Q, R, **kw):
    from oracle import c_oracle as orc
    B = d["p"].shape[0]
    return orc.kf_run_batch(d["p"], d["f"], d["dp"], d["imu"], d["contact"], d["x0"], np.tile(Q, (B, 1, 1)), Q, R, **kw)


def _feature_rows(d, ref):
    return np.concatenate([ref["x"], d["accel"].astype(np.float64), d["f"].astype(np.float64), ref["p_rot"],
                           d["dp"].astype(np.float64), d["imu"].astype(np.float64)], axis=2)


@pytest.mark.parametrize("layers,split,two_kernel,B,tile", [(1, False, False, 700, 256), (2, False, False, 700, 256), (1, 3, False, 700, 0), (1, 2, False, 700, 0),
                                                            (2, False, True, 8256, 0), (1, False, False, 700, 128), (2, False, False, 700, 128),
                                                            (1, False, False, 700, 64), (1, False, False, 700, 32), (1, False, False, 700, 16)])
def test_fused_run_with_a_full_process_noise_matrix(layers, split, two_kernel, B, tile):
    """fused_kf_gru_kernel_v2<fullQ, *, 2 | 1> (tiles 256 / 128), fused_kf_gru_kernel_v3<fullQ, 1 | 2 | 4> (tiles 64 / 32 / 16),
    fused_kf_gru_bf16_kernel<fullQ, 2|3> and -- two-kernel path at a batch past the rows kernel's range --
    kf_run_sym_kernel<features, fullQ>: KF state and GRU head against the oracle chain."""
    from optistate_amd import Engine, RNN, flatten_state_dict
    from optistate_amd.synth import synth_numpy, R_FITTED
    from oracle import c_oracle as orc
    T = 12
    Q, R = _full_q(), R_FITTED
    d = synth_numpy(B, T, seed=41)
    ref = _kf_ref(d, Q, R)
    rows = _feature_rows(d, ref)
    mn, mx = rows.reshape(-1, 60).min(0), rows.reshape(-1, 60).max(0)
    torch.manual_seed(7)
    m = RNN(60, 64, layers, 24, torch.device("cpu"))
    ref_out, _, _ = orc.gru_forward((rows - mn) / (mx - mn), orc.flatten_state_dict(m.state_dict(), layers), 60, 64, layers, 24)
    eng = Engine(0)
    eng.set_noise(Q, R)
    eng.set_fused_tile(tile)
    eng.load_gru(flatten_state_dict(m.state_dict(), layers), 60, 64, layers, 24)
    s = {k: eng.pack(torch.as_tensor(d[k])) for k in ("p", "f", "dp", "imu", "accel")}
    c = eng.pack_contact(torch.as_tensor(d["contact"]))
    x = torch.as_tensor(d["x0"].T.copy()).cuda()
    P = torch.as_tensor(np.tile(Q.astype(np.float32).reshape(144, 1), (1, B))).cuda()
    mm = torch.as_tensor(np.stack([mn, mx]).astype(np.float32)).cuda()
    r = eng.fused_run(s["p"], s["f"], s["dp"], s["imu"], c, s["accel"], mm, x, P, two_kernel=two_kernel, split_bf16=split)
    torch.cuda.synchronize()
    if two_kernel:
        assert eng.kernel_name("kf") == "kf_run_sym_kernel"
    elif split:
        assert eng.kernel_name("fused").startswith("fused_kf_gru_bf16_kernel")
    else:
        want = {256: "fused_kf_gru_kernel_v2", 128: "fused_kf_gru_kernel_v2<32 per wave>", 64: "fused_kf_gru_kernel_v3<1>",
                32: "fused_kf_gru_kernel_v3<2>", 16: "fused_kf_gru_kernel_v3<4>"}[tile]
        assert eng.kernel_name("fused") == want
    assert int(eng.failed(r["status"]).sum()) == 0
    assert np.abs(eng.unpack(r["x_out"]).cpu().numpy() - ref["x"]).max() < STATE_TOL
    assert np.abs(r["out"].cpu().numpy() - ref_out).max() < 1e-5


@pytest.mark.parametrize("full_q,want", [(False, "p_rot"), (True, "p_rot"), (True, "trace")])
def test_lane_kernel_outputs_at_a_batch_past_the_rows_kernel(full_q, want):
    """kf_run_sym_kernel<plain without the half-step-ahead pick-up, diag | full Q> (the rotated foot positions are written from the
    plain body) and <P_trace, full Q>, B = 8,256: state, rotated feet / P_trace / K_gain and the final P against the oracle."""
    from optistate_amd import Engine
    from optistate_amd.synth import synth_numpy, Q_DEFAULT, R_DEFAULT
    B, T = 8256, 6
    Q, R = (_full_q(32) if full_q else Q_DEFAULT), R_DEFAULT
    d = synth_numpy(B, T, seed=42)
    ref = _kf_ref(d, Q, R)
    eng = Engine(0)
    eng.set_noise(Q, R)
    s = {k: eng.pack(torch.as_tensor(d[k])) for k in ("p", "f", "dp", "imu")}
    c = eng.pack_contact(torch.as_tensor(d["contact"]))
    x = torch.as_tensor(d["x0"].T.copy()).cuda()
    P = torch.as_tensor(np.tile(Q.astype(np.float32).reshape(144, 1), (1, B))).cuda()
    r = eng.kf_run(s["p"], s["f"], s["dp"], s["imu"], c, x, P, want_p_rot=want == "p_rot", want_trace=want == "trace", want_gain=want == "trace")
    torch.cuda.synchronize()
    assert eng.kernel_name("kf") == "kf_run_sym_kernel"
    assert int(eng.failed(r["status"]).sum()) == 0
    assert np.abs(eng.unpack(r["x_out"]).cpu().numpy() - ref["x"]).max() < STATE_TOL
    if want == "p_rot":
        assert np.abs(eng.unpack(r["p_rot"]).cpu().numpy() - ref["p_rot"]).max() < 1e-5
    else:
        assert np.abs(r["P_trace"].cpu().numpy().T / ref["P_trace"] - 1).max() < 1e-3
        assert np.abs(r["K_gain"].cpu().numpy().T - ref["K_gain"]).max() < 1e-3 * max(1.0, np.abs(ref["K_gain"]).max())
    Pf = P.cpu().numpy().T.reshape(B, 12, 12)
    assert np.abs(Pf - ref["P_final"]).max() < 1e-3 * np.abs(ref["P_final"]).max()


@pytest.mark.parametrize("dense", [False, True], ids=["predict(p,f)", "predict_mpc"])
def test_two_kernel_fused_path_with_the_full_p_sequential_filter(dense):
    """kf_dense_rows_kernel<SEQ, -, FEAT, predict(p,f) | predict_mpc>: the feature rows of the two-kernel fused path from the float64
    row-layout filter with the SEQUENTIAL update (symmetric=False keeps the full P; dense_fd takes predict_mpc's covariance)."""
    from optistate_amd import Engine, RNN, flatten_state_dict
    from optistate_amd.synth import synth_numpy, Q_FITTED, R_FITTED
    from oracle import c_oracle as orc
    B, T = (150, 14) if dense else (8256, 6)          # (below 8,193 trajectories the predict(p,f) case is the rows kernel's, which keeps the full P too)
    Q, R = Q_FITTED, R_FITTED
    d = synth_numpy(B, T, seed=43)
    kw = {}
    if dense:
        d["body_ref"] = np.zeros((B, T, 12), dtype=np.float32)
        d["body_ref"][..., 0:3] = d["imu"][..., 0:3]
        kw = dict(body_ref=d["body_ref"], mode=1)
    ref = _kf_ref(d, Q, R, **kw)
    rows = _feature_rows(d, ref)
    mn, mx = rows.reshape(-1, 60).min(0), rows.reshape(-1, 60).max(0)
    torch.manual_seed(8)
    m = RNN(60, 64, 2, 24, torch.device("cpu"))
    ref_out, _, _ = orc.gru_forward((rows - mn) / (mx - mn), orc.flatten_state_dict(m.state_dict(), 2), 60, 64, 2, 24)
    eng = Engine(0)
    eng.set_noise(Q, R)
    eng.load_gru(flatten_state_dict(m.state_dict(), 2), 60, 64, 2, 24)
    s = {k: eng.pack(torch.as_tensor(d[k])) for k in ("p", "f", "dp", "imu", "accel")}
    c = eng.pack_contact(torch.as_tensor(d["contact"]))
    br = eng.pack(torch.as_tensor(d["body_ref"])) if dense else None
    x = torch.as_tensor(d["x0"].T.copy()).cuda()
    P = torch.as_tensor(np.tile(Q.astype(np.float32).reshape(144, 1), (1, B))).cuda()
    mm = torch.as_tensor(np.stack([mn, mx]).astype(np.float32)).cuda()
    r = eng.fused_run(s["p"], s["f"], s["dp"], s["imu"], c, s["accel"], mm, x, P, body_ref=br, dense_fd=dense, sequential=True, symmetric=False,
                      two_kernel=True)
    torch.cuda.synchronize()
    assert eng.kernel_name("kf").startswith("kf_dense_rows_kernel<SEQ")
    assert int(eng.failed(r["status"]).sum()) == 0
    assert np.abs(eng.unpack(r["x_out"]).cpu().numpy() - ref["x"]).max() < STATE_TOL
    assert np.abs(r["out"].cpu().numpy() - ref_out).max() < 1e-5


@pytest.mark.parametrize("dense", [False, True], ids=["predict", "predict_mpc"])
@pytest.mark.parametrize("sequential", [False, True], ids=["batch", "sequential"])
def test_single_step_pieces_with_a_float64_covariance(dense, sequential):
    """os_kf_predict / os_kf_update with OS_KF_P_FLOAT64 (P and K as DOUBLE arrays; kf_predict_rows_kernel<*, double>,
    kf_update_rows_kernel<*, double>): eight predict -> update rounds through the C-ABI against the oracle's float64 pieces; the
    float64 P must stay within 1e-6 relative of the oracle's (the float32-P pieces are held to 1e-3)."""
    from optistate_amd import Engine, _capi
    from optistate_amd.engine import _ptr
    from optistate_amd.synth import synth_numpy, Q_FITTED, R_FITTED
    B, T = 37, 8
    Q, R = Q_FITTED, R_FITTED
    d = synth_numpy(B, T, seed=44)
    kw = {}
    if dense:
        d["body_ref"] = np.zeros((B, T, 12), dtype=np.float32)
        d["body_ref"][..., 0:3] = d["imu"][..., 0:3]
        kw = dict(body_ref=d["body_ref"], mode=1)
    ref = _kf_ref(d, Q, R, **kw)                            # (the oracle's batch update; the sequential form is the same filter for a diagonal R)
    eng = Engine(0)
    eng.set_noise(Q, R)
    x = torch.as_tensor(d["x0"].T.copy()).cuda()
    P = torch.as_tensor(np.tile(np.asarray(Q, dtype=np.float32).astype(np.float64).reshape(144, 1), (1, B))).cuda()       # float64 [144][B]
    assert P.dtype == torch.float64
    cpk = eng.pack_contact(torch.as_tensor(d["contact"]))                                                         # [T][B] packed
    col = lambda k, t: torch.as_tensor(np.ascontiguousarray(d[k][:, t].T)).cuda()
    status = torch.empty((B,), dtype=torch.int32, device="cuda")
    ptr, kg = torch.empty((B,), device="cuda"), torch.empty((B,), device="cuda")
    vp = lambda t_: C.c_void_p(t_.data_ptr())
    fl = _capi.OS_KF_P_FLOAT64
    worst = 0.0
    for t in range(T):
        p, f, dp, imu = col("p", t), col("f", t), col("dp", t), col("imu", t)
        br = col("body_ref", t) if dense else None
        z = eng.kf_odom(p, dp, cpk[t].contiguous(), imu)
        eng._check(eng.lib.os_kf_predict(eng._h, B, _ptr(p), _ptr(f), _ptr(br), _ptr(x), vp(P), None,
                                         fl | (_capi.OS_KF_DENSE_FD if dense else 0), eng._stream()), "os_kf_predict")
        eng._check(eng.lib.os_kf_update(eng._h, B, _ptr(z), _ptr(x), vp(P), None, _ptr(ptr), _ptr(kg), _ptr(status),
                                        fl | (_capi.OS_KF_SEQUENTIAL_UPDATE if sequential else 0), eng._stream()), "os_kf_update")
        torch.cuda.synchronize()
        assert int(eng.failed(status).sum()) == 0
        worst = max(worst, float(np.abs(x.cpu().numpy().T - ref["x"][:, t]).max()))
        assert np.abs(ptr.cpu().numpy() / ref["P_trace"][:, t] - 1).max() < 1e-4
    assert worst < STATE_TOL, worst
    Pf = P.cpu().numpy().T.reshape(B, 12, 12)
    assert np.abs(Pf - ref["P_final"]).max() < (2e-5 if not dense else 1e-4) * np.abs(ref["P_final"]).max()


def test_window_stream_past_the_64_row_gi_tile():
    """os_gru_forward_windows on a stream long enough for gru_gi_kernel<2> (64-row tiles from 2 x 64 x CUs rows on): against the
    materialised windows on the exact path (all windows) and the float64 oracle (a sample)."""
    from optistate_amd import RNN
    from oracle import c_oracle as orc
    I, H, L, Cc, W = 188, 128, 2, 24, 10
    torch.manual_seed(9)
    m = RNN(I, H, L, Cc, torch.device("cuda")).to("cuda").eval()
    N = 64 * 2 * torch.cuda.get_device_properties(0).multi_processor_count + 777
    rows = torch.rand(N, I)
    with torch.no_grad():
        out = m.forward_windows(rows.cuda(), W).cpu().numpy()
        assert m._engine.kernel_name("gru_layer") != ""
        idx = np.concatenate([np.arange(0, 40), np.arange(N - W + 1 - 40, N - W + 1), np.random.default_rng(0).choice(N - W + 1, 120, replace=False)])
        win = torch.stack([rows[i:i + W] for i in idx])
        mat = m(win.cuda()).cpu().numpy()
    assert out.shape == (N - W + 1, Cc)
    assert np.abs(out[idx] - mat).max() < 2e-6
    ref, _, _ = orc.gru_forward(win[:48].numpy(), orc.flatten_state_dict(m.state_dict(), L), I, H, L, Cc)
    assert np.abs(out[idx[:48]] - ref).max() < 1e-5


def test_training_forward_of_a_single_h128_layer_on_the_split_kernel(monkeypatch):
    """gru_layer_split_kernel<4, SAVE>: one H = 128 layer (no stack launch), the run-ahead kernel switched off (OS_GRU_AHEAD=0, the
    A/B knob): output and every gradient against fp64 autograd."""
    from optistate_amd import Engine, RNN, flatten_state_dict
    monkeypatch.setenv("OS_GRU_AHEAD", "0")
    I, H, Cc, B, T = 100, 128, 24, 96, 7
    torch.manual_seed(10)
    m = RNN(I, H, 1, Cc, torch.device("cpu"))
    eng = Engine(0)
    eng.load_gru(flatten_state_dict(m.state_dict(), 1), I, H, 1, Cc)
    x = torch.rand(B, T, I) * 2 - 1
    y = torch.rand(B, Cc // 2, device="cuda")
    o = eng.gru_forward_train(x.cuda())
    assert eng.kernel_name("gru_layer") == "gru_layer_split_kernel"
    _, dout, _ = eng.gru_loss(o, y, want_target=True)
    g = eng.gru_backward(x.cuda(), o, dout).double().cpu()
    md = torch.nn.GRU(I, H, 1, batch_first=True).double()
    fc = torch.nn.Linear(H, Cc).double()
    sd = {k: v.double() for k, v in m.state_dict().items()}
    md.load_state_dict({k[4:]: v for k, v in sd.items() if k.startswith("gru.")})
    fc.load_state_dict({k[3:]: v for k, v in sd.items() if k.startswith("fc.")})
    hseq, _ = md(x.double())
    od = torch.sigmoid(fc(hseq[:, -1]))
    od.backward(dout.double().cpu())
    refg = torch.cat([p.grad.reshape(-1) for p in list(md.parameters()) + list(fc.parameters())])
    assert (o.double().cpu() - od.detach()).abs().max().item() < 1e-5
    assert (g - refg).abs().max().item() < 1e-4 * refg.abs().max().item()


def test_weight_gradients_of_a_188_wide_layer_under_a_64_wide_hidden_state():
    """dw2_kernel<6>: 188 input columns with hidden 64 (the fused two-product kernel covers 188 columns only at hidden 128): gradients
    against fp64 autograd."""
    from optistate_amd import RNN
    I, H, L, Cc, B, T = 188, 64, 2, 24, 64, 5
    torch.manual_seed(11)
    m = RNN(I, H, L, Cc, torch.device("cuda")).to("cuda")
    x = torch.rand(B, T, I) * 2 - 1
    y = torch.rand(B, Cc // 2)
    xg = x.cuda()
    out = m(xg)
    tgt = torch.cat([y.cuda(), (out[:, :Cc // 2].detach() - y.cuda()).abs()], dim=1)
    loss = torch.nn.functional.mse_loss(out, tgt)
    loss.backward()
    md = torch.nn.GRU(I, H, L, batch_first=True).double()
    fc = torch.nn.Linear(H, Cc).double()
    sd = {k: v.detach().double().cpu() for k, v in m.state_dict().items()}
    md.load_state_dict({k[4:]: v for k, v in sd.items() if k.startswith("gru.")})
    fc.load_state_dict({k[3:]: v for k, v in sd.items() if k.startswith("fc.")})
    hseq, _ = md(x.double())
    od = torch.sigmoid(fc(hseq[:, -1]))
    tg = torch.cat([y.double(), (od[:, :Cc // 2].detach() - y.double()).abs()], dim=1)
    torch.nn.functional.mse_loss(od, tg).backward()
    ref = {**{f"gru.{k}": p.grad for k, p in md.named_parameters()}, **{f"fc.{k}": p.grad for k, p in fc.named_parameters()}}
    for k, p in m.named_parameters():
        r = ref[k]
        assert (p.grad.double().cpu() - r).abs().max().item() < 2e-4 * max(r.abs().max().item(), 1e-8) + 1e-9, k


@pytest.mark.parametrize("knob,value,phase,name", [("OS_VIT_ATT_DMA", "0", "vit_attn", "attention_mfma_kernel<7>"),
                                                   ("OS_VIT_MLP_BM", "128", "vit_gemm", "vit_mlp_kernel")])
def test_vit_tuning_knobs_other_arms_against_the_oracle(monkeypatch, knob, value, phase, name):
    """attention_mfma_kernel<7> (one workgroup per head) and vit_mlp_kernel (128-row tiles): the non-default arms of two tuning knobs,
    on a frame count with tile tails, against the float64 restatement."""
    from optistate_amd import Engine, _capi
    from optistate_amd.engine import _ptr
    from optistate_amd.transformer_model import Transformer_Autoencoder
    from oracle import vit_oracle
    monkeypatch.setenv(knob, value)
    torch.manual_seed(12)
    m = Transformer_Autoencoder().to("cuda")
    with torch.no_grad():
        for p in m.parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    e = Engine(0)
    d = _capi.OsVitDims(m.img_size, m.patch_size, m.in_chans, m.embed_dim, m.depth, m.num_heads, m.mlp_hidden)
    flat = m._flat(torch.device("cuda:0"))
    e._check(e.lib.os_vit_load(e._h, C.byref(d), _ptr(flat)), "os_vit_load")
    N = 9
    img = torch.rand(N, 224, 224, device="cuda")
    lat = torch.empty((N, 128), dtype=torch.float32, device="cuda")
    e._check(e.lib.os_vit_encode(e._h, N, _ptr(img), _ptr(lat), e._stream()), "os_vit_encode")
    torch.cuda.synchronize()
    assert e.kernel_name(phase) == name, e.kernel_name(phase)
    sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    ref = vit_oracle.encode(img.cpu().numpy(), sd)
    assert np.abs(lat.cpu().numpy() - ref).max() < 2e-5
