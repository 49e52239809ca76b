"""GPU parity: HIP GRU kernels vs the reference RNN's outputs (golden) and vs the float64 oracle."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu

GRU_TOL = 1e-5   # exact-fp32 MFMA path; sigmoid outputs in (0,1)


def load_module(name):
    from optistate_amd import RNN
    g = load_golden(f"gru_g5_{name}.npz")
    I, H, L, C = [int(v) for v in g["dims"]]
    m = RNN(I, H, L, C, torch.device("cuda"))
    sd = {k[2:]: torch.as_tensor(g[k]) for k in g.files if k.startswith("w:")}
    m.load_state_dict(sd)           # the reference's keys must load unchanged
    return m.to("cuda").eval(), g


@pytest.mark.parametrize("name", ["small", "ref"])
def test_g5_forward_matches_reference(name):
    m, g = load_module(name)
    with torch.no_grad():
        out = m(torch.as_tensor(g["x"]).cuda())
    torch.cuda.synchronize()
    err = np.abs(out.cpu().numpy() - g["out"]).max()
    assert out.shape == g["out"].shape
    assert err < GRU_TOL, err


@pytest.mark.parametrize("name", ["small", "ref"])
def test_g5_one_window_per_call_matches_reference(name):
    """The reference's evaluation loop feeds ONE window per call (gru/gru_test.py:157-177): the golden batch's windows one at a time
    (and four at a time) through the drop-in class -- gru_vec_kernel where the model's widths allow it -- against the reference's
    own outputs for them."""
    m, g = load_module(name)
    x = torch.as_tensor(g["x"]).cuda()
    n = min(6, x.shape[0])
    with torch.no_grad():
        for i in range(n):
            out = m(x[i:i + 1])
            assert np.abs(out.cpu().numpy() - g["out"][i:i + 1]).max() < GRU_TOL, i
        if x.shape[1] * 4 <= 48 and m.hidden_size in (64, 128) and m.input_size <= 192:
            assert m._engine.kernel_name("gru_layer") == "gru_vec_kernel", m._engine.kernel_name("gru_layer")
        out4 = m(x[:4])
    assert np.abs(out4.cpu().numpy() - g["out"][:4]).max() < GRU_TOL


@pytest.mark.parametrize("name", ["small", "ref"])
def test_h_last_matches_reference(name):
    m, g = load_module(name)
    x = torch.as_tensor(g["x"]).cuda()
    with torch.no_grad():
        m(x)
    out, hl = m._engine.gru_forward(x, want_h_last=True)
    assert np.abs(hl.cpu().numpy() - g["h_last"]).max() < GRU_TOL


@pytest.mark.parametrize("dims", [(60, 64, 1, 24), (60, 64, 4, 24), (61, 32, 2, 5), (60, 128, 4, 24)])
def test_ragged_batch_vs_oracle(dims):
    """B not a multiple of the 128-row tile, odd input width, several layer counts; float64 oracle as truth."""
    from optistate_amd import RNN
    from oracle import c_oracle as orc
    I, H, L, C = dims
    torch.manual_seed(3)
    m = RNN(I, H, L, C, torch.device("cuda")).to("cuda").eval()
    B, T = 333, 7
    x = torch.rand(B, T, I)
    with torch.no_grad():
        out = m(x.cuda()).cpu().numpy()
    w = orc.flatten_state_dict(m.state_dict(), L)
    ref, _, _ = orc.gru_forward(x.numpy(), w, I, H, L, C)
    err = np.abs(out - ref).max()
    assert err < GRU_TOL, err


@pytest.mark.parametrize("I,B,T", [(61, 1, 1), (60, 33, 2), (188, 40, 3), (128, 257, 5)])
def test_run_ahead_layer_kernel_edges(monkeypatch, I, B, T):
    """H = 128 at small batch, a launch per layer (OS_GRU_STACK=0), takes gru_layer_ahead_kernel (input half one to two steps ahead, (B, T, I) read directly): its
    prologue / epilogue at T = 1, 2, 3, odd and 188-wide inputs, partial tiles; float64 oracle as truth, and the eight-wave
    split kernel (OS_GRU_AHEAD=0) must give the same numbers up to summation order."""
    import os
    from optistate_amd import RNN, Engine
    from optistate_amd import engine as eng_mod
    from oracle import c_oracle as orc
    H, L, C = 128, 2, 24
    monkeypatch.setenv("OS_GRU_STACK", "0")                  # tuning knobs are read when an Engine is created
    monkeypatch.setenv("OS_GRU_VEC", "0")
    torch.manual_seed(11)
    m = RNN(I, H, L, C, torch.device("cpu"))
    x = torch.rand(B, T, I)
    w = orc.flatten_state_dict(m.state_dict(), L)
    e1 = Engine(0)
    e1.load_gru(torch.as_tensor(w, dtype=torch.float32).cuda(), I, H, L, C)
    out = e1.gru_forward(x.cuda())
    out = (out[0] if isinstance(out, (tuple, list)) else out).cpu().numpy()
    assert e1.kernel_name("gru_layer") == "gru_layer_ahead_kernel"
    ref, _, _ = orc.gru_forward(x.numpy(), w, I, H, L, C)
    assert np.abs(out - ref).max() < GRU_TOL
    os.environ["OS_GRU_AHEAD"] = "0"
    try:
        e2 = Engine(0)
        e2.load_gru(torch.as_tensor(w, dtype=torch.float32).cuda(), I, H, L, C)
        out2 = e2.gru_forward(x.cuda())
        out2 = (out2[0] if isinstance(out2, (tuple, list)) else out2).cpu().numpy()
        assert e2.kernel_name("gru_layer") == "gru_layer_split_kernel"
    finally:
        del os.environ["OS_GRU_AHEAD"]
    assert np.abs(out2 - out).max() < 2e-6


@pytest.mark.parametrize("two_kernel,L,B,T", [(False, 1, 200, 20), (True, 1, 200, 20), (False, 4, 333, 12),
                                                (False, 1, 1000, 100)])
def test_fused_matches_oracle_chain(two_kernel, L, B, T):
    """os_fused_run (KF -> 60-feature row -> min-max -> GRU) vs oracle KF -> oracle feature rows -> oracle GRU;
    single-kernel path (features stay in registers), forced two-kernel path, a 4-layer stack, full T = 100."""
    from optistate_amd import Engine, RNN, flatten_state_dict
    from optistate_amd.synth import synth_numpy, Q_FITTED, R_FITTED
    from oracle import c_oracle as orc
    d = synth_numpy(B, T, seed=21)
    ref = orc.kf_run_batch(d["p"], d["f"], d["dp"], d["imu"], d["contact"], d["x0"], np.tile(Q_FITTED, (B, 1, 1)),
                           Q_FITTED, R_FITTED)
    rows = np.concatenate([ref["x"], d["accel"].astype(np.float64), d["f"].astype(np.float64), ref["p_rot"],
                           d["dp"].astype(np.float64), d["imu"].astype(np.float64)], axis=2)      # [B][T][60]
    mn, mx = rows.reshape(-1, 60).min(0), rows.reshape(-1, 60).max(0)
    norm = (rows - mn) / (mx - mn)
    torch.manual_seed(5)
    m = RNN(60, 64, L, 24, torch.device("cpu"))
    w = orc.flatten_state_dict(m.state_dict(), L)
    ref_out, _, _ = orc.gru_forward(norm, w, 60, 64, L, 24)

    eng = Engine(0)
    eng.set_noise(Q_FITTED, R_FITTED)
    eng.load_gru(flatten_state_dict(m.state_dict(), L), 60, 64, L, 24)
    s = {k: eng.pack(torch.as_tensor(d[k])) for k in ("p", "f", "dp", "imu", "accel")}
    c = eng.pack_contact(torch.as_tensor(d["contact"]))
    x = torch.as_tensor(d["x0"].T.copy()).cuda()
    P = torch.as_tensor(np.tile(Q_FITTED.astype(np.float32).reshape(144, 1), (1, B))).cuda()
    mm = torch.as_tensor(np.stack([mn, mx]).astype(np.float32)).cuda()
    r = eng.fused_run(s["p"], s["f"], s["dp"], s["imu"], c, s["accel"], mm, x, P, two_kernel=two_kernel)
    torch.cuda.synchronize()
    xo = eng.unpack(r["x_out"]).cpu().numpy()
    assert int(r["status"].abs().sum()) == 0
    assert np.abs(xo - ref["x"]).max() < 1e-4
    Pf = P.cpu().numpy().T.reshape(B, 12, 12)
    assert np.abs(Pf - ref["P_final"]).max() < 1e-3 * np.abs(ref["P_final"]).max()
    assert np.abs(x.cpu().numpy().T - ref["x_final"]).max() < 1e-4
    err = np.abs(r["out"].cpu().numpy() - ref_out).max()
    assert err < 1e-4, err       # feature rows carry fp32 KF noise (~1e-6) through T GRU steps


@pytest.mark.parametrize("mode", ["batch_update", "dense_fd", "full_Q"])
def test_fused_general_paths(mode):
    """Fused chain through the non-default Kalman variants: batch (Cholesky) update, predict_mpc covariance with body_ref
    (two-kernel path), and a non-diagonal Q in the single kernel (QDIAG = false instantiation)."""
    from optistate_amd import Engine, RNN, flatten_state_dict
    from optistate_amd.synth import synth_numpy, Q_FITTED, R_FITTED
    from oracle import c_oracle as orc
    B, T = 150, 12
    d = synth_numpy(B, T, seed=33)
    rng = np.random.default_rng(5)
    Q = Q_FITTED
    if mode == "full_Q":
        A = rng.normal(size=(12, 12)); Q = (A @ A.T / 12 + np.eye(12)) * 1e-3
    body_ref = None
    if mode == "dense_fd":
        body_ref = np.zeros((B, T, 12), dtype=np.float32)
        body_ref[..., 0:3] = d["imu"][..., 0:3] + rng.normal(0, 0.01, (B, T, 3)).astype(np.float32)
    ref = orc.kf_run_batch(d["p"], d["f"], d["dp"], d["imu"], d["contact"], d["x0"], np.tile(Q, (B, 1, 1)), Q, R_FITTED,
                           body_ref=body_ref, mode=1 if mode == "dense_fd" else 0)
    rows = np.concatenate([ref["x"], d["accel"], d["f"], ref["p_rot"], d["dp"], d["imu"]], axis=2)
    mn, mx = rows.reshape(-1, 60).min(0), rows.reshape(-1, 60).max(0)
    torch.manual_seed(8)
    m = RNN(60, 64, 1, 24, torch.device("cpu"))
    ref_out, _, _ = orc.gru_forward((rows - mn) / (mx - mn), orc.flatten_state_dict(m.state_dict(), 1), 60, 64, 1, 24)
    eng = Engine(0)
    eng.set_noise(Q, R_FITTED)
    eng.load_gru(flatten_state_dict(m.state_dict(), 1), 60, 64, 1, 24)
    s = {k: eng.pack(torch.as_tensor(d[k])) for k in ("p", "f", "dp", "imu", "accel")}
    c = eng.pack_contact(torch.as_tensor(d["contact"]))
    x = torch.as_tensor(d["x0"].T.copy()).cuda()
    P = torch.as_tensor(np.tile(np.asarray(Q, dtype=np.float32).reshape(144, 1), (1, B))).cuda()
    mm = torch.as_tensor(np.stack([mn, mx]).astype(np.float32)).cuda()
    kw = {}
    if mode == "batch_update":
        kw = dict(sequential=False, symmetric=False)
    if mode == "dense_fd":
        kw = dict(body_ref=eng.pack(torch.as_tensor(body_ref)), dense_fd=True, sequential=False, symmetric=False)
    r = eng.fused_run(s["p"], s["f"], s["dp"], s["imu"], c, s["accel"], mm, x, P, **kw)
    torch.cuda.synchronize()
    assert np.abs(eng.unpack(r["x_out"]).cpu().numpy() - ref["x"]).max() < 1e-4
    assert np.abs(r["out"].cpu().numpy() - ref_out).max() < 1e-4


def test_large_batch_kernels_agree_with_small_batch_kernels():
    """The layer kernel has three launch shapes (64-row tiles with two workgroups per CU at large batches, 32-row tiles, and
    the eight-wave split kernel at small batches for H = 128): a ragged large batch must reproduce what its slices give
    through the small-batch kernels (those are the ones the oracle tests above cover)."""
    from optistate_amd import Engine, RNN, flatten_state_dict
    eng = Engine(0)
    for (I, H, L) in ((60, 128, 2), (60, 64, 2)):
        torch.manual_seed(5)
        m = RNN(I, H, L, 24, torch.device("cuda"))
        eng.load_gru(flatten_state_dict(m.state_dict(), L, "cuda"), I, H, L, 24)
        B, T = 40000 + 37, 6
        xs = torch.rand(T, I, B, device="cuda")
        full = eng.gru_forward_soa(xs)
        for lo, n in ((0, 4096), (17000, 333), (B - 2048, 2048)):
            part = eng.gru_forward_soa(xs[:, :, lo:lo + n].contiguous())
            assert (part - full[lo:lo + n]).abs().max().item() < 2e-6, (I, H, L, lo, n)


@pytest.mark.parametrize("I,H", [(188, 128), (60, 128), (61, 128), (60, 64), (64, 64)])
def test_stage_kernel_large_batch_bit_identical_to_plain_kernel_and_close_to_oracle(monkeypatch, I, H):
    """gru_layer_stage_kernel<4 | 2> (H = 128 / 64, at least two row tiles per CU, inference: x tile by LDS-DMA, h in registers,
    16-byte seq_out stores) against gru_layer_kernel<2,2> (OS_GRU_STAGE=0) on the same inputs -- the same products in the same
    order, so the outputs must be IDENTICAL -- and against the float64 oracle on a sample of trajectories, including the last,
    partial tile (B is a multiple of 4, not of 64) and an odd input width (zero-padded k-pair through the range check)."""
    from optistate_amd import Engine, RNN, flatten_state_dict
    from oracle import c_oracle as orc
    B, T, L, C = (32768 if H == 128 else 65536) + 36, 4, 2, 24
    torch.manual_seed(11)
    m = RNN(I, H, L, C, torch.device("cpu"))
    xs = torch.rand(T, I, B, device="cuda") * 2 - 1                    # SoA [T][I][B]
    outs = {}
    for stage in ("1", "0"):
        monkeypatch.setenv("OS_GRU_STAGE", stage)
        e = Engine(0)
        e.load_gru(flatten_state_dict(m.state_dict(), L, "cuda"), I, H, L, C)
        outs[stage] = e.gru_forward_soa(xs)
        torch.cuda.synchronize()
        assert e.kernel_name("gru_layer").startswith("gru_layer_stage_kernel" if stage == "1" else "gru_layer_kernel<2,2>"), e.kernel_name("gru_layer")
    assert torch.equal(outs["1"], outs["0"])
    pick = torch.cat([torch.arange(0, 160), torch.arange(16000, 16064), torch.arange(B - 100, B)])
    x = xs[:, :, pick.cuda()].permute(2, 0, 1).cpu().numpy()           # [n][T][I]
    ref, _, _ = orc.gru_forward(x, orc.flatten_state_dict(m.state_dict(), L), I, H, L, C)
    err = np.abs(outs["1"][pick.cuda()].cpu().numpy() - ref).max()
    assert err < GRU_TOL, err


@pytest.mark.parametrize("I,H,L,B,T", [(188, 128, 4, 128, 10), (188, 128, 4, 1, 10), (60, 64, 4, 64, 100), (61, 32, 2, 33, 7),
                                       (60, 128, 8, 900, 25), (188, 128, 4, 2048, 3), (60, 64, 2, 5, 1), (188, 128, 4, 4000, 3), (60, 64, 5, 2500, 4)])
def test_layer_pipelined_stack_kernel_bit_identical_to_per_layer_launches(monkeypatch, I, H, L, B, T):
    """Small batches run their layer stack as ONE launch (gru_stack_kernel: blockIdx.y = layer, layer l takes step t of layer
    l - 1 through a progress flag as soon as it is published).  The workgroups run gru_layer_split_kernel's body, so against a
    launch per layer of THAT kernel (OS_GRU_STACK=0 OS_GRU_AHEAD=0) the outputs and every layer's h_T must be identical -- any
    consumer that read a step before it was complete would show here -- on every one of 20 repeats; float64 oracle as truth.
    Shapes: the reference's model at config 5's 128 trajectories and at its own batch of 1, 4 x 64 over 100 steps, odd input
    width with a partial tile, eight layers x 29 tiles (232 of 256 CUs), the largest batch that is one launch (64 tiles x 4), T = 1, and
    two batches that take several launches (4,000: two layers at a time; 2,500 x five layers: three + two)."""
    from optistate_amd import Engine, RNN, flatten_state_dict
    from oracle import c_oracle as orc
    C = 24
    torch.manual_seed(23)
    m = RNN(I, H, L, C, torch.device("cpu"))
    x = (torch.rand(B, T, I) * 2 - 1).cuda()
    monkeypatch.setenv("OS_GRU_VEC", "0")                      # (B <= 4 would otherwise take gru_vec_kernel)
    monkeypatch.setenv("OS_GRU_WIDE", "0")                     # (H = 128 up to 512 windows would otherwise take gru_wide_kernel: its own test below)
    monkeypatch.setenv("OS_GRU_STACK", "0"); monkeypatch.setenv("OS_GRU_AHEAD", "0")
    e0 = Engine(0)
    e0.load_gru(flatten_state_dict(m.state_dict(), L, "cuda"), I, H, L, C)
    out0, hl0 = e0.gru_forward(x, want_h_last=True)
    torch.cuda.synchronize()
    assert e0.kernel_name("gru_layer").startswith("gru_layer_split_kernel") or H == 32, e0.kernel_name("gru_layer")
    monkeypatch.setenv("OS_GRU_STACK", "1"); monkeypatch.delenv("OS_GRU_AHEAD")
    e1 = Engine(0)
    e1.load_gru(flatten_state_dict(m.state_dict(), L, "cuda"), I, H, L, C)
    for rep in range(20):
        out1, hl1 = e1.gru_forward(x, want_h_last=True)
        torch.cuda.synchronize()
        assert e1.kernel_name("gru_layer") == "gru_stack_kernel"
        if H == 32:        # a launch per layer runs H = 32 through gru_layer_kernel<1,3>: same sums in another order
            assert (out1 - out0).abs().max().item() < 2e-6 and (hl1 - hl0).abs().max().item() < 2e-6, rep
        else:
            assert torch.equal(out1, out0), (rep, (out1 - out0).abs().max().item())
            assert torch.equal(hl1, hl0), rep
    ref, _, _ = orc.gru_forward(x.cpu().numpy(), orc.flatten_state_dict(m.state_dict(), L), I, H, L, C)
    assert np.abs(out1.cpu().numpy() - ref).max() < GRU_TOL


def test_layer_pipelined_stack_kernel_behind_the_fused_first_layer(monkeypatch):
    """os_fused_run shape: the fused Kalman + GRU layer-0 kernel writes the sequence, layers 1..3 follow as one stack launch
    (first_layer = 1); identical numbers to a launch per layer of the split kernel."""
    from optistate_amd import Engine, RNN, flatten_state_dict
    from optistate_amd.synth import synth_numpy, Q_FITTED, R_FITTED
    I, H, L, C, B, T = 60, 64, 4, 24, 96, 40
    d = synth_numpy(B, T, seed=31)
    torch.manual_seed(29)
    m = RNN(I, H, L, C, torch.device("cpu"))
    outs = {}
    for stack in ("0", "1"):
        monkeypatch.setenv("OS_GRU_STACK", stack)
        eng = Engine(0)
        eng.set_noise(Q_FITTED, R_FITTED)
        eng.load_gru(flatten_state_dict(m.state_dict(), L), I, H, L, C)
        s = {k: eng.pack(torch.as_tensor(d[k])) for k in ("p", "f", "dp", "imu", "accel")}
        c = eng.pack_contact(torch.as_tensor(d["contact"]))
        x = torch.as_tensor(d["x0"].T.copy()).cuda()
        P = torch.as_tensor(np.tile(Q_FITTED.astype(np.float32).reshape(144, 1), (1, B))).cuda()
        mm = torch.stack([torch.full((60,), -3.0), torch.full((60,), 3.0)]).cuda()
        outs[stack] = eng.fused_run(s["p"], s["f"], s["dp"], s["imu"], c, s["accel"], mm, x, P)["out"]
        torch.cuda.synchronize()
        assert (eng.kernel_name("gru_layer") == "gru_stack_kernel") == (stack == "1"), eng.kernel_name("gru_layer")
    assert torch.isfinite(outs["1"]).all()
    assert (outs["1"] - outs["0"]).abs().max().item() < 2e-6


@pytest.mark.parametrize("I,H,L,B,T", [(188, 128, 4, 64, 10), (188, 128, 4, 1, 10), (188, 128, 4, 512, 6), (61, 128, 2, 33, 7), (60, 128, 8, 250, 25),
                                       (192, 128, 3, 100, 1), (128, 128, 2, 97, 2), (1, 128, 2, 40, 5), (188, 128, 4, 8, 120)])
def test_wide_kernel_four_cus_per_layer_tile(monkeypatch, I, H, L, B, T):
    """gru_wide_kernel: the stack of a small H = 128 batch as ONE launch with the 128 hidden units of a (layer, tile) split over four
    workgroups that exchange their slices of h_t every step through progress counters.  Against the float64 oracle, against a launch
    per layer (1e-6: another summation order) for the output and every layer's h_T, and bit for bit against ITSELF on 20 repeats (its
    summation order is fixed: a consumer that read a slice before it was complete would show as a run-to-run difference).  Shapes: the
    reference's model at its training batch, at one window, at the largest batch the kernel takes (16 tiles x 4 layers x 4 = 256
    workgroups); odd input width with a partial tile; eight layers (tiles <= 8); the widest input at T = 1; T = 2; input width 1; 120 steps."""
    from optistate_amd import Engine, RNN, flatten_state_dict
    from oracle import c_oracle as orc
    C = 24
    torch.manual_seed(24)
    m = RNN(I, H, L, C, torch.device("cpu"))
    x = (torch.rand(B, T, I) * 2 - 1).cuda()
    monkeypatch.setenv("OS_GRU_VEC", "0")
    monkeypatch.setenv("OS_GRU_STACK", "0")
    e0 = Engine(0)
    e0.load_gru(flatten_state_dict(m.state_dict(), L, "cuda"), I, H, L, C)
    out0, hl0 = e0.gru_forward(x, want_h_last=True)
    torch.cuda.synchronize()
    assert "wide" not in e0.kernel_name("gru_layer") and "stack" not in e0.kernel_name("gru_layer")
    monkeypatch.setenv("OS_GRU_STACK", "1")
    e1 = Engine(0)
    e1.load_gru(flatten_state_dict(m.state_dict(), L, "cuda"), I, H, L, C)
    first = None
    for rep in range(20):
        out1, hl1 = e1.gru_forward(x, want_h_last=True)
        torch.cuda.synchronize()
        assert e1.kernel_name("gru_layer") == "gru_wide_kernel"
        if first is None:
            first = (out1.clone(), hl1.clone())
        assert torch.equal(out1, first[0]) and torch.equal(hl1, first[1]), rep
    assert (out1 - out0).abs().max().item() < 1e-6 and (hl1 - hl0).abs().max().item() < 2e-6
    # the same kernel fed through its other input layout (os_gru_forward_soa: the [T][I][B] stream instead of the (B, T, I) tensor):
    # only the staging of layer 0's tile differs, the numbers must not
    out_soa = e1.gru_forward_soa(x.permute(1, 2, 0).contiguous())
    torch.cuda.synchronize()
    assert e1.kernel_name("gru_layer") == "gru_wide_kernel" and torch.equal(out_soa, out1)
    ref, _, _ = orc.gru_forward(x.cpu().numpy(), orc.flatten_state_dict(m.state_dict(), L), I, H, L, C)
    assert np.abs(out1.cpu().numpy() - ref).max() < GRU_TOL


@pytest.mark.parametrize("I,H,L,B,T", [(188, 128, 4, 1, 10), (188, 128, 4, 4, 10), (60, 64, 4, 1, 10), (60, 64, 2, 3, 16), (61, 128, 1, 2, 1),
                                       (128, 128, 3, 1, 48), (188, 128, 4, 4, 12), (60, 128, 2, 2, 13), (192, 64, 5, 4, 7)])
def test_single_window_vector_kernel_matches_oracle_and_mfma_path(monkeypatch, I, H, L, B, T):
    """B <= 4 (the reference's own evaluation loop: one window per call, gru/gru_test.py:157-177) runs the whole model -- every
    layer, the head, h_T of every layer -- in ONE single-workgroup launch on the vector pipe (gru_vec_kernel).  Float64 oracle as
    truth (1e-5), the MFMA path (OS_GRU_VEC=0) within summation-order distance; shapes cover one and two passes of phase A
    (B T <= 24 / <= 48, a ragged second pass at 26), odd and maximal input widths, T = 1, both hidden sizes."""
    from optistate_amd import Engine, RNN, flatten_state_dict
    from oracle import c_oracle as orc
    C = 24
    torch.manual_seed(37)
    m = RNN(I, H, L, C, torch.device("cpu"))
    x = (torch.rand(B, T, I) * 2 - 1).cuda()
    res = {}
    for vec in ("1", "0"):
        monkeypatch.setenv("OS_GRU_VEC", vec)
        e = Engine(0)
        e.load_gru(flatten_state_dict(m.state_dict(), L, "cuda"), I, H, L, C)
        for rep in range(3):
            out, hl = e.gru_forward(x, want_h_last=True)
        torch.cuda.synchronize()
        assert (e.kernel_name("gru_layer") == "gru_vec_kernel") == (vec == "1"), e.kernel_name("gru_layer")
        res[vec] = (out.clone(), hl.clone())
    assert (res["1"][0] - res["0"][0]).abs().max().item() < 2e-6
    assert (res["1"][1] - res["0"][1]).abs().max().item() < 2e-6
    ref, hl_ref, _ = orc.gru_forward(x.cpu().numpy(), orc.flatten_state_dict(m.state_dict(), L), I, H, L, C)
    assert np.abs(res["1"][0].cpu().numpy() - ref).max() < GRU_TOL
    assert np.abs(res["1"][1].cpu().numpy() - hl_ref).max() < GRU_TOL


def test_vector_kernel_follows_weight_updates(monkeypatch):
    """The transposed weight image gru_vec_kernel reads is rebuilt when os_gru_load packs new weights (and only then)."""
    from optistate_amd import Engine, RNN, flatten_state_dict
    from oracle import c_oracle as orc
    I, H, L, C = 60, 64, 2, 24
    e = Engine(0)
    x = torch.rand(1, 10, I).cuda()
    for seed in (1, 2):
        torch.manual_seed(seed)
        m = RNN(I, H, L, C, torch.device("cpu"))
        e.load_gru(flatten_state_dict(m.state_dict(), L, "cuda"), I, H, L, C)
        out = e.gru_forward(x)
        out = (out[0] if isinstance(out, (tuple, list)) else out).cpu().numpy()
        ref, _, _ = orc.gru_forward(x.cpu().numpy(), orc.flatten_state_dict(m.state_dict(), L), I, H, L, C)
        assert e.kernel_name("gru_layer") == "gru_vec_kernel" and np.abs(out - ref).max() < GRU_TOL, seed


@pytest.mark.parametrize("stack", ["0", "1"])
def test_h64_with_192_inputs_falls_back_where_the_split_body_does_not_fit_lds(monkeypatch, stack):
    """H = 64 with K = 189..192 needs 164 KB for the eight-wave split body (h + x double buffers + three exchange buffers per
    chunk): such a first layer takes the plain kernel, with or without the stack launch (was: launch failure, round 4)."""
    from optistate_amd import Engine, RNN, flatten_state_dict
    from oracle import c_oracle as orc
    I, H, L, C, B, T = 192, 64, 3, 24, 40, 6
    monkeypatch.setenv("OS_GRU_STACK", stack)
    torch.manual_seed(41)
    m = RNN(I, H, L, C, torch.device("cpu"))
    e = Engine(0)
    e.load_gru(flatten_state_dict(m.state_dict(), L, "cuda"), I, H, L, C)
    x = torch.rand(B, T, I)
    out = e.gru_forward(x.cuda())
    out = (out[0] if isinstance(out, (tuple, list)) else out).cpu().numpy()
    ref, _, _ = orc.gru_forward(x.numpy(), orc.flatten_state_dict(m.state_dict(), L), I, H, L, C)
    assert np.abs(out - ref).max() < GRU_TOL


@pytest.mark.parametrize("B,T,expect", [(4, 12, "gru_vec_kernel"), (4, 13, "gru_wide_kernel"), (5, 9, "gru_wide_kernel"), (1, 48, "gru_vec_kernel"),
                                        (1, 49, "gru_wide_kernel"), (3, 16, "gru_vec_kernel"), (3, 17, "gru_wide_kernel"),
                                        (512, 3, "gru_wide_kernel"), (513, 3, "gru_stack_kernel"), (2048, 2, "gru_stack_kernel"), (2049, 2, "gru_layer"), (4096, 2, "gru_stack_kernel"), (4097, 2, "gru_layer")])
def test_small_batch_dispatch_boundaries(B, T, expect):
    """Either side of every dispatch boundary of os_gru_forward (B <= 4 and B T <= 48: gru_vec_kernel; up to 512 windows at four layers of 128: one
    gru_wide_kernel launch, four CUs per (layer, tile); (layer, tile) workgroups <= 256: one gru_stack_kernel launch; up to 128 tiles: as many layers per launch as fit -- 2,049: three + the fourth on its own, 4,096: two
    + two; beyond: a launch per layer) gives the float64 oracle's numbers, through the drop-in class."""
    from optistate_amd import RNN
    from oracle import c_oracle as orc
    I, H, L, C = 188, 128, 4, 24
    torch.manual_seed(43)
    m = RNN(I, H, L, C, torch.device("cuda")).to("cuda").eval()
    x = torch.rand(B, T, I)
    with torch.no_grad():
        out = m(x.cuda()).cpu().numpy()
    assert m._engine.kernel_name("gru_layer").startswith(expect), m._engine.kernel_name("gru_layer")
    pick = np.unique(np.r_[0:min(B, 40), max(B - 40, 0):B])
    ref, _, _ = orc.gru_forward(x.numpy()[pick], orc.flatten_state_dict(m.state_dict(), L), I, H, L, C)
    assert np.abs(out[pick] - ref).max() < GRU_TOL


@pytest.mark.parametrize("I,H,L,N,W", [(188, 128, 4, 8201, 10), (188, 128, 4, 4059, 10), (60, 64, 1, 300, 10), (60, 64, 3, 75, 7),
                                        (61, 128, 2, 40, 40), (188, 128, 4, 10, 10), (188, 128, 1, 33, 1)])
def test_window_stream_forward_equals_materialised_windows(I, H, L, N, W):
    """os_gru_forward_windows (gru/gru_test.py:138-140,174-191 without the window tensor; layer 0's x W_ih^T once per row) against
    the same model on the materialised windows (1e-6: the input half is summed in another order) and against the float64 oracle on
    a sample of windows; the reference's real shape at the bench size and at its own data-set size (~4,050 windows: the stack
    path behind the first layer), H = 64, an odd input width, one window covering the whole stream, window = 1."""
    from optistate_amd import Engine, RNN, flatten_state_dict
    from oracle import c_oracle as orc
    C = 24
    torch.manual_seed(41)
    m = RNN(I, H, L, C, torch.device("cpu"))
    rows = (torch.rand(N, I) * 2 - 1).cuda()
    eng = Engine(0)
    eng.load_gru(flatten_state_dict(m.state_dict(), L, "cuda"), I, H, L, C)
    out = eng.gru_forward_windows(rows, W)
    assert out.shape == (N - W + 1, C)
    assert eng.kernel_name("gru_layer") in ("gru_layer_split_kernel<GI>", "gru_layer_ahead_kernel", "gru_stack_kernel", "gru_wide_kernel", "gru_layer_split_kernel",
                                            "gru_layer_kernel<2,2>", "gru_layer_kernel<1,3>", "gru_vec_kernel", "gru_layer_stage_kernel")
    win = rows.unfold(0, W, 1).permute(0, 2, 1).contiguous()
    ref_gpu = eng.gru_forward(win)
    assert (out - ref_gpu).abs().max().item() < 1e-6
    pick = np.unique(np.concatenate([np.arange(0, min(48, N - W + 1)), np.arange(max(0, N - W + 1 - 48), N - W + 1)]))
    ref, _, _ = orc.gru_forward(win[pick].cpu().numpy(), orc.flatten_state_dict(m.state_dict(), L), I, H, L, C)
    assert np.abs(out[pick].cpu().numpy() - ref).max() < GRU_TOL


def test_window_stream_through_the_module_and_pipeline_g5():
    """RNN.forward_windows / pipeline.predict_rows on the reference-generated G5 weights: every window's output equals the module's
    own forward on that window (the reference's one-window call, gru_test.py:174-191)."""
    from optistate_amd import RNN
    from optistate_amd import pipeline as pl
    torch.manual_seed(1)
    m = RNN(188, 128, 4, 24, torch.device("cuda")).to("cuda").eval()
    rows = torch.rand(57, 188, device="cuda")
    with torch.no_grad():
        out = m.forward_windows(rows, 10)
        one_by_one = torch.cat([m(rows[i:i + 10][None]) for i in range(48)])
    assert (out - one_by_one).abs().max().item() < 1e-6
    mn, mx = torch.zeros(12, device="cuda") - 2.0, torch.zeros(12, device="cuda") + 3.0
    pred, above, below = pl.predict_rows(m, rows, 10, mn, mx)
    assert torch.allclose(pred, out[:, :12] * 5.0 - 2.0, atol=1e-6) and torch.allclose(above - pred, out[:, 12:] * 5.0, atol=1e-5)


def test_window_stream_shape_the_entry_point_declines_falls_back_to_materialised_windows():
    """hidden 64 with 191-192 inputs: the split body's LDS does not fit, os_gru_forward_windows returns -4 and RNN.forward_windows
    materialises (round 5, found by tools/fuzz_shapes.py: Engine.gru_windows_supported said yes and the call raised)."""
    from optistate_amd import RNN
    torch.manual_seed(2)
    m = RNN(192, 64, 1, 24, torch.device("cuda")).to("cuda").eval()
    rows = torch.rand(300, 192, device="cuda")
    with torch.no_grad():
        out = m.forward_windows(rows, 10)
        ref = m(rows.unfold(0, 10, 1).permute(0, 2, 1).contiguous())
    assert not m._engine.gru_windows_supported() and torch.equal(out, ref)
    m2 = RNN(190, 64, 1, 24, torch.device("cuda")).to("cuda").eval()
    with torch.no_grad():
        o2 = m2.forward_windows(rows[:, :190].contiguous(), 10)
        r2 = m2(rows[:, :190].contiguous().unfold(0, 10, 1).permute(0, 2, 1).contiguous())
    assert m2._engine.gru_windows_supported() and (o2 - r2).abs().max().item() < 1e-6
