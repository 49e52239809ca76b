"""GPU, BASELINE.json full size (B = 65,536, T = 100): size-independent properties of the hot path, plus an oracle spot check.

* independence: a trajectory's result does not depend on which other trajectories share the launch (a contiguous slice run
  alone and a permuted batch reproduce the full run bit for bit);
* the single-kernel fused path and the two-kernel path agree;
* a random sample of trajectories matches the float64 oracle within the parity bars."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

B, T = 65536, 100


@pytest.fixture(scope="module")
def setup():
    from optistate_amd import Engine, RNN, flatten_state_dict
    from optistate_amd.synth import synth_torch, Q_DEFAULT, R_DEFAULT
    eng = Engine(0)
    eng.set_noise(Q_DEFAULT, R_DEFAULT)
    d = synth_torch(B, T, "cuda", seed=2026)
    d["contact_p"] = eng.contact_soa_to_packed(d["contact"])
    torch.manual_seed(0)
    m = RNN(60, 64, 1, 24, torch.device("cpu"))
    eng.load_gru(flatten_state_dict(m.state_dict(), 1), 60, 64, 1, 24)
    mm = torch.stack([torch.full((60,), -30.0), torch.full((60,), 30.0)]).cuda()
    x, P = d["x0"].clone(), d["P0"].clone()
    full = eng.fused_run(d["p"], d["f"], d["dp"], d["imu"], d["contact_p"], d["accel"], mm, x, P)
    torch.cuda.synchronize()
    return eng, d, m, mm, full, x, P


def _slice(d, idx):
    s = {k: d[k][:, :, idx].contiguous() for k in ("p", "f", "dp", "imu", "accel")}
    s["contact_p"] = d["contact_p"][:, idx].contiguous()
    s["x0"], s["P0"] = d["x0"][:, idx].contiguous(), d["P0"][:, idx].contiguous()
    return s


def test_full_size_runs_clean(setup):
    eng, d, m, mm, full, x, P = setup
    assert int((full["status"] != 0).sum()) == 0
    assert torch.isfinite(full["x_out"]).all() and torch.isfinite(full["out"]).all()
    assert 0.0 < full["out"].min().item() and full["out"].max().item() < 1.0        # sigmoid head


def test_slice_and_permutation_independence(setup):
    eng, d, m, mm, full, x, P = setup
    idx = torch.arange(12288, 12288 + 4096, device="cuda")                          # 16 whole workgroups
    s = _slice(d, idx)
    xs, Ps = s["x0"].clone(), s["P0"].clone()
    eng.set_fused_tile(256)                                                         # the same kernel (tile shape) as the full batch
    try:
        r = eng.fused_run(s["p"], s["f"], s["dp"], s["imu"], s["contact_p"], s["accel"], mm, xs, Ps, two_kernel=False)
        assert torch.equal(r["x_out"], full["x_out"][:, :, idx])
        assert torch.equal(r["out"], full["out"][idx])
        assert torch.equal(Ps, P[:, idx])
        perm = torch.randperm(B, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))[:8192]
        s = _slice(d, perm)
        xs, Ps = s["x0"].clone(), s["P0"].clone()
        r = eng.fused_run(s["p"], s["f"], s["dp"], s["imu"], s["contact_p"], s["accel"], mm, xs, Ps, two_kernel=False)
    finally:
        eng.set_fused_tile(0)
    assert torch.equal(r["x_out"], full["x_out"][:, :, perm])                       # lane/wave placement does not matter
    assert (r["out"] - full["out"][perm]).abs().max().item() < 1e-6                 # rows move between MFMA row blocks


def test_single_kernel_equals_two_kernel_path(setup):
    eng, d, m, mm, full, x, P = setup
    x2, P2 = d["x0"].clone(), d["P0"].clone()
    two = eng.fused_run(d["p"], d["f"], d["dp"], d["imu"], d["contact_p"], d["accel"], mm, x2, P2, two_kernel=True)
    assert (two["x_out"] - full["x_out"]).abs().max().item() < 1e-6
    assert (two["out"] - full["out"]).abs().max().item() < 1e-5


def test_random_sample_matches_oracle(setup):
    from oracle import c_oracle as orc
    from optistate_amd.synth import Q_DEFAULT, R_DEFAULT
    eng, d, m, mm, full, x, P = setup
    idx = torch.randperm(B, generator=torch.Generator().manual_seed(3))[:192].cuda()
    g = lambda k: d[k][:, :, idx].permute(2, 0, 1).cpu().numpy()
    contact = d["contact"][:, :, idx].permute(2, 0, 1).cpu().numpy()
    ref = orc.kf_run_batch(g("p"), g("f"), g("dp"), g("imu"), contact, d["x0"][:, idx].t().cpu().numpy(),
                           np.tile(Q_DEFAULT, (192, 1, 1)), Q_DEFAULT, R_DEFAULT)
    xo = full["x_out"][:, :, idx].permute(2, 0, 1).cpu().numpy()
    assert np.abs(xo - ref["x"]).max() < 1e-4
    rows = np.concatenate([ref["x"], g("accel"), g("f"), ref["p_rot"], g("dp"), g("imu")], axis=2)
    ro, _, _ = orc.gru_forward((rows + 30.0) / 60.0, orc.flatten_state_dict(m.state_dict(), 1), 60, 64, 1, 24)
    assert np.abs(full["out"][idx].cpu().numpy() - ro).max() < 1e-4


# ------------------------------------------------------------------------------------------------------------------
# Round 6: the shards of a FIXED 65,536 batch (GPU g gets trajectories [g B/G, (g+1) B/G)) and the tile shapes that
# keep the chip full on them (os_fused_set_tile: 256 | 128 | 64 | 32 | 16 trajectories per CU)
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n_gpus,shard,tile", [(2, 1, 0), (4, 3, 0), (8, 5, 0), (2, 0, 128), (4, 1, 64), (8, 2, 32), (8, 7, 16), (16, 9, 16), (16, 11, 0)])
def test_shard_of_the_fixed_batch_equals_its_slice_of_the_full_run(setup, n_gpus, shard, tile):
    """Rank `shard` of `n_gpus` runs its contiguous slice alone, with the shape os_fused_run picks for that size (tile 0) or a pinned
    one: the filter is the same lane code in every shape (fp32 noise between instantiations: 1e-6 on x_out / x, 1e-9 on P), the gate
    sums of the smaller tiles run in another k order (1e-6 on the sigmoid outputs).  A random sample of the shard matches the
    float64 oracle within the parity bars."""
    from oracle import c_oracle as orc
    from optistate_amd.synth import Q_DEFAULT, R_DEFAULT
    eng, d, m, mm, full, x, P = setup
    n = B // n_gpus
    idx = torch.arange(shard * n, (shard + 1) * n, device="cuda")
    s = _slice(d, idx)
    xs, Ps = s["x0"].clone(), s["P0"].clone()
    eng.set_fused_tile(tile)
    try:
        r = eng.fused_run(s["p"], s["f"], s["dp"], s["imu"], s["contact_p"], s["accel"], mm, xs, Ps)
        name = eng.kernel_name("fused")
    finally:
        eng.set_fused_tile(0)
    if tile == 0:                                                   # what the library picks: the shape that fills the chip
        want = {2: "32 per wave", 4: "v3<1>", 8: "v3<2>", 16: "v3<4>"}[n_gpus]
        assert want in name, name
    # same filter arithmetic, but every tile shape is its own instantiation: hipcc contracts / orders a handful of fp32 operations
    # differently between them (measured 2e-7 on the state, 2e-12 on P), and the gate sums of the 16-wide tiles run in another k order
    assert (r["x_out"] - full["x_out"][:, :, idx]).abs().max().item() < 1e-6 and (xs - x[:, idx]).abs().max().item() < 1e-6
    assert (Ps - P[:, idx]).abs().max().item() < 1e-9
    assert int((r["status"] != 0).sum()) == 0
    assert (r["out"] - full["out"][idx]).abs().max().item() < 1e-6
    pick = torch.randperm(n, generator=torch.Generator().manual_seed(n_gpus * 100 + shard))[:64].cuda()
    gi = idx[pick]
    g = lambda k: d[k][:, :, gi].permute(2, 0, 1).cpu().numpy()
    contact = d["contact"][:, :, gi].permute(2, 0, 1).cpu().numpy()
    ref = orc.kf_run_batch(g("p"), g("f"), g("dp"), g("imu"), contact, d["x0"][:, gi].t().cpu().numpy(),
                           np.tile(Q_DEFAULT, (64, 1, 1)), Q_DEFAULT, R_DEFAULT)
    assert np.abs(r["x_out"][:, :, pick].permute(2, 0, 1).cpu().numpy() - ref["x"]).max() < 1e-4
    rows = np.concatenate([ref["x"], g("accel"), g("f"), ref["p_rot"], g("dp"), g("imu")], axis=2)
    ro, _, _ = orc.gru_forward((rows + 30.0) / 60.0, orc.flatten_state_dict(m.state_dict(), 1), 60, 64, 1, 24)
    assert np.abs(r["out"][pick].cpu().numpy() - ro).max() < 1e-5


@pytest.mark.parametrize("tile", [128, 64, 32, 16])
def test_tile_shapes_on_a_ragged_batch_and_run_to_run_determinism(setup, tile):
    """A batch that ends inside a tile (shadow lanes, a partly dead split pair) and is smaller than one round; two runs of the same
    launch are bit-identical (the kernels' inline-assembly MFMA / ds_read interleave has no compiler hazard cover)."""
    eng, d, m, mm, full, x, P = setup
    n = 4096 + 37
    idx = torch.arange(20000, 20000 + n, device="cuda")
    s = _slice(d, idx)
    eng.set_fused_tile(tile)
    try:
        outs = []
        for _ in range(2):
            xs, Ps = s["x0"].clone(), s["P0"].clone()
            r = eng.fused_run(s["p"], s["f"], s["dp"], s["imu"], s["contact_p"], s["accel"], mm, xs, Ps)
            outs.append((r, xs, Ps))
    finally:
        eng.set_fused_tile(0)
    (r, xs, Ps), (r2, xs2, Ps2) = outs
    assert torch.equal(r["out"], r2["out"]) and torch.equal(r["x_out"], r2["x_out"]) and torch.equal(Ps, Ps2) and torch.equal(xs, xs2)
    assert (r["x_out"] - full["x_out"][:, :, idx]).abs().max().item() < 1e-6 and (Ps - P[:, idx]).abs().max().item() < 1e-9
    assert (xs - x[:, idx]).abs().max().item() < 1e-6 and int((r["status"] != 0).sum()) == 0
    assert (r["out"] - full["out"][idx]).abs().max().item() < 1e-6
    # within ONE shape a trajectory's result does not depend on its neighbours: a sub-slice run alone reproduces it bit for bit
    sub = torch.arange(512, 512 + 1024 + 5, device="cuda")
    s2 = {k: (v[..., sub].contiguous()) for k, v in s.items()}
    eng.set_fused_tile(tile)
    try:
        xs3, Ps3 = s2["x0"].clone(), s2["P0"].clone()
        r3 = eng.fused_run(s2["p"], s2["f"], s2["dp"], s2["imu"], s2["contact_p"], s2["accel"], mm, xs3, Ps3)
    finally:
        eng.set_fused_tile(0)
    assert torch.equal(r3["x_out"], r["x_out"][:, :, sub]) and torch.equal(Ps3, Ps[:, sub]) and torch.equal(r3["out"], r["out"][sub])


# ------------------------------------------------------------------------------------------------------------------
# Hostile inputs at full size: everything the nominal bench data never exercises (VERDICT r2 "what's weak" 2, 3)
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("noise", ["default", "fitted"])
def test_hostile_contacts_and_unwrapped_yaw_through_every_kernel_family(noise):
    """0...4 stance legs per step (standing and flight stretches), yaw unwrapping past +-pi at up to 8 rad/s, roll / pitch
    to 1 rad, exact 0 / k pi/2 attitude starts (the int64-truncation predicates): 65,536 x 100 through
    fused_kf_gru_kernel_v2 and kf_run_sym_kernel, a 4,096-slice through kf_run_rows2_kernel, each against the float64
    oracle (16,384 random trajectories x all 100 steps + the whole exact-start block), under both reference noise sets."""
    from optistate_amd import Engine, RNN, flatten_state_dict
    from optistate_amd.synth import synth_torch, NOISE_SETS, _hostile_exact_starts
    from oracle import c_oracle as orc
    Q, R = NOISE_SETS[noise]
    eng = Engine(0)
    eng.set_noise(Q, R)
    d = synth_torch(B, T, "cuda", seed=77, hostile=True)
    cp = eng.contact_soa_to_packed(d["contact"])
    stance = d["contact"].sum(dim=1)
    assert all(int((stance == k).sum()) > B for k in range(5))                     # every count 0..4 really occurs, often
    assert d["imu"][:, 2].abs().max().item() > 6.0                                  # yaw leaves (-pi, pi]
    P0 = torch.tensor(np.asarray(Q, dtype=np.float32).reshape(144, 1), device="cuda").repeat(1, B).contiguous()
    torch.manual_seed(0)
    m = RNN(60, 64, 1, 24, torch.device("cpu"))
    eng.load_gru(flatten_state_dict(m.state_dict(), 1), 60, 64, 1, 24)
    mm = torch.stack([torch.full((60,), -60.0), torch.full((60,), 60.0)]).cuda()    # fitted-noise omega_z reaches +-55
    x, P = d["x0"].clone(), P0.clone()
    fz = eng.fused_run(d["p"], d["f"], d["dp"], d["imu"], cp, d["accel"], mm, x, P, two_kernel=False)
    assert eng.kernel_name("fused").startswith("fused_kf_gru_kernel_v2")
    x, P = d["x0"].clone(), P0.clone()
    sy = eng.kf_run(d["p"], d["f"], d["dp"], d["imu"], cp, x, P)
    assert eng.kernel_name("kf") == "kf_run_sym_kernel"
    sl = torch.arange(2048, 2048 + 4096, device="cuda")                             # covers exact-start trajectories too
    g4 = lambda k: d[k][:, :, sl].contiguous()
    xs, Ps = d["x0"][:, sl].contiguous(), P0[:, sl].contiguous()
    rw = eng.kf_run(g4("p"), g4("f"), g4("dp"), g4("imu"), cp[:, sl].contiguous(), xs, Ps)
    assert eng.kernel_name("kf") == "kf_run_rows2_kernel"
    torch.cuda.synchronize()
    n_exact = _hostile_exact_starts(B)[-1][1]
    for r in (fz, sy, rw):
        assert int(((r["status"] & ~16) != 0).sum()) == 0 and torch.isfinite(r["x_out"]).all()
    # status bit 4 (int64-truncation knife edge, include/optistate_hip.h): only the exact k pi/2 starts may carry it (an entry
    # of R within 2^-40 of +-1 at t = 0; both filters hold the identical exact attitude there, so the decisions agree)
    for r, off in ((fz, 0), (sy, 0), (rw, 2048)):
        flagged = torch.nonzero(r["status"] & 16).flatten() + off
        assert flagged.numel() == 0 or (int(flagged.min()) >= n_exact // 5 and int(flagged.max()) < n_exact), flagged[:8]
    # the two lane kernels run the same arithmetic on the filter state (omega_z reaches +-55 under the fitted set: 1 ulp = 4e-6)
    assert (fz["x_out"] - sy["x_out"]).abs().max().item() < 5e-5
    pick = torch.cat([torch.arange(n_exact), n_exact + torch.randperm(B - n_exact, generator=torch.Generator().manual_seed(9))[:16384 - 4096],
                      ]).unique().cuda()
    orc.set_threads(orc.max_threads())
    g = lambda k: d[k][:, :, pick].permute(2, 0, 1).double().cpu().numpy()
    n = int(pick.numel())
    ref = orc.kf_run_batch(g("p"), g("f"), g("dp"), g("imu"), d["contact"][:, :, pick].permute(2, 0, 1).contiguous().cpu().numpy(),
                           d["x0"][:, pick].t().double().cpu().numpy(), np.tile(Q, (n, 1, 1)), Q, R)
    orc.set_threads(1)
    assert int(ref["status"].sum()) == 0
    for name, r in (("fused_v2", fz), ("sym", sy)):
        err = np.abs(r["x_out"][:, :, pick].permute(2, 0, 1).cpu().numpy() - ref["x"])
        assert err.max() < 1e-4, (name, float(err.max()), np.unravel_index(int(err.argmax()), err.shape))
    # rows kernel: its slice against the oracle rows of the same trajectories
    pos = {int(v): i for i, v in enumerate(pick.cpu().tolist())}
    both = [v for v in sl.cpu().tolist() if v in pos]
    assert len(both) > 1500
    a = rw["x_out"][:, :, torch.tensor([v - 2048 for v in both], device="cuda")].permute(2, 0, 1).cpu().numpy()
    b = ref["x"][[pos[v] for v in both]]
    assert np.abs(a - b).max() < 1e-4, float(np.abs(a - b).max())
    # GRU head of the fused kernel on the oracle's feature rows
    rows = np.concatenate([ref["x"], g("accel"), g("f"), ref["p_rot"], g("dp"), g("imu")], axis=2)
    orc.set_threads(orc.max_threads())
    ro, _, _ = orc.gru_forward((rows + 60.0) / 120.0, orc.flatten_state_dict(m.state_dict(), 1), 60, 64, 1, 24)
    orc.set_threads(1)
    assert np.abs(fz["out"][pick].cpu().numpy() - ro).max() < 1e-4


@pytest.mark.parametrize("noise", ["default", "fitted"])
def test_gimbal_lock_knife_edge_is_flagged(noise):
    """The block the generator used to steer around: pitch = float32(pi/2) with roll = yaw = 0.  There
    R[1][1] = cos(yaw - roll) sits within one rounding of 1 in float64, so whether the reference's int64 A[0:3,6:9]
    (misc/force_controller.py:248-251,271) picks up a 1 is decided by the last bits of ITS float64 state.  Property:
    every trajectory whose state error exceeds the 1e-4 bar carries status bit 4, through every kernel family; where a
    decision did flip, the error at its first appearance is one dt * omega integration step (times |I - K H| <= 2)."""
    from optistate_amd import Engine
    from optistate_amd.synth import synth_torch, NOISE_SETS, _hostile_exact_starts
    from oracle import c_oracle as orc
    Q_DEFAULT, R_DEFAULT = NOISE_SETS[noise]          # (names kept: the set under test)
    Bg, Tg = 65536, 100                               # seed 77 + the fitted set is the case round 3 measured a flip on
    eng = Engine(0)
    eng.set_noise(Q_DEFAULT, R_DEFAULT)
    d = synth_torch(Bg, Tg, "cuda", seed=77, hostile=True, gimbal_lock=True)
    blocks = _hostile_exact_starts(Bg, gimbal_lock=True)
    lo, hi = blocks[-1][0], blocks[-1][1]
    assert blocks[-1][2][0] == 0.0 and blocks[-1][2][2] == 0.0
    cp = eng.contact_soa_to_packed(d["contact"])
    P0 = torch.tensor(np.asarray(Q_DEFAULT, dtype=np.float32).reshape(144, 1), device="cuda").repeat(1, Bg).contiguous()
    torch.manual_seed(0)
    from optistate_amd import RNN, flatten_state_dict
    m = RNN(60, 64, 1, 24, torch.device("cpu"))
    eng.load_gru(flatten_state_dict(m.state_dict(), 1), 60, 64, 1, 24)
    mm = torch.stack([torch.full((60,), -60.0), torch.full((60,), 60.0)]).cuda()
    x, P = d["x0"].clone(), P0.clone()
    fz = eng.fused_run(d["p"], d["f"], d["dp"], d["imu"], cp, d["accel"], mm, x, P, two_kernel=False)
    x, P = d["x0"].clone(), P0.clone()
    sy = eng.kf_run(d["p"], d["f"], d["dp"], d["imu"], cp, x, P)
    assert eng.kernel_name("kf") == "kf_run_sym_kernel"
    sl = torch.arange(lo - 1024, hi + 1024, device="cuda")                            # the gimbal block + neighbours: rows kernel
    g4 = lambda k: d[k][:, :, sl].contiguous()
    xs, Ps = d["x0"][:, sl].contiguous(), P0[:, sl].contiguous()
    rw = eng.kf_run(g4("p"), g4("f"), g4("dp"), g4("imu"), cp[:, sl].contiguous(), xs, Ps)
    assert eng.kernel_name("kf") == "kf_run_rows2_kernel"
    torch.cuda.synchronize()
    pick = torch.cat([torch.arange(0, hi), hi + torch.randperm(Bg - hi, generator=torch.Generator().manual_seed(10))[:4096]]).cuda()
    orc.set_threads(orc.max_threads())
    g = lambda k: d[k][:, :, pick].permute(2, 0, 1).double().cpu().numpy()
    n = int(pick.numel())
    ref = orc.kf_run_batch(g("p"), g("f"), g("dp"), g("imu"), d["contact"][:, :, pick].permute(2, 0, 1).contiguous().cpu().numpy(),
                           d["x0"][:, pick].t().double().cpu().numpy(), np.tile(Q_DEFAULT, (n, 1, 1)), Q_DEFAULT, R_DEFAULT)
    orc.set_threads(1)
    pk = pick.cpu().numpy()
    report = {}
    for name, r, off in (("fused_v2", fz, 0), ("sym", sy, 0), ("rows2", rw, lo - 1024)):
        st = r["status"].cpu().numpy()
        assert int(((st & ~16) != 0).sum()) == 0
        sel = np.nonzero((pk >= off) & (pk < off + st.shape[0]))[0]
        xg = r["x_out"][:, :, torch.as_tensor(pk[sel] - off, device="cuda")].permute(2, 0, 1).cpu().numpy()
        err = np.abs(xg - ref["x"][sel])
        worst = err.reshape(len(sel), -1).max(1)
        flagged = (st[pk[sel] - off] & 16) != 0
        # (a) the property: above the bar => flagged
        assert np.all(flagged[worst >= 1e-4]), (name, np.nonzero((worst >= 1e-4) & ~flagged)[0][:8])
        assert worst[~flagged].max() < 1e-4
        # (b) the gimbal block is where the flags are (plus the exact k pi/2 starts, flagged at t = 0 only)
        in_block = (pk[sel] >= lo) & (pk[sel] < hi)
        assert flagged[in_block].all() and not flagged[pk[sel] >= hi].any()
        # (c) a flipped decision costs one integration step: dt * sum |omega_prior|, through |I - K H| <= 2
        bad = np.nonzero(worst >= 1e-4)[0]
        for i in bad:
            tstar = int(np.argmax(err[i].max(1) >= 1e-4))
            w = np.abs(ref["x_prior"][sel[i], tstar, 6:9]).sum() if tstar > 0 else np.abs(ref["x_prior"][sel[i], 0, 6:9]).sum()
            assert err[i, tstar].max() <= 2 * 0.01 * max(w, np.abs(ref["x"][sel[i], max(tstar - 1, 0), 6:9]).sum()) + 1e-4, (name, int(pk[sel[i]]), tstar)
        report[name] = (int(flagged.sum()), int(len(bad)), float(worst.max()))
    print(f"gimbal block [{noise}]: (flagged, above the bar, worst linf) per kernel:", report)
    # the block is not vacuous: the reference really does integrate through the knife edge on some of these trajectories
    # (theta_y moves away from pi/2 by dt * omega steps in the oracle's prior) -- whether the GPU agrees is what (a) covers


# ------------------------------------------------------------------------------------------------------------------
# BASELINE.json configs[1]: Kalman only, B = 4096, T = 1000 (default dispatch = the 16-lanes-per-trajectory kernel)
# ------------------------------------------------------------------------------------------------------------------
def test_config2_kf_4096x1000_oracle_sample_and_slice_independence():
    from optistate_amd import Engine
    from optistate_amd.synth import synth_torch, Q_DEFAULT, R_DEFAULT
    from oracle import c_oracle as orc
    B2, T2 = 4096, 1000
    eng = Engine(0)
    eng.set_noise(Q_DEFAULT, R_DEFAULT)
    d = synth_torch(B2, T2, "cuda", seed=404)
    cp = eng.contact_soa_to_packed(d["contact"])
    x, P = d["x0"].clone(), d["P0"].clone()
    full = eng.kf_run(d["p"], d["f"], d["dp"], d["imu"], cp, x, P)
    torch.cuda.synchronize()
    assert eng.kernel_name("kf") == "kf_run_rows2_kernel"
    assert int((full["status"] != 0).sum()) == 0 and torch.isfinite(full["x_out"]).all()
    # a contiguous slice alone reproduces the full run bit for bit (same kernel: B <= 8,192)
    idx = torch.arange(1024, 1024 + 512, device="cuda")
    sl = lambda k: d[k][:, :, idx].contiguous()
    xs, Ps = d["x0"][:, idx].contiguous(), d["P0"][:, idx].contiguous()
    r = eng.kf_run(sl("p"), sl("f"), sl("dp"), sl("imu"), cp[:, idx].contiguous(), xs, Ps)
    assert torch.equal(r["x_out"], full["x_out"][:, :, idx]) and torch.equal(Ps, P[:, idx])
    # 64 random trajectories x 1000 steps against the float64 oracle
    pick = torch.randperm(B2, generator=torch.Generator().manual_seed(5))[:64].cuda()
    g = lambda k: d[k][:, :, pick].permute(2, 0, 1).cpu().numpy()
    ref = orc.kf_run_batch(g("p"), g("f"), g("dp"), g("imu"), d["contact"][:, :, pick].permute(2, 0, 1).cpu().numpy(),
                           d["x0"][:, pick].t().cpu().numpy(), np.tile(Q_DEFAULT, (64, 1, 1)), Q_DEFAULT, R_DEFAULT, aux=False)
    err = np.abs(full["x_out"][:, :, pick].permute(2, 0, 1).cpu().numpy() - ref["x"])
    assert err.max() < 1e-4, err.max()
    Pf = P[:, pick].t().cpu().numpy().reshape(64, 12, 12)
    assert np.abs(Pf - ref["P_final"]).max() < 1e-3 * np.abs(ref["P_final"]).max()


# ------------------------------------------------------------------------------------------------------------------
# BASELINE.json configs[3]: the gru_train.py step at 8192 windows x 10, RNN(188,128,4,24), forward + backward
# ------------------------------------------------------------------------------------------------------------------
def test_small_batch_kernel_steps_aside_when_the_streams_outgrow_its_32_bit_offsets():
    """kf_run_rows2_kernel carries a step's position in a 32-bit buffer offset (T * 48 B bytes): beyond that the dispatcher must
    take the lane-per-trajectory kernel (per-step descriptors), and a causal filter's first steps must not depend on which."""
    from optistate_amd import Engine
    from optistate_amd.synth import synth_torch, Q_DEFAULT, R_DEFAULT
    eng = Engine(0); eng.set_noise(Q_DEFAULT, R_DEFAULT)
    Bs, Tl, Ts = 8000, 11200, 200                      # 8000 <= the rows/lane switch; 11200 * 48 * 8000 = 4.30e9 >= 2^32
    assert Tl * 48 * Bs >= 2 ** 32 and Ts * 48 * Bs < 2 ** 32
    d = synth_torch(Bs, Tl, "cuda", seed=41)
    cp = eng.contact_soa_to_packed(d["contact"])
    x, P = d["x0"].clone(), d["P0"].clone()
    r = eng.kf_run(d["p"], d["f"], d["dp"], d["imu"], cp, x, P)
    assert eng.kernel_name("kf") == "kf_run_sym_kernel"
    assert int(r["status"].abs().sum()) == 0 and bool(torch.isfinite(r["x_out"][-1]).all())
    x2, P2 = d["x0"].clone(), d["P0"].clone()
    r2 = eng.kf_run(d["p"][:Ts].contiguous(), d["f"][:Ts].contiguous(), d["dp"][:Ts].contiguous(), d["imu"][:Ts].contiguous(),
                    cp[:Ts].contiguous(), x2, P2)
    assert eng.kernel_name("kf") == "kf_run_rows2_kernel"
    assert float((r["x_out"][:Ts] - r2["x_out"]).abs().max()) < 5e-5          # two kernels, same filter (cross-kernel bar)


def test_config4_train_step_8192_windows_vs_float64_autograd():
    from optistate_amd import RNN
    from test_gpu_train import torch_reference_grads
    dims, Bt, Tt = (188, 128, 4, 24), 8192, 10
    torch.manual_seed(11)
    m = RNN(*dims, torch.device("cuda")).to("cuda")
    x, y = torch.rand(Bt, Tt, dims[0]), torch.rand(Bt, 12)
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    # the float64 CPU reference of the whole batch: a few tens of threads, and the setting is RESTORED -- left at every hardware
    # thread of the GPU box's host (256) it made each later small torch reference ~100x slower (a third of the GPU suite's time)
    prev_threads = torch.get_num_threads()
    torch.set_num_threads(max(1, min(32, len(__import__("os").sched_getaffinity(0)))))
    try:
        ref_out, _, ref_loss, ref_g, _ = torch_reference_grads(sd, dims, x, y)
    finally:
        torch.set_num_threads(prev_threads)
    out = m(x.cuda())
    assert np.abs(out.detach().cpu().numpy() - ref_out.numpy()).max() < 1e-5
    tgt = torch.cat([y.cuda(), (out[:, :12].detach() - y.cuda()).abs()], dim=1)
    loss = torch.nn.functional.mse_loss(out, tgt)
    loss.backward()
    assert abs(loss.item() - ref_loss) < 1e-6
    for k, p in m.named_parameters():
        r = ref_g[k]
        scale = max(r.abs().max().item(), 1e-8)
        assert (p.grad.cpu().double() - r).abs().max().item() < 2e-4 * scale + 1e-9, k


# ------------------------------------------------------------------------------------------------------------------
# BASELINE.json configs[4]: 1024 depth frames through the ViT encoder (parity unpinned: own float64 restatement)
# ------------------------------------------------------------------------------------------------------------------
def test_config5_vit_1024_frames_slice_independence_and_oracle_sample():
    from optistate_amd.transformer_model import Transformer_Autoencoder
    from oracle import vit_oracle
    torch.manual_seed(0)
    vit = Transformer_Autoencoder().to("cuda")
    with torch.no_grad():
        for p in vit.parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    frames = torch.rand(1024, 1, 224, 224, generator=torch.Generator().manual_seed(2)).cuda()
    lat = vit.forward_encoder(frames)
    assert lat.shape == (1024, 1, 128) and torch.isfinite(lat).all()
    assert 0.0 < lat.min().item() and lat.max().item() < 1.0
    sub = vit.forward_encoder(frames[100:132].contiguous())
    assert (sub - lat[100:132]).abs().max().item() < 1e-6                  # a frame's latent does not depend on its batch
    pick = torch.randperm(1024, generator=torch.Generator().manual_seed(9))[:32]
    sd = {k: v.detach().cpu().numpy() for k, v in vit.state_dict().items()}
    ref = vit_oracle.encode(frames[pick.cuda(), 0].cpu().numpy(), sd)
    assert np.abs(lat[pick.cuda(), 0].cpu().numpy() - ref).max() < 2e-5
