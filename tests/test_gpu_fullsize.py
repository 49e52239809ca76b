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
    r = eng.fused_run(s["p"], s["f"], s["dp"], s["imu"], s["contact_p"], s["accel"], mm, xs, Ps, two_kernel=False)   # same (single) kernel as the full batch
    assert torch.equal(r["x_out"], full["x_out"][:, :, idx])
    assert torch.equal(r["out"], full["out"][idx])
    assert torch.equal(Ps, P[:, idx])
    perm = torch.randperm(B, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))[:8192]
    s = _slice(d, perm)
    xs, Ps = s["x0"].clone(), s["P0"].clone()
    r = eng.fused_run(s["p"], s["f"], s["dp"], s["imu"], s["contact_p"], s["accel"], mm, xs, Ps, two_kernel=False)   # same (single) kernel as the full batch
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
