"""GPU: the OPT-IN split-bf16 gate GEMM of the fused kernel (OS_FUSED_SPLIT_BF16: three bf16 terms per fp32 operand, six bf16
MFMAs per product block; OS_FUSED_SPLIT_BF16_2: two terms, three MFMAs; fp32 accumulate).  Reported beside the exact-fp32
kernel, never instead of it, and held to the same bars: GRU head l-inf < 1e-5 vs the float64 oracle on the reference-generated
G5 weights, fused chain < 1e-4.  Also: run-to-run bit-identity (the kernels schedule inline-asm MFMAs by hand; a register
hazard there shows up as results that change between launches)."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _run(B, T, seed, sd=None, split=True, flat_np=None):
    from optistate_amd import Engine, RNN, flatten_state_dict
    from optistate_amd.synth import synth_numpy, Q_FITTED, R_FITTED
    from oracle import c_oracle as orc
    d = synth_numpy(B, T, seed=seed)
    ref = orc.kf_run_batch(d["p"], d["f"], d["dp"], d["imu"], d["contact"], d["x0"], np.tile(Q_FITTED, (B, 1, 1)), Q_FITTED, R_FITTED)
    rows = np.concatenate([ref["x"], d["accel"].astype(np.float64), d["f"].astype(np.float64), ref["p_rot"],
                           d["dp"].astype(np.float64), d["imu"].astype(np.float64)], axis=2)
    mn, mx = rows.reshape(-1, 60).min(0), rows.reshape(-1, 60).max(0)
    m = RNN(60, 64, 1, 24, torch.device("cpu"))
    if sd is not None:
        m.load_state_dict(sd)
    ref_out, _, _ = orc.gru_forward((rows - mn) / (mx - mn), orc.flatten_state_dict(m.state_dict(), 1), 60, 64, 1, 24)
    eng = Engine(0)
    eng.set_noise(Q_FITTED, R_FITTED)
    eng.load_gru(flatten_state_dict(m.state_dict(), 1), 60, 64, 1, 24)
    s = {k: eng.pack(torch.as_tensor(d[k])) for k in ("p", "f", "dp", "imu", "accel")}
    c = eng.pack_contact(torch.as_tensor(d["contact"]))
    x = torch.as_tensor(d["x0"].T.copy()).cuda()
    P = torch.as_tensor(np.tile(Q_FITTED.astype(np.float32).reshape(144, 1), (1, B))).cuda()
    mm = torch.as_tensor(np.stack([mn, mx]).astype(np.float32)).cuda()
    r = eng.fused_run(s["p"], s["f"], s["dp"], s["imu"], c, s["accel"], mm, x, P, two_kernel=False, split_bf16=split)
    torch.cuda.synchronize()
    name = eng.kernel_name("fused")
    e_state = float(np.abs(eng.unpack(r["x_out"]).cpu().numpy() - ref["x"]).max())
    e_out = float(np.abs(r["out"].cpu().numpy() - ref_out).max())
    return name, e_state, e_out, int(r["status"].abs().sum())


@pytest.mark.parametrize("terms", [3, 2])
def test_split_bf16_meets_the_fp32_bars_with_the_g5_reference_weights(terms):
    g = load_golden("gru_g5_small.npz")
    sd = {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w:")}          # the reference RNN(60,64,1,24), seed 1
    name, e_state, e_out, st = _run(700, 100, seed=3, sd=sd, split=terms)
    assert name == f"fused_kf_gru_bf16_kernel<{terms}>"
    assert st == 0 and e_state < 1e-4 and e_out < 1e-5, (e_state, e_out)


@pytest.mark.parametrize("split", [False, 3, 2])
def test_fused_kernels_are_bit_deterministic_at_full_size(split):
    from optistate_amd import Engine, RNN, flatten_state_dict
    from optistate_amd.synth import synth_torch, Q_DEFAULT, R_DEFAULT
    B, T = 65536, 40
    d = synth_torch(B, T, "cuda", seed=1)
    c = Engine.contact_soa_to_packed(d["contact"])
    torch.manual_seed(0)
    m = RNN(60, 64, 1, 24, torch.device("cpu"))
    e = Engine(0); e.set_noise(Q_DEFAULT, R_DEFAULT); e.load_gru(flatten_state_dict(m.state_dict(), 1, "cuda"), 60, 64, 1, 24)
    mm = torch.stack([torch.full((60,), -30.0), torch.full((60,), 30.0)]).cuda()
    outs = []
    for _ in range(4):
        x, P = d["x0"].clone(), d["P0"].clone()
        r = e.fused_run(d["p"], d["f"], d["dp"], d["imu"], c, d["accel"], mm, x, P, two_kernel=False, split_bf16=split)
        torch.cuda.synchronize()
        outs.append((r["out"].clone(), r["x_out"].clone()))
    for o, xo in outs[1:]:
        assert torch.equal(o, outs[0][0]) and torch.equal(xo, outs[0][1])
    assert torch.isfinite(outs[0][0]).all()


def test_split_bf16_ragged_batch_and_default_is_still_fp32():
    torch.manual_seed(5)
    name, e_state, e_out, st = _run(333, 37, seed=8)
    assert name == "fused_kf_gru_bf16_kernel<3>" and st == 0 and e_state < 1e-4 and e_out < 1e-5
    torch.manual_seed(5)
    name, e_state, e_out, st = _run(333, 37, seed=8, split=False)
    assert name.startswith("fused_kf_gru_kernel_v") and "bf16" not in name and e_out < 1e-5      # the flag is opt-in: the default never takes the bf16 path (an fp32 tile shape picked for the batch)


def test_split_bf16_refuses_other_shapes():
    from optistate_amd import Engine, RNN, flatten_state_dict
    from optistate_amd.synth import synth_torch, Q_DEFAULT, R_DEFAULT
    eng = Engine(0); eng.set_noise(Q_DEFAULT, R_DEFAULT)
    m = RNN(60, 128, 2, 24, torch.device("cpu"))
    eng.load_gru(flatten_state_dict(m.state_dict(), 2), 60, 128, 2, 24)
    d = synth_torch(64, 3, "cuda", seed=0)
    mm = torch.stack([torch.full((60,), -30.0), torch.full((60,), 30.0)]).cuda()
    with pytest.raises(RuntimeError, match="OS_FUSED_SPLIT_BF16"):
        eng.fused_run(d["p"], d["f"], d["dp"], d["imu"], eng.contact_soa_to_packed(d["contact"]), d["accel"], mm, d["x0"].clone(),
                      d["P0"].clone(), split_bf16=True)
