"""GPU: ViT-encoder latent (optional row A10 / config 5).  timm 0.3.2 and the trained weights are absent: the HIP path is compared
with this repo's float64 restatement (oracle/vit_oracle.py), which is pinned to the reference-owned GLUE (forward_encoder,
initialize_weights) by G11 -- the reference's own class run with a timm stand-in (tests/test_vit_glue_golden.py) -- and to an
independent implementation of the transformer block by G14 (the reference class around Hugging Face's ViTLayer)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_encoder_latent_matches_float64_restatement():
    from optistate_amd.transformer_model import Transformer_Autoencoder
    from oracle import vit_oracle
    torch.manual_seed(0)
    m = Transformer_Autoencoder().to("cuda")
    with torch.no_grad():            # make biases / LayerNorm affine non-trivial so every term is exercised
        for p in m.parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    x = torch.rand(5, 1, 224, 224)
    lat = m.forward_encoder(x.cuda())
    assert lat.shape == (5, 1, 128)
    sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    ref = vit_oracle.encode(x[:, 0].numpy(), sd)
    err = np.abs(lat.cpu().numpy()[:, 0] - ref).max()
    assert err < 2e-5, err
    assert 0.0 < lat.min().item() and lat.max().item() < 1.0


def test_hip_encoder_matches_the_reference_glue_fixture_g11():
    """G11 ("blocks = stand-in, glue = reference"): the reference's Transformer_Autoencoder.forward_encoder
    (transformer/transformer_model.py:113-135) on 8 seeded frames under seeded weights; the HIP encoder gets the same weights
    through the reference's state_dict keys (load_state_dict) and must reproduce the latents."""
    from optistate_amd.transformer_model import Transformer_Autoencoder
    from conftest import load_golden
    from golden_recipes import g11_encoder_state, g11_frames
    g = load_golden("vit_g11_glue.npz")
    m = Transformer_Autoencoder()
    res = m.load_state_dict({k: torch.from_numpy(v) for k, v in g11_encoder_state().items()}, strict=False)
    assert res.missing_keys == ["pos_embed"] and not res.unexpected_keys          # the fixed table is the constructor's (= G10)
    m = m.to("cuda")
    lat = m.forward_encoder(torch.from_numpy(g11_frames()).unsqueeze(1).cuda())[:, 0].cpu().numpy()
    assert np.abs(lat - g["latent_f64"]).max() < 2e-5, np.abs(lat - g["latent_f64"]).max()
    assert np.abs(lat - g["latent_f32"]).max() < 2e-5
    # G14: the same frames and weights through the reference's glue with Hugging Face transformers' ViTLayer / ViTPatchEmbeddings as
    # the blocks (tools/hf_vit_blocks.py): nothing of ours between the reference's code and these latents
    h = load_golden("vit_g14_hf_blocks.npz")
    assert np.abs(lat - h["latent_f64"]).max() < 2e-5 and np.abs(lat - h["latent_f32"]).max() < 2e-5


def test_pos_embed_table_and_state_dict_keys():
    from optistate_amd.transformer_model import Transformer_Autoencoder
    from conftest import load_golden
    m = Transformer_Autoencoder().to("cuda")
    # G10: the reference's own transformer/pos_embed.py output (tools/gen_golden.py), not a twin of the product's formula
    g10 = load_golden("vit_g10_pos_embed.npz")["enc_128_14"]
    assert torch.equal(m.pos_embed[0].cpu(), torch.from_numpy(g10).float())
    keys = set(m.state_dict().keys())
    for k in ("patch_embed.proj.weight", "cls_token", "pos_embed", "blocks.0.attn.qkv.weight", "blocks.2.mlp.fc2.bias",
              "norm.weight"):
        assert k in keys
    assert sum(p.numel() for n, p in m.named_parameters() if n != "pos_embed") + 197 * 128 == \
        128 * 256 + 128 + 128 + 197 * 128 + 3 * (2 * 128 + 384 * 128 + 384 + 128 * 128 + 128 + 2 * 128 + 512 * 128 + 512 + 128 * 512 + 128) + 256


def test_config5_pipeline_latent_feeds_gru():
    """depth frames -> latent -> appended to the 60 Kalman features -> GRU(188,128,4,24) (gru/gru_test.py:119-136)."""
    from optistate_amd import Engine, RNN, flatten_state_dict
    from optistate_amd.transformer_model import Transformer_Autoencoder
    from optistate_amd.synth import synth_numpy, Q_FITTED, R_FITTED
    from oracle import c_oracle as orc, vit_oracle
    B, T = 6, 5
    torch.manual_seed(1)
    vit = Transformer_Autoencoder().to("cuda")
    frames = torch.rand(B * T, 1, 224, 224)
    lat = vit.forward_encoder(frames.cuda()).reshape(B, T, 128)
    ref_lat = vit_oracle.encode(frames[:, 0].numpy(), {k: v.cpu().numpy() for k, v in vit.state_dict().items()}).reshape(B, T, 128)
    d = synth_numpy(B, T, seed=31)
    ref = orc.kf_run_batch(d["p"], d["f"], d["dp"], d["imu"], d["contact"], d["x0"], np.tile(Q_FITTED, (B, 1, 1)), Q_FITTED, R_FITTED)
    rows = np.concatenate([ref["x"], d["accel"], d["f"], ref["p_rot"], d["dp"], d["imu"]], axis=2)
    mn, mx = rows.reshape(-1, 60).min(0), rows.reshape(-1, 60).max(0)
    full = np.concatenate([(rows - mn) / (mx - mn), ref_lat], axis=2)                       # [B][T][188]
    m = RNN(188, 128, 4, 24, torch.device("cpu"))
    ref_out, _, _ = orc.gru_forward(full, orc.flatten_state_dict(m.state_dict(), 4), 188, 128, 4, 24)
    eng = Engine(0)
    eng.set_noise(Q_FITTED, R_FITTED)
    eng.load_gru(flatten_state_dict(m.state_dict(), 4), 188, 128, 4, 24)
    s = {k: eng.pack(torch.as_tensor(d[k])) for k in ("p", "f", "dp", "imu", "accel")}
    c = eng.pack_contact(torch.as_tensor(d["contact"]))
    x = torch.as_tensor(d["x0"].T.copy()).cuda()
    P = torch.as_tensor(np.tile(Q_FITTED.astype(np.float32).reshape(144, 1), (1, B))).cuda()
    mm = torch.as_tensor(np.stack([mn, mx]).astype(np.float32)).cuda()
    r = eng.fused_run(s["p"], s["f"], s["dp"], s["imu"], c, s["accel"], mm, x, P, latent=eng.pack(lat))
    torch.cuda.synchronize()
    assert np.abs(r["out"].cpu().numpy() - ref_out).max() < 2e-4


def test_fused_mlp_kernel_agrees_with_the_two_gemm_form(monkeypatch):
    """The block tail has three forms: vit_mlp_kernel with the attention projection inside (projection + residual + LayerNorm
    + fc1 + GELU + fc2 + residual in one kernel, default), the same without the projection (OS_VIT_MLP_FUSED=1), and
    layernorm_kernel + vit_gemm launches (OS_VIT_MLP_FUSED=0).  Same latent up to summation order, on a frame count whose
    token matrix ends in a partial 128-row tile."""
    import ctypes as C
    from optistate_amd import Engine, _capi
    from optistate_amd.engine import _ptr
    from optistate_amd.transformer_model import Transformer_Autoencoder
    torch.manual_seed(4)
    m = Transformer_Autoencoder().to("cuda")
    with torch.no_grad():
        for p in m.parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    flat = m._flat(torch.device("cuda:0"))
    d = _capi.OsVitDims(m.img_size, m.patch_size, m.in_chans, m.embed_dim, m.depth, m.num_heads, m.mlp_hidden)
    img = torch.rand(7, 224, 224, device="cuda")
    out = {}
    for mode in ("3", "2", "1", "0"):
        monkeypatch.setenv("OS_VIT_MLP_FUSED", mode)
        e = Engine(0)
        e._check(e.lib.os_vit_load(e._h, C.byref(d), _ptr(flat)), "os_vit_load")
        lat = torch.empty((7, 128), dtype=torch.float32, device="cuda")
        e.profile(True)
        e._check(e.lib.os_vit_encode(e._h, 7, _ptr(img), _ptr(lat), e._stream()), "os_vit_encode")
        torch.cuda.synchronize()
        out[mode] = (lat.clone(), e.profile_read()["vit_gemm"][1])
    # GEMM-phase launches: patch + per block (qkv, proj+LN+mlp) | (qkv, proj, LN+mlp) | (qkv, proj, fc1, fc2)
    assert out["2"][1] == 1 + 3 * 2 and out["1"][1] == 1 + 3 * 3 and out["0"][1] == 1 + 3 * 4
    assert out["3"][1] == 1 + 2 + 2                                      # default: later blocks' LayerNorm + qkv ride in the previous tail
    assert float((out["3"][0] - out["0"][0]).abs().max()) < 5e-6
    assert float((out["1"][0] - out["0"][0]).abs().max()) < 5e-6
    assert float((out["2"][0] - out["0"][0]).abs().max()) < 5e-6


@pytest.mark.parametrize("img,dim,depth,heads", [(64, 128, 2, 2), (96, 128, 1, 4), (64, 256, 2, 4), (224, 128, 1, 2)])
def test_other_constructor_arguments_take_the_general_attention_kernels(img, dim, depth, heads):
    """The reference's constructor takes img_size / embed_dim / depth / num_heads (transformer/transformer_model.py:11-29); only its
    defaults (197 tokens, 4 heads x 32) reach the matrix-core attention kernel.  Other token counts and a head dimension of 64 run
    the one-query-per-lane kernels (attention_kernel<32>, <64>): same float64 restatement, same bar."""
    from optistate_amd.transformer_model import Transformer_Autoencoder
    from oracle import vit_oracle
    torch.manual_seed(3)
    m = Transformer_Autoencoder(img_size=img, embed_dim=dim, depth=depth, num_heads=heads).to("cuda")
    with torch.no_grad():
        for p in m.parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    x = torch.rand(3, 1, img, img)
    lat = m.forward_encoder(x.cuda())
    assert lat.shape == (3, 1, dim)
    sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    ref = vit_oracle.encode(x[:, 0].numpy(), sd, heads=heads)
    err = np.abs(lat.cpu().numpy()[:, 0] - ref).max()
    assert err < 2e-5, err
    from optistate_amd.engine import default_engine
    assert default_engine(0).kernel_name("vit_attn") == "attention_kernel"
