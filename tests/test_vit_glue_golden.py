"""G11 -- "blocks = stand-in, glue = reference": the reference's OWN `Transformer_Autoencoder` (transformer/transformer_model.py,
imported unmodified by tools/gen_golden.py g11) run in the build container with tools/timm_standin.py providing the two timm
0.3.2 classes it imports.  Pins the reference-owned lines of the optional ViT row: `forward_encoder` :113-135 (patch embedding
+ pos[1:], cls + pos[0], concatenation, block loop, final LayerNorm, token 0, sigmoid) and `initialize_weights` :54-82.  The
transformer blocks themselves follow the published timm 0.3.2 definition and stay PARITY UNPINNED (timm is absent).
G14 (round 4, tools/gen_golden_vit_hf.py) repeats G11 with NOTHING of ours between the reference's glue and the latent: the two
classes are thin containers around Hugging Face transformers' ViTPatchEmbeddings / ViTLayer (tools/hf_vit_blocks.py), an
independent third-party implementation of the published block -- still not timm 0.3.2 itself.
CPU here: the numpy oracle and the product's weight container; the HIP encoder is checked in test_gpu_vit.py."""
import numpy as np

from conftest import load_golden
from golden_recipes import g11_encoder_state, g11_frames, G11_SEED


def test_recipe_is_the_one_the_fixture_was_generated_from():
    g = load_golden("vit_g11_glue.npz")
    assert int(g["seed"][0]) == G11_SEED and g["latent_f64"].shape == (8, 128)
    fr = g11_frames()
    assert fr.shape == (8, 224, 224) and fr.dtype == np.float32 and fr[6].max() == 0.0 and fr[7].min() == 1.0


def test_oracle_encoder_matches_the_reference_glue():
    """oracle/vit_oracle.py (the checker the HIP kernels are compared with elsewhere) against the reference's class."""
    from oracle import vit_oracle
    g = load_golden("vit_g11_glue.npz")
    sd = dict(g11_encoder_state())
    sd["pos_embed"] = g["init_pos_embed"][None]                         # the table initialize_weights installs (= G10)
    lat = vit_oracle.encode(g11_frames(), sd)
    assert np.abs(lat - g["latent_f64"]).max() < 1e-9                   # float64 vs the reference class in float64
    assert np.abs(lat - g["latent_f32"]).max() < 2e-5                   # and vs the reference as it runs (float32)


def test_product_container_initialises_like_the_reference():
    """What initialize_weights leaves behind (:54-82): the sin-cos table, zero biases, unit LayerNorm weights, Xavier-uniform
    Linear weights (|w| up to the bound sqrt(6 / (fan_in + fan_out))), the patch projection Xavier on its (D, P*P) view,
    cls token ~ N(0, 0.02) -- and the encoder's state_dict keys / shapes."""
    import torch
    from optistate_amd.transformer_model import Transformer_Autoencoder
    g = load_golden("vit_g11_glue.npz")
    torch.manual_seed(0)
    m = Transformer_Autoencoder()
    sd = {k: v.detach().double().numpy() for k, v in m.state_dict().items()}
    assert sorted(sd) == list(g["init_encoder_keys"])
    assert [str(tuple(sd[k].shape)) for k in sorted(sd)] == list(g["init_encoder_shapes"])
    assert np.array_equal(sd["pos_embed"][0].astype(np.float32), g["init_pos_embed"].astype(np.float32))
    assert float(g["init_bias_absmax"][0]) == 0.0 and float(g["init_ln_weight_dev"][0]) == 0.0      # the reference's values ...
    assert max(np.abs(sd[k]).max() for k in sd if k.endswith(".bias") and not k.startswith("patch_embed")) == 0.0   # ... and the product's
    # the Conv2d bias is NOT re-initialised by _init_weights (Linear / LayerNorm only): torch's default U(+-1/sqrt(256)) stays
    assert 0.9 < float(g["init_patch_bias_absmax_over_bound"][0]) <= 1.0
    assert 0.9 < np.abs(sd["patch_embed.proj.bias"]).max() * 16.0 <= 1.0
    assert max(np.abs(sd[k] - 1).max() for k in sd if "norm" in k and k.endswith("weight")) == 0.0
    for k, ref_ratio in zip(g["init_linear_keys"], g["init_linear_absmax_over_bound"]):
        r = np.abs(sd[str(k)]).max() / np.sqrt(6.0 / sum(sd[str(k)].shape))
        assert 0.99 < ref_ratio <= 1.0 and 0.99 < r <= 1.0, (k, ref_ratio, r)
    r = np.abs(sd["patch_embed.proj.weight"]).max() / np.sqrt(6.0 / (256 + 128))
    assert 0.99 < float(g["init_patch_absmax_over_bound"][0]) <= 1.0 and 0.99 < r <= 1.0
    assert 0.015 < float(g["init_cls_std"][0]) < 0.025 and 0.015 < sd["cls_token"].std() < 0.025


def test_g14_third_party_blocks_reproduce_g11():
    """The timm stand-in G11 was generated with and transformers' ViTLayer give the SAME float32 latents bit for bit; in float64
    they differ by 5e-8 because modeling_vit.py evaluates the attention softmax in float32 whatever the input type."""
    g11, g14 = load_golden("vit_g11_glue.npz"), load_golden("vit_g14_hf_blocks.npz")
    assert int(g14["seed"][0]) == G11_SEED and str(g14["blocks"][0]).startswith("hf-transformers-")
    assert np.array_equal(g14["latent_f32"], g11["latent_f32"])
    assert np.abs(g14["latent_f64"] - g11["latent_f64"]).max() < 2e-7


def test_oracle_encoder_and_blocks_match_g14():
    """oracle/vit_oracle.py against the reference glue + third-party blocks: latents, and the token rows G14 kept of every
    block's output (cls, first and last patches of frames 0 and 7)."""
    from oracle import vit_oracle
    g11, g = load_golden("vit_g11_glue.npz"), load_golden("vit_g14_hf_blocks.npz")
    sd = dict(g11_encoder_state())
    sd["pos_embed"] = g11["init_pos_embed"][None]
    taps = []
    lat = vit_oracle.encode(g11_frames(), sd, block_outputs=taps)
    assert np.abs(lat - g["latent_f64"]).max() < 2e-7 and np.abs(lat - g["latent_f32"]).max() < 2e-5
    rows = g["block_rows"]
    assert len(taps) == 3 and g["block_out_f64"].shape == (3, 2, len(rows), 128)
    for i in range(3):
        mine = taps[i][[0, 7]][:, rows]
        assert np.abs(mine - g["block_out_f64"][i]).max() < 1e-6 * max(1.0, np.abs(g["block_out_f64"][i]).max()), i
