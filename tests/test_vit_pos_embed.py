"""CPU: the ViT encoder's fixed 2-D sin-cos position table against G10 -- the output of the reference's OWN
transformer/pos_embed.py (imported unmodified by tools/gen_golden.py; it needs only numpy), called as
transformer/transformer_model.py:64-67 calls it.  This is the part of the optional ViT row that CAN be pinned (timm and the
trained weights are absent, so the encoder as a whole stays parity-unpinned)."""
import numpy as np
import pytest

from conftest import load_golden

CASES = {"enc_128_14": (128, 14), "dec_64_14": (64, 14), "small_32_5": (32, 5)}


@pytest.mark.parametrize("key", sorted(CASES))
def test_product_table_equals_reference_table(key):
    from optistate_amd.transformer_model import _sincos_pos_embed
    dim, grid = CASES[key]
    g = load_golden("vit_g10_pos_embed.npz")[key]
    t = _sincos_pos_embed(dim, grid)
    assert t.shape == g.shape == (grid * grid + 1, dim)
    assert np.abs(t - g).max() < 1e-15                       # same float64 formula, same evaluation order


def test_module_parameter_holds_the_reference_table():
    import torch
    from optistate_amd.transformer_model import Transformer_Autoencoder
    m = Transformer_Autoencoder()
    g = load_golden("vit_g10_pos_embed.npz")["enc_128_14"]
    assert m.pos_embed.shape == (1, 197, 128) and not m.pos_embed.requires_grad
    assert torch.equal(m.pos_embed[0], torch.from_numpy(g).float())        # float32 rounding of the float64 table, bit for bit
    assert float(m.pos_embed[0, 0].abs().max()) == 0.0                     # cls row is zeros (pos_embed.py:34-35)


def test_oracle_table_equals_reference_table():
    from oracle import vit_oracle
    g = load_golden("vit_g10_pos_embed.npz")
    for key, (dim, grid) in CASES.items():
        assert np.abs(vit_oracle.sincos_pos_embed(dim, grid) - g[key]).max() < 1e-15
