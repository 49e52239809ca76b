"""G14 -- the ViT encoder with NOTHING of ours between the reference's glue and the latent: the reference's own
`Transformer_Autoencoder` (transformer/transformer_model.py, imported unmodified) with tools/hf_vit_blocks.py providing
`PatchEmbed` / `Block` as thin containers around Hugging Face transformers' `ViTPatchEmbeddings` / `ViTLayer`.  Same seeded
weights and frames as G11 (tests/golden_recipes.py), so G14 == G11 shows that the timm stand-in G11 used and an independent
third-party implementation of the block agree to float64 rounding.

    python tools/gen_golden_vit_hf.py        (build container only: reads /root/reference; writes tests/golden/vit_g14_hf_blocks.npz)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import hf_vit_blocks                                    # noqa: E402
from golden_recipes import g11_encoder_state, g11_frames, G11_SEED   # noqa: E402

provider = hf_vit_blocks.install()
sys.path.insert(0, "/root/reference")
from transformer.transformer_model import Transformer_Autoencoder as RefViT      # noqa: E402

torch.manual_seed(0)
m = RefViT()
w = g11_encoder_state()
missing = m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=False)
assert all(k.startswith("decoder") or k == "pos_embed" for k in missing.missing_keys), missing.missing_keys
assert not missing.unexpected_keys
fr = torch.from_numpy(g11_frames()).unsqueeze(1)
m.eval()
taps = []
hooks = [b.register_forward_hook(lambda mod, i, o: taps.append(o.detach().double().numpy())) for b in m.blocks]
with torch.no_grad():
    lat32 = m.forward_encoder(fr).numpy()
    taps.clear()
    lat64 = m.double().forward_encoder(fr.double()).numpy()
for h in hooks:
    h.remove()
assert lat64.shape == (8, 1, 128) and len(taps) == 3
# token rows of every block's output for frames 0 and 7: cls, the first patches, the last patches (a block-level handle for debugging)
rows = np.r_[0:6, 191:197]
blk = np.stack([t[[0, 7]][:, rows] for t in taps])               # [3][2][12][128]
out = os.path.join(ROOT, "tests", "golden", "vit_g14_hf_blocks.npz")
np.savez_compressed(out, latent_f32=lat32[:, 0], latent_f64=lat64[:, 0], block_rows=rows, block_out_f64=blk,
                    seed=np.array([G11_SEED]), blocks=np.array([provider]))
g11 = np.load(os.path.join(ROOT, "tests", "golden", "vit_g11_glue.npz"))
print("G14:", provider, "| |G14 - G11| f64", np.abs(lat64[:, 0] - g11["latent_f64"]).max(), "f32", np.abs(lat32[:, 0] - g11["latent_f32"]).max())
