"""Seeded input recipes shared by tools/gen_golden.py (which runs the reference on them in the build container) and by the
tests (which rebuild the same inputs on the GPU box): numpy Generators are reproducible across machines, so the fixture only
has to store the reference's OUTPUTS."""
import numpy as np

G11_SEED, G11_FRAMES = 20261002, 8
VIT_DIMS = dict(img_size=224, patch_size=16, in_chans=1, embed_dim=128, depth=3, num_heads=4, mlp_hidden=512)


def g11_encoder_state(seed=G11_SEED):
    """Encoder weights with the reference's state_dict keys and shapes (transformer/transformer_model.py:20-29 with the
    constructor's defaults).  NOT an initialisation the reference would produce: every bias and LayerNorm parameter is
    non-trivial on purpose, so that a dropped bias, a swapped LayerNorm or a wrong position row shows up in the latent."""
    rng = np.random.default_rng(seed)
    D, Hm, depth, P = 128, 512, 3, 16
    u = lambda shape, fan_in, fan_out: rng.uniform(-1, 1, shape) * np.sqrt(6.0 / (fan_in + fan_out))
    sd = {"patch_embed.proj.weight": u((D, 1, P, P), P * P, D), "patch_embed.proj.bias": rng.normal(0, 0.05, D),
          "cls_token": rng.normal(0, 0.2, (1, 1, D))}
    for i in range(depth):
        b = f"blocks.{i}."
        sd[b + "norm1.weight"] = 1 + rng.normal(0, 0.1, D); sd[b + "norm1.bias"] = rng.normal(0, 0.1, D)
        sd[b + "attn.qkv.weight"] = 2.0 * u((3 * D, D), D, 3 * D); sd[b + "attn.qkv.bias"] = rng.normal(0, 0.1, 3 * D)
        sd[b + "attn.proj.weight"] = u((D, D), D, D); sd[b + "attn.proj.bias"] = rng.normal(0, 0.05, D)
        sd[b + "norm2.weight"] = 1 + rng.normal(0, 0.1, D); sd[b + "norm2.bias"] = rng.normal(0, 0.1, D)
        sd[b + "mlp.fc1.weight"] = u((Hm, D), D, Hm); sd[b + "mlp.fc1.bias"] = rng.normal(0, 0.1, Hm)
        sd[b + "mlp.fc2.weight"] = u((D, Hm), Hm, D); sd[b + "mlp.fc2.bias"] = rng.normal(0, 0.05, D)
    sd["norm.weight"] = 1 + rng.normal(0, 0.1, D); sd["norm.bias"] = rng.normal(0, 0.1, D)
    return {k: v.astype(np.float32) for k, v in sd.items()}


def g11_frames(seed=G11_SEED, n=G11_FRAMES):
    """Depth-like frames in [0, 1] (gru/gru_test.py:49-53): smooth blobs + noise, one all-zero and one all-one frame."""
    rng = np.random.default_rng(seed + 1)
    yy, xx = np.mgrid[0:224, 0:224] / 224.0
    fr = []
    for i in range(n):
        cx, cy, s = rng.uniform(0.2, 0.8), rng.uniform(0.2, 0.8), rng.uniform(0.05, 0.3)
        img = np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * s * s)) * rng.uniform(0.3, 1.0) + rng.uniform(0, 0.2, (224, 224))
        fr.append(np.clip(img, 0.0, 1.0))
    fr[n - 2][:] = 0.0
    fr[n - 1][:] = 1.0
    return np.stack(fr).astype(np.float32)
