"""oracle/vit_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

float64 numpy restatement of the reference's ViT encoder `Transformer_Autoencoder.forward_encoder`
(transformer/transformer_model.py:113-135) with timm==0.3.2 `PatchEmbed`/`Block` semantics as published for that release
(Conv2d(k=16,s=16) -> flatten -> transpose; x + Attn(LN(x)); x + MLP(LN(x)); fused qkv Linear with bias split as
(3, heads, head_dim), scale head_dim**-0.5, softmax, proj; Linear -> GELU(erf) -> Linear; LayerNorm eps 1e-5) and the
MAE-style fixed 2-D sin-cos position table (transformer/pos_embed.py:20-67).

PARITY UNPINNED: timm 0.3.2 is not installed in the build image, the trained weights are not in the repository and the
reference has no test for this path, so nothing here is checked against the reference itself.
"""
import numpy as np
from scipy.special import erf


def sincos_pos_embed(dim, grid):
    """transformer/pos_embed.py:20-67: [1 + grid*grid][dim], cls row zero; w goes first in the meshgrid."""
    def one_d(d, pos):
        omega = 1.0 / 10000 ** (np.arange(d // 2, dtype=np.float64) / (d / 2.0))
        out = np.einsum("m,d->md", pos.reshape(-1), omega)
        return np.concatenate([np.sin(out), np.cos(out)], axis=1)
    gh, gw = np.arange(grid, dtype=np.float32), np.arange(grid, dtype=np.float32)
    g = np.stack(np.meshgrid(gw, gh), axis=0).reshape(2, 1, grid, grid)
    emb = np.concatenate([one_d(dim // 2, g[0]), one_d(dim // 2, g[1])], axis=1)
    return np.concatenate([np.zeros((1, dim)), emb], axis=0)


def layer_norm(x, w, b, eps=1e-5):
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * w + b


def encode(images, sd, patch=16, heads=4, block_outputs=None):
    """images [N][H][W] float; sd: dict of numpy arrays with the reference's state_dict keys -> latent [N][D].
    block_outputs: optional list that receives the token tensor [N][1 + G*G][D] after every block."""
    g = lambda k: np.asarray(sd[k], dtype=np.float64)
    x = np.asarray(images, dtype=np.float64)
    N, Hh, Ww = x.shape
    G = Hh // patch
    pw = g("patch_embed.proj.weight").reshape(-1, patch * patch)          # [D][P*P]
    D = pw.shape[0]
    patches = x.reshape(N, G, patch, G, patch).transpose(0, 1, 3, 2, 4).reshape(N, G * G, patch * patch)
    tok = patches @ pw.T + g("patch_embed.proj.bias")
    pos = g("pos_embed").reshape(-1, D)
    tok = tok + pos[1:]
    cls = np.broadcast_to(g("cls_token").reshape(1, 1, D) + pos[:1], (N, 1, D))
    x = np.concatenate([cls, tok], axis=1)
    depth = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))
    hd = D // heads
    for i in range(depth):
        p = f"blocks.{i}."
        y = layer_norm(x, g(p + "norm1.weight"), g(p + "norm1.bias"))
        qkv = (y @ g(p + "attn.qkv.weight").T + g(p + "attn.qkv.bias")).reshape(N, -1, 3, heads, hd).transpose(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        att = (q @ k.transpose(0, 1, 3, 2)) * hd ** -0.5
        att = np.exp(att - att.max(-1, keepdims=True))
        att = att / att.sum(-1, keepdims=True)
        o = (att @ v).transpose(0, 2, 1, 3).reshape(N, -1, D)
        x = x + o @ g(p + "attn.proj.weight").T + g(p + "attn.proj.bias")
        y = layer_norm(x, g(p + "norm2.weight"), g(p + "norm2.bias"))
        h = y @ g(p + "mlp.fc1.weight").T + g(p + "mlp.fc1.bias")
        h = 0.5 * h * (1.0 + erf(h / np.sqrt(2.0)))
        x = x + h @ g(p + "mlp.fc2.weight").T + g(p + "mlp.fc2.bias")
        if block_outputs is not None:
            block_outputs.append(x.copy())
    x = layer_norm(x, g("norm.weight"), g("norm.bias"))
    return 1.0 / (1.0 + np.exp(-x[:, 0, :]))
