"""A stand-in for the two timm==0.3.2 classes the reference's ViT imports (`transformer/transformer_model.py:3`:
`from timm.models.vision_transformer import PatchEmbed, Block`), so that the REFERENCE'S OWN `Transformer_Autoencoder`
(constructor :11-29, `initialize_weights` :54-82, `forward_encoder` :113-135) can be imported and run unmodified in the
build container, where timm is not installed (no network).  Fixture generation only (tools/gen_golden.py g11); never
shipped, never imported by the product or by the tests.

What this pins and what it does not: the fixture G11 is labelled "blocks = stand-in, glue = reference".  `PatchEmbed`
and `Block` below are plain torch modules written to the PUBLISHED timm 0.3.2 definition (module / parameter names
as in that release, so the reference's `state_dict` keys come out right):
  PatchEmbed: Conv2d(in_chans, embed_dim, kernel = stride = patch) -> flatten(2) -> transpose(1, 2)
  Block:      x + Attention(LayerNorm(x)); x + Mlp(LayerNorm(x))   (drop-path 0 = identity)
  Attention:  qkv = Linear(dim, 3 dim, bias=qkv_bias) reshaped (B, N, 3, heads, head_dim) -> permute(2, 0, 3, 1, 4);
              softmax(q k^T * head_dim^-0.5) v; transpose(1, 2) -> reshape (B, N, C); proj Linear(dim, dim)
  Mlp:        Linear -> GELU (erf form, nn.GELU default) -> Linear
Written from knowledge of that release; it cannot be verified here, which is why the ViT row stays "parity unpinned"
for the blocks.  Everything AROUND them is the reference's code: patch + position embedding order, cls token, the
concatenation, the block loop, the final LayerNorm, token 0, the sigmoid, and the weight initialisation.
"""
import sys
import types

import torch
import torch.nn as nn


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)

    def forward(self, x):
        return self.drop(self.fc2(self.drop(self.act(self.fc1(x)))))


class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0.):
        super().__init__()
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = qk_scale or head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)

    def forward(self, x):
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        attn = (q @ k.transpose(-2, -1)) * self.scale
        attn = self.attn_drop(attn.softmax(dim=-1))
        x = (attn @ v).transpose(1, 2).reshape(B, N, C)
        return self.proj_drop(self.proj(x))


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0., attn_drop=0., drop_path=0.,
                 act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=drop)
        self.drop_path = nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)

    def forward(self, x):
        x = x + self.drop_path(self.attn(self.norm1(x)))
        x = x + self.drop_path(self.mlp(self.norm2(x)))
        return x


class PatchEmbed(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
        super().__init__()
        img_size = (img_size, img_size) if isinstance(img_size, int) else tuple(img_size)
        patch_size = (patch_size, patch_size) if isinstance(patch_size, int) else tuple(patch_size)
        self.img_size, self.patch_size = img_size, patch_size
        self.num_patches = (img_size[1] // patch_size[1]) * (img_size[0] // patch_size[0])
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)

    def forward(self, x):
        B, C, H, W = x.shape
        assert H == self.img_size[0] and W == self.img_size[1]
        return self.proj(x).flatten(2).transpose(1, 2)


def install():
    """Registers timm / timm.models / timm.models.vision_transformer stand-in modules (only if timm is really absent)."""
    try:
        import timm  # noqa: F401
        return False
    except ImportError:
        pass
    t = types.ModuleType("timm"); tm = types.ModuleType("timm.models"); tv = types.ModuleType("timm.models.vision_transformer")
    tv.PatchEmbed, tv.Block, tv.Attention, tv.Mlp = PatchEmbed, Block, Attention, Mlp
    t.models = tm; tm.vision_transformer = tv
    t.__version__ = "0.3.2-standin"
    sys.modules["timm"], sys.modules["timm.models"], sys.modules["timm.models.vision_transformer"] = t, tm, tv
    return True
