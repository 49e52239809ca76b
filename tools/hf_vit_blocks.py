"""A SECOND provider of the two timm==0.3.2 classes the reference's ViT imports (`transformer/transformer_model.py:3`), whose
arithmetic is NOT written in this repository: `Block.forward` and `PatchEmbed.forward` delegate to Hugging Face transformers'
`ViTLayer` / `ViTPatchEmbeddings` (modeling_vit.py of the installed wheel -- third-party code, an independent implementation of
the published ViT encoder block that exists to run timm-trained checkpoints).  Only the parameter containers live here, under
timm's names (norm1, attn.qkv, attn.proj, norm2, mlp.fc1, mlp.fc2; proj), so that the REFERENCE'S OWN `Transformer_Autoencoder`
constructs, initialises and loads unmodified; at forward time they are handed to the HF module by transformers' own
timm -> HF conversion rule (convert_vit_timm_to_pytorch.py: rows [0, D) of qkv = query, [D, 2D) = key, [2D, 3D) = value).

Fixture generation only (tools/gen_golden_vit_hf.py -> G14); never shipped, never imported by the product or the tests.
What G14 adds over G11: G11's blocks were tools/timm_standin.py (ours, written to timm 0.3.2 as published); here nothing between
the reference's glue and the latent is ours.  It still is not timm 0.3.2 itself, which this image does not have."""
import sys
import types

import torch
import torch.nn as nn


def _hf_config(dim, heads, hidden, eps, img=224, patch=16, chans=1):
    from transformers.models.vit.modeling_vit import ViTConfig
    cfg = ViTConfig(hidden_size=dim, num_hidden_layers=1, num_attention_heads=heads, intermediate_size=hidden, hidden_act="gelu",
                    layer_norm_eps=eps, qkv_bias=True, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
                    image_size=img, patch_size=patch, num_channels=chans)
    cfg._attn_implementation = "eager"          # the plain softmax(q k^T / sqrt(d)) v path of modeling_vit.py
    return cfg


class _Attn(nn.Module):                     # container only (timm names)
    def __init__(self, dim, qkv_bias):
        super().__init__()
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)


class _Mlp(nn.Module):                      # container only
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0., attn_drop=0., drop_path=0.,
                 act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        assert qk_scale is None and drop == 0. and attn_drop == 0. and drop_path == 0. and act_layer is nn.GELU and qkv_bias
        self.dim, self.heads, self.hidden = dim, num_heads, int(dim * mlp_ratio)
        self.norm1 = norm_layer(dim)
        self.attn = _Attn(dim, qkv_bias)
        self.norm2 = norm_layer(dim)
        self.mlp = _Mlp(dim, self.hidden)

    def forward(self, x):
        from transformers.models.vit.modeling_vit import ViTLayer
        D = self.dim
        hf = ViTLayer(_hf_config(D, self.heads, self.hidden, self.norm1.eps)).to(x.dtype).eval()
        w, b = self.attn.qkv.weight.detach(), self.attn.qkv.bias.detach()
        sd = {"attention.q_proj.weight": w[:D], "attention.q_proj.bias": b[:D],
              "attention.k_proj.weight": w[D:2 * D], "attention.k_proj.bias": b[D:2 * D],
              "attention.v_proj.weight": w[2 * D:], "attention.v_proj.bias": b[2 * D:],
              "attention.o_proj.weight": self.attn.proj.weight, "attention.o_proj.bias": self.attn.proj.bias,
              "layernorm_before.weight": self.norm1.weight, "layernorm_before.bias": self.norm1.bias,
              "layernorm_after.weight": self.norm2.weight, "layernorm_after.bias": self.norm2.bias,
              "mlp.fc1.weight": self.mlp.fc1.weight, "mlp.fc1.bias": self.mlp.fc1.bias,
              "mlp.fc2.weight": self.mlp.fc2.weight, "mlp.fc2.bias": self.mlp.fc2.bias}
        hf.load_state_dict({k: v.detach().to(x.dtype) for k, v in sd.items()}, strict=True)
        y = hf(x)
        return y[0] if isinstance(y, (tuple, list)) else y


class PatchEmbed(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
        super().__init__()
        self.img_size, self.patch_size = (img_size, img_size), (patch_size, patch_size)
        self.num_patches = (img_size // patch_size) ** 2
        self.in_chans, self.embed_dim = in_chans, embed_dim
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)      # container (timm name)

    def forward(self, x):
        from transformers.models.vit.modeling_vit import ViTPatchEmbeddings
        hf = ViTPatchEmbeddings(_hf_config(self.embed_dim, 1, self.embed_dim, 1e-5, self.img_size[0], self.patch_size[0],
                                           self.in_chans)).to(x.dtype).eval()
        hf.load_state_dict({"projection.weight": self.proj.weight.detach().to(x.dtype),
                            "projection.bias": self.proj.bias.detach().to(x.dtype)}, strict=True)
        return hf(x)


def install():
    import transformers
    t = types.ModuleType("timm"); tm = types.ModuleType("timm.models"); tv = types.ModuleType("timm.models.vision_transformer")
    tv.PatchEmbed, tv.Block = PatchEmbed, Block
    t.models = tm; tm.vision_transformer = tv
    t.__version__ = "hf-transformers-" + transformers.__version__
    sys.modules["timm"], sys.modules["timm.models"], sys.modules["timm.models.vision_transformer"] = t, tm, tv
    return t.__version__
