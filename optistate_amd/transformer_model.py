"""Encoder half of the reference's `Transformer_Autoencoder` (transformer/transformer_model.py:10-135) on the HIP path.

Same constructor signature and the same `state_dict` keys for the encoder (timm 0.3.2 naming: patch_embed.proj.*,
cls_token, pos_embed, blocks.{i}.norm1/attn.qkv/attn.proj/norm2/mlp.fc1/mlp.fc2.*, norm.*), so a checkpoint saved by
`transformer/autoencoder_training.py:128-131` loads with `load_state_dict(..., strict=False)` (decoder keys are
ignored: the decoder and its loss are training-only and out of scope).  `forward_encoder(x)` takes (N,1,224,224) in [0,1]
(gru/gru_test.py:49-53) and returns (N,1,128) like the reference; it runs `os_vit_encode` (hand-written HIP kernels only: fp32-MFMA GEMM with fused epilogues, MFMA attention).
Parity: the reference's own class (unmodified glue: patch + position table, cls token, block loop, final LayerNorm, token 0, sigmoid)
around a plain-torch restatement of the timm-0.3.2 blocks (golden G11) and around Hugging Face transformers' ViTLayer (G14: identical
float32 latents); timm 0.3.2 itself and the trained weights are absent from the image.
"""
import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from . import _capi
from .engine import default_engine, _ptr


def _sincos_pos_embed(dim, grid):
    def one_d(d, pos):
        omega = 1.0 / 10000 ** (np.arange(d // 2, dtype=np.float64) / (d / 2.0))
        out = np.einsum("m,d->md", pos.reshape(-1), omega)
        return np.concatenate([np.sin(out), np.cos(out)], axis=1)
    gh, gw = np.arange(grid, dtype=np.float32), np.arange(grid, dtype=np.float32)
    g = np.stack(np.meshgrid(gw, gh), axis=0).reshape(2, 1, grid, grid)          # w first (pos_embed.py:27)
    emb = np.concatenate([one_d(dim // 2, g[0]), one_d(dim // 2, g[1])], axis=1)
    return np.concatenate([np.zeros((1, dim)), emb], axis=0)


class _Attn(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)


class _Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


class _Block(nn.Module):          # weight container with timm's parameter names
    def __init__(self, dim, hidden):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn = _Attn(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = _Mlp(dim, hidden)


class _PatchEmbed(nn.Module):
    def __init__(self, img_size, patch_size, in_chans, embed_dim):
        super().__init__()
        self.patch_size = (patch_size, patch_size)
        self.num_patches = (img_size // patch_size) ** 2
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)


class Transformer_Autoencoder(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=1, embed_dim=128, depth=3, num_heads=4,
                 decoder_embed_dim=128, decoder_depth=3, decoder_num_heads=4, mlp_ratio=4., norm_layer=nn.LayerNorm):
        super().__init__()
        self.in_chans, self.img_size, self.patch_size = in_chans, img_size, patch_size
        self.embed_dim, self.depth, self.num_heads = embed_dim, depth, num_heads
        self.mlp_hidden = int(embed_dim * mlp_ratio)
        self.patch_embed = _PatchEmbed(img_size, patch_size, in_chans, embed_dim)
        self.num_patches = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, self.num_patches + 1, embed_dim), requires_grad=False)
        self.blocks = nn.ModuleList([_Block(embed_dim, self.mlp_hidden) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim)
        self.pos_embed.data.copy_(torch.from_numpy(_sincos_pos_embed(embed_dim, int(self.num_patches ** .5))).float().unsqueeze(0))
        torch.nn.init.normal_(self.cls_token, std=.02)                 # transformer_model.py:66
        w = self.patch_embed.proj.weight.data
        torch.nn.init.xavier_uniform_(w.view([w.shape[0], -1]))        # transformer_model.py:62-63
        for m in self.modules():                                       # transformer_model.py:71-82
            if isinstance(m, nn.Linear):
                torch.nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
        self._loaded = None

    def _flat(self, dev):
        sd = self.state_dict()
        parts = [sd["patch_embed.proj.weight"].reshape(-1), sd["patch_embed.proj.bias"], sd["cls_token"].reshape(-1),
                 sd["pos_embed"].reshape(-1)]
        for i in range(self.depth):
            p = f"blocks.{i}."
            for k in ("norm1.weight", "norm1.bias", "attn.qkv.weight", "attn.qkv.bias", "attn.proj.weight", "attn.proj.bias",
                      "norm2.weight", "norm2.bias", "mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight", "mlp.fc2.bias"):
                parts.append(sd[p + k].reshape(-1))
        parts += [sd["norm.weight"], sd["norm.bias"]]
        return torch.cat([t.detach().float().reshape(-1) for t in parts]).to(dev).contiguous()

    def forward_encoder(self, x):
        if not x.is_cuda:
            raise RuntimeError("optistate_amd Transformer_Autoencoder.forward_encoder needs a tensor on the MI355X (no CPU fallback)")
        eng = default_engine(x.device.index or 0)          # the calling thread's context on this GPU
        with eng.lock:                                     # load + encode as one unit: a context holds ONE encoder
            return self._encode_locked(eng, x)

    def _encode_locked(self, eng, x):
        versions = tuple((p.data_ptr(), p._version) for p in self.parameters())
        if self._loaded is None or self._loaded[0] != versions or getattr(eng, "_vit_owner", None) is not self:
            flat = self._flat(x.device)
            d = _capi.OsVitDims(self.img_size, self.patch_size, self.in_chans, self.embed_dim, self.depth, self.num_heads,
                                self.mlp_hidden)
            assert flat.numel() == eng.lib.os_vit_param_count(C.byref(d)), "flat ViT weight vector has the wrong length"
            eng._check(eng.lib.os_vit_load(eng._h, C.byref(d), _ptr(flat)), "os_vit_load")
            self._loaded = (versions, flat)        # keep the flat tensor alive: the library references it
            eng._vit_owner = self                  # a context holds ONE encoder: another module's load evicts this one
        N = x.shape[0]
        img = x.reshape(N, self.img_size, self.img_size).to(torch.float32).contiguous()
        lat = torch.empty((N, self.embed_dim), dtype=torch.float32, device=x.device)
        eng._check(eng.lib.os_vit_encode(eng._h, N, _ptr(img), _ptr(lat), eng._stream()), "os_vit_encode")
        return lat.reshape(N, 1, self.embed_dim)

    def forward(self, *a, **k):
        raise NotImplementedError("only forward_encoder (the latent used by the GRU, gru/gru_test.py:119-132) is implemented; "
                                  "the decoder and reconstruction loss are training-only and out of scope")
