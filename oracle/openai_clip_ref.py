"""A minimal OpenAI-``clip``-style CLIP in plain ``torch.nn`` -- TEST INFRASTRUCTURE, not product code.

Why it exists: ``reproducibility/embedders/factory.py:21-25`` builds the model with the OpenAI ``clip`` package
(``clip.load(arch)`` + ``load_state_dict(torch.load(backbone))``) -- a third-party dependency that is neither
vendored in /root/reference nor pinned there nor installed here (github.com/openai/CLIP, ``clip/model.py``; the
notebook's "All keys matched successfully" is the only evidence of the layout the reference holds).  The converter
``plip_amd.weights.convert_openai_state_dict`` therefore needs a pin that is NOT its own inverse.  This module restates
the published OpenAI architecture with the SAME module tree and parameter names, so that ``state_dict()`` of it
yields the genuine layout by construction:

* ``visual.conv1.weight`` (no bias), ``visual.class_embedding``, ``visual.positional_embedding``,
  ``visual.ln_pre / ln_post``, ``visual.proj`` stored ``[width, output_dim]`` and applied as ``x @ proj``;
* ``…resblocks.N.attn`` = ``torch.nn.MultiheadAttention`` -> packed ``in_proj_weight [3D, D]`` (q | k | v rows),
  ``in_proj_bias``, ``out_proj.{weight,bias}``; ``ln_1``, ``ln_2``; ``mlp.c_fc``, ``mlp.c_proj`` with QuickGELU
  ``x * sigmoid(1.702 x)`` between them; pre-LN residual blocks;
* ``token_embedding.weight``, ``positional_embedding``, ``ln_final``, ``text_projection`` stored
  ``[width, embed_dim]`` and applied as ``x @ text_projection``, ``logit_scale``;
* text: additive causal mask (``-inf`` above the diagonal), pooled at ``text.argmax(dim=-1)`` (the EOT token has the
  highest id); MultiheadAttention runs sequence-first (``[S, B, D]``), as in the original.

``torch.nn.MultiheadAttention`` does the q/k/v split, the head reshape and the scaling itself -- an implementation
of multi-head attention that shares no code with HF ``CLIPAttention`` or with this repository.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np
import torch
from torch import nn


class QuickGELU(nn.Module):
    def forward(self, x):
        return x * torch.sigmoid(1.702 * x)


class ResidualAttentionBlock(nn.Module):
    def __init__(self, d_model: int, n_head: int, d_mlp: int, attn_mask=None):
        super().__init__()
        self.attn = nn.MultiheadAttention(d_model, n_head)
        self.ln_1 = nn.LayerNorm(d_model)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", nn.Linear(d_model, d_mlp)), ("gelu", QuickGELU()),
                                              ("c_proj", nn.Linear(d_mlp, d_model))]))
        self.ln_2 = nn.LayerNorm(d_model)
        self.attn_mask = attn_mask

    def forward(self, x):
        h = self.ln_1(x)
        m = None if self.attn_mask is None else self.attn_mask.to(dtype=x.dtype)
        x = x + self.attn(h, h, h, need_weights=False, attn_mask=m)[0]
        return x + self.mlp(self.ln_2(x))


class Transformer(nn.Module):
    def __init__(self, width, layers, heads, d_mlp, attn_mask=None):
        super().__init__()
        self.resblocks = nn.Sequential(*[ResidualAttentionBlock(width, heads, d_mlp, attn_mask) for _ in range(layers)])

    def forward(self, x):
        return self.resblocks(x)


class VisionTransformer(nn.Module):
    def __init__(self, input_resolution, patch_size, width, layers, heads, d_mlp, output_dim):
        super().__init__()
        self.conv1 = nn.Conv2d(3, width, kernel_size=patch_size, stride=patch_size, bias=False)
        scale = width ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn((input_resolution // patch_size) ** 2 + 1, width))
        self.ln_pre = nn.LayerNorm(width)
        self.transformer = Transformer(width, layers, heads, d_mlp)
        self.ln_post = nn.LayerNorm(width)
        self.proj = nn.Parameter(scale * torch.randn(width, output_dim))

    def forward(self, x):
        x = self.conv1(x)                                              # [B, width, grid, grid]
        x = x.reshape(x.shape[0], x.shape[1], -1).permute(0, 2, 1)     # [B, grid*grid, width]
        cls = self.class_embedding.to(x.dtype) + torch.zeros(x.shape[0], 1, x.shape[-1], dtype=x.dtype)
        x = torch.cat([cls, x], dim=1) + self.positional_embedding.to(x.dtype)
        x = self.ln_pre(x)
        x = self.transformer(x.permute(1, 0, 2)).permute(1, 0, 2)      # sequence-first through the blocks
        x = self.ln_post(x[:, 0, :])
        return x @ self.proj


class OpenAIStyleCLIP(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.context_length = cfg.context_length
        self.visual = VisionTransformer(cfg.image_size, cfg.patch_size, cfg.v_width, cfg.v_layers, cfg.v_heads,
                                        cfg.v_mlp, cfg.projection_dim)
        mask = torch.full((cfg.context_length, cfg.context_length), float("-inf")).triu_(1)
        self.transformer = Transformer(cfg.t_width, cfg.t_layers, cfg.t_heads, cfg.t_mlp, attn_mask=mask)
        self.token_embedding = nn.Embedding(cfg.vocab_size, cfg.t_width)
        self.positional_embedding = nn.Parameter(torch.empty(cfg.context_length, cfg.t_width))
        self.ln_final = nn.LayerNorm(cfg.t_width)
        self.text_projection = nn.Parameter(torch.empty(cfg.t_width, cfg.projection_dim))
        self.logit_scale = nn.Parameter(torch.ones([]) * np.log(1 / 0.07))

    def encode_image(self, image):
        return self.visual(image)

    def encode_text(self, text):
        x = self.token_embedding(text) + self.positional_embedding
        x = self.transformer(x.permute(1, 0, 2)).permute(1, 0, 2)
        x = self.ln_final(x)
        return x[torch.arange(x.shape[0]), text.argmax(dim=-1)] @ self.text_projection

    def forward(self, image, text):
        i, t = self.encode_image(image), self.encode_text(text)
        i, t = i / i.norm(dim=1, keepdim=True), t / t.norm(dim=1, keepdim=True)
        lpi = self.logit_scale.exp() * i @ t.t()
        return lpi, lpi.t()


def build_random(cfg, seed: int = 0) -> OpenAIStyleCLIP:
    """An OpenAI-layout model with every parameter drawn at random (biases and LayerNorm affine included, so a
    dropped or swapped tensor cannot hide behind a zero / one initial value).  Deterministic per seed."""
    torch.manual_seed(seed)
    m = OpenAIStyleCLIP(cfg).eval()
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in m.named_parameters():
            if name == "logit_scale":
                continue
            if name.endswith("ln_1.weight") or name.endswith("ln_2.weight") or "ln_pre.weight" in name or \
                    "ln_post.weight" in name or "ln_final.weight" in name:
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            elif p.dim() == 1:
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
            elif "embedding" in name:
                p.copy_(0.02 * torch.randn(p.shape, generator=g))
            else:
                fan_in = p.shape[1] if p.dim() == 2 and "proj" not in name.split(".")[-1] else p.shape[0] if p.dim() == 2 else \
                    int(np.prod(p.shape[1:]))
                p.copy_(torch.randn(p.shape, generator=g) * (fan_in ** -0.5))
    return m


def numpy_state_dict(model: OpenAIStyleCLIP):
    return {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}
