"""Architecture description of the PLIP / CLIP dual encoder.

PLIP is CLIP ViT-B/32 fine-tuned on pathology image-text pairs
(/root/reference README.md:3-4, reproducibility/config_example.env:4), so the
engine is parameterised by the same numbers HuggingFace keeps in
``CLIPConfig`` (transformers/models/clip/configuration_clip.py:47-64,97-109,
160-161) and OpenAI-clip keeps in its ``clip.load(name)`` table
(reproducibility/embedders/factory.py:21).
"""
from __future__ import annotations

import dataclasses
import math
from dataclasses import dataclass


@dataclass(frozen=True)
class PlipConfig:
    # vision tower (ViT)
    image_size: int = 224
    patch_size: int = 32
    v_width: int = 768
    v_layers: int = 12
    v_heads: int = 12
    v_mlp: int = 3072
    # text tower
    vocab_size: int = 49408
    context_length: int = 77
    t_width: int = 512
    t_layers: int = 12
    t_heads: int = 8
    t_mlp: int = 2048
    # joint space
    projection_dim: int = 512
    layer_norm_eps: float = 1e-5
    # HF picks the pooled text row as the first ``eos_token_id`` (49407); the
    # legacy value 2 means "argmax of the ids" (modeling_clip.py:561-581).
    eos_token_id: int = 49407
    bos_token_id: int = 49406
    logit_scale_init: float = 2.6592

    # ---- derived -----------------------------------------------------
    @property
    def grid(self) -> int:
        return self.image_size // self.patch_size

    @property
    def num_patches(self) -> int:
        return self.grid * self.grid

    @property
    def v_tokens(self) -> int:
        return self.num_patches + 1

    @property
    def patch_dim(self) -> int:
        return 3 * self.patch_size * self.patch_size

    @property
    def head_dim(self) -> int:
        return self.v_width // self.v_heads

    def validate(self) -> None:
        if self.v_width % self.v_heads or self.t_width % self.t_heads:
            raise ValueError("width must be a multiple of heads")
        if self.v_width // self.v_heads != 64 or self.t_width // self.t_heads != 64:
            raise ValueError("the MI355X kernels are built for head_dim == 64 "
                             "(true for every published CLIP/PLIP variant)")
        if self.image_size % self.patch_size:
            raise ValueError("image_size must be a multiple of patch_size")
        for name in ("v_width", "v_mlp", "t_width", "t_mlp"):
            if getattr(self, name) % 128:
                raise ValueError(f"{name} must be a multiple of 128 (GEMM tile)")
        # 3*patch^2 need not be a multiple of the GEMM K tile: the unfold kernel
        # zero-pads the patch rows (ViT-L/14: 588 -> 640).

    # ---- algorithmic work ------------------------
    def _tower_flops(self, tokens: int, width: int, layers: int, mlp: int) -> float:
        per_layer = 2.0 * tokens * width * (3 * width)      # q,k,v projections
        per_layer += 2.0 * tokens * width * width           # out projection
        per_layer += 2.0 * 2.0 * tokens * width * mlp       # fc1 + fc2
        per_layer += 2.0 * 2.0 * tokens * tokens * width    # QK^T and PV, dense
        return layers * per_layer

    def image_flops(self) -> float:
        f = 2.0 * self.num_patches * self.patch_dim * self.v_width
        f += self._tower_flops(self.v_tokens, self.v_width, self.v_layers, self.v_mlp)
        f += 2.0 * self.v_width * self.projection_dim
        return f

    def text_flops(self) -> float:
        f = self._tower_flops(self.context_length, self.t_width, self.t_layers, self.t_mlp)
        f += 2.0 * self.t_width * self.projection_dim
        return f

    def pair_flops(self) -> float:
        return self.image_flops() + self.text_flops()

    def replace(self, **kw) -> "PlipConfig":
        return dataclasses.replace(self, **kw)


# ``PC_CLIP_ARCH`` names used by reproducibility/config_example.env:4 and
# OpenAI-clip's ``clip.load`` (reproducibility/embedders/factory.py:21).
PRESETS = {
    "ViT-B/32": PlipConfig(),
    "ViT-B/16": PlipConfig(patch_size=16),
    "ViT-L/14": PlipConfig(patch_size=14, v_width=1024, v_layers=24, v_heads=16, v_mlp=4096,
                           t_width=768, t_heads=12, t_mlp=3072, projection_dim=768),
    "ViT-L/14@336px": PlipConfig(image_size=336, patch_size=14, v_width=1024, v_layers=24,
                                 v_heads=16, v_mlp=4096, t_width=768, t_heads=12, t_mlp=3072,
                                 projection_dim=768),
    # small shapes for CPU-side tests and golden fixtures (same code paths,
    # every GEMM/LN/attention constraint of the kernels still holds)
    "tiny": PlipConfig(image_size=64, patch_size=16, v_width=128, v_layers=2, v_heads=2,
                       v_mlp=256, vocab_size=512, context_length=16, t_width=128, t_layers=2,
                       t_heads=2, t_mlp=256, projection_dim=64, eos_token_id=511,
                       bos_token_id=510),
}
# the tiny model with a 4-pixel patch: 257 vision tokens (the ViT-L/14 count) -> the long-sequence attention path
PRESETS["tiny-p4"] = PRESETS["tiny"].replace(patch_size=4)
# widths of 256: every GEMM of the tower takes the 256-column tiles
PRESETS["tiny-w256"] = PRESETS["tiny"].replace(v_width=256, v_heads=4, v_mlp=512, t_width=256, t_heads=4, t_mlp=512)


def get_config(name: str = "ViT-B/32") -> PlipConfig:
    try:
        return PRESETS[name]
    except KeyError:
        raise KeyError(f"unknown architecture {name!r}; known: {sorted(PRESETS)}") from None


def from_hf_config(hf) -> PlipConfig:
    """Build a PlipConfig from a ``transformers.CLIPConfig`` (or its dict)."""
    d = hf if isinstance(hf, dict) else hf.to_dict()
    t, v = d["text_config"], d["vision_config"]
    return PlipConfig(
        image_size=v["image_size"], patch_size=v["patch_size"], v_width=v["hidden_size"],
        v_layers=v["num_hidden_layers"], v_heads=v["num_attention_heads"],
        v_mlp=v["intermediate_size"], vocab_size=t["vocab_size"],
        context_length=t["max_position_embeddings"], t_width=t["hidden_size"],
        t_layers=t["num_hidden_layers"], t_heads=t["num_attention_heads"],
        t_mlp=t["intermediate_size"], projection_dim=d["projection_dim"],
        layer_norm_eps=v.get("layer_norm_eps", 1e-5),
        eos_token_id=t.get("eos_token_id", 49407) if not isinstance(t.get("eos_token_id"), list)
        else t["eos_token_id"][0],
        bos_token_id=t.get("bos_token_id", 49406),
        logit_scale_init=d.get("logit_scale_init_value", 2.6592),
    )


LN100 = math.log(100.0)
