"""``PlipModel`` -- the object that replaces ``self.model`` in the reference.

It answers to the three call surfaces SURVEY.md section 8b lists:

* HF ``CLIPModel`` (plip.py:18,50,68; README.md:45-50): ``.to()``, ``.eval()``,
  ``.get_image_features(pixel_values=)``, ``.get_text_features(input_ids=, attention_mask=)``
  returning plain tensors (transformers-4.x semantics, which plip.py:50,68 relies on via
  ``.detach().cpu().numpy()``), and ``model(input_ids=, pixel_values=, attention_mask=)`` ->
  an output with ``logits_per_image / logits_per_text / image_embeds / text_embeds``.
* OpenAI ``clip`` model (reproducibility/embedders/plip.py:48,66; scripts/extract_embedding.py:33,52):
  ``.encode_image(images)``, ``.encode_text(tokens)``, ``.logit_scale``, ``model(images, tokens)``
  -> ``(logits_per_image, logits_per_text)``.

All arithmetic runs in libplipmi.so on the MI355X; there is no CPU path.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Mapping, Optional

import numpy as np
import torch

from . import weights as W
from .config import PlipConfig, get_config
from .engine import Engine


@dataclass
class PlipOutput:
    """Fields of HF ``CLIPOutput`` (modeling_clip.py:106-135) that the hot path produces."""
    logits_per_image: torch.Tensor
    logits_per_text: torch.Tensor
    text_embeds: torch.Tensor
    image_embeds: torch.Tensor
    loss: Optional[torch.Tensor] = None

    def __getitem__(self, k):
        return getattr(self, k)

    def to_tuple(self):
        return (self.logits_per_image, self.logits_per_text, self.text_embeds, self.image_embeds)


class PlipModel:
    def __init__(self, cfg: PlipConfig, state_dict: Mapping[str, object], device="cuda:0", dtype="bf16",
                 max_batch: int = 256, **engine_options):
        """``dtype``: "bf16", "f16" (the reference's own GPU type; 8x smaller operand rounding at the same speed) or "f32".
        ``engine_options``: the per-handle switches of :class:`plip_amd.engine.Engine` (ln_fold, pooled_last_block,
        pack_captions, mfma_attention, graph_batch, text_f16, text_f16_layers, latency_batch)."""
        self.config = cfg
        self.engine = Engine(cfg, state_dict, device=device, dtype=dtype, max_batch=max_batch, **engine_options)
        self.device = self.engine.device
        self.dtype = torch.float32          # I/O dtype; the compute dtype is engine.dtype_name
        self.training = False
        # OpenAI-clip exposes the parameter itself (training_model/clip.py:206 clamps it)
        self.logit_scale = torch.tensor(self.engine.logit_scale, dtype=torch.float32, device=self.device)

    # ---- construction -----------------------------------------------------
    @classmethod
    def from_state_dict(cls, state_dict, cfg: Optional[PlipConfig] = None, **kw) -> "PlipModel":
        sd, cfg = W.normalize_state_dict(state_dict, cfg)
        return cls(cfg, sd, **kw)

    @classmethod
    def from_pretrained(cls, path: str, arch: Optional[str] = None, **kw) -> "PlipModel":
        """Local HF directory (what ``CLIPModel.from_pretrained`` takes, plip.py:26) or an
        OpenAI-clip ``.pt`` state dict (factory.py:23-25).  No hub download: this box has no network."""
        sd, cfg = W.load_checkpoint(path, arch)
        return cls(cfg, sd, **kw)

    @classmethod
    def from_synthetic(cls, arch: str = "ViT-B/32", seed: int = 0, logit_scale=None, **kw) -> "PlipModel":
        cfg = get_config(arch) if isinstance(arch, str) else arch
        return cls(cfg, W.synthetic_state_dict(cfg, seed, logit_scale), **kw)

    # ---- nn.Module-ish no-ops the callers use ---------------------------------
    def to(self, device=None, *a, **kw):
        if device is not None and torch.device(device).type != "cuda":
            raise RuntimeError("PlipModel lives on the MI355X it was created on; there is no CPU path")
        return self

    def eval(self):
        return self

    def float(self):
        return self

    def parameters(self):
        return iter(())

    # ---- HF surface ---------------------------------------------------------
    @torch.no_grad()
    def get_image_features(self, pixel_values=None, **_ignored) -> torch.Tensor:
        if pixel_values is None:
            raise ValueError("You have to specify pixel_values")
        return self.engine.encode_image(pixel_values, normalize=False)

    @torch.no_grad()
    def get_text_features(self, input_ids=None, attention_mask=None, **_ignored) -> torch.Tensor:
        if input_ids is None:
            raise ValueError("You have to specify input_ids")
        return self.engine.encode_text(input_ids, attention_mask, normalize=False)

    @torch.no_grad()
    def forward(self, input_ids=None, pixel_values=None, attention_mask=None, **_ignored):
        # OpenAI calling convention: model(images, tokens) -> (logits_per_image, logits_per_text)
        openai_style = (input_ids is not None and torch.is_tensor(input_ids) and input_ids.is_floating_point()
                        and pixel_values is not None and not pixel_values.is_floating_point())
        if openai_style:
            input_ids, pixel_values = pixel_values, input_ids
        if input_ids is None or pixel_values is None:
            raise ValueError("You have to specify input_ids and pixel_values")
        # modeling_clip.py:793-811: the two towers, here side by side on two HIP streams (Engine.encode_pair: the bits of the
        # one-stream order, the bench's step -- 4.4 -> 4.2 ms at bs = 256)
        img, txt = self.engine.encode_pair(pixel_values, input_ids, attention_mask, normalize=True)
        lpi, lpt, _ = self.engine.logits(img, txt, scale=float(np.exp(float(self.logit_scale))))  # :814-817
        if openai_style:
            return lpi, lpt
        return PlipOutput(logits_per_image=lpi, logits_per_text=lpt, text_embeds=txt, image_embeds=img)

    __call__ = forward

    # ---- OpenAI-clip surface --------------------------------------------------
    def encode_image(self, image: torch.Tensor) -> torch.Tensor:
        return self.engine.encode_image(image, normalize=False)

    def encode_text(self, text: torch.Tensor) -> torch.Tensor:
        # clip.tokenize pads with 0 and OpenAI pools at text.argmax(-1): the legacy rule (eos id 2 / <0)
        return self.engine.encode_text(text, None, normalize=False, eos_token_id=-1)
