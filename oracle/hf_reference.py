"""The reference's actual arithmetic: HuggingFace ``CLIPModel`` on CPU
(what /root/reference/plip.py:26,50,68 and README.md:38-49 call) --
TEST INFRASTRUCTURE.  Used (a) by ``make_golden.py`` to produce the fixtures the
oracle and the HIP path are pinned to, (b) by the tests for live comparisons
when ``transformers`` is importable, (c) by ``bench.py`` as the timed
``cpu_baseline`` of kind "reference" (the third-party forward the reference runs).

The reference ``PLIP`` class itself does not construct under transformers 5.x
(``use_auth_token`` kwarg, plip.py:26), so ``CLIPModel`` is driven directly.
"""
from __future__ import annotations

import numpy as np


def available() -> bool:
    try:
        import transformers  # noqa: F401
        return True
    except Exception:
        return False


def hf_config(cfg, attn_implementation="sdpa"):
    from transformers import CLIPConfig
    hc = CLIPConfig(
        text_config=dict(vocab_size=cfg.vocab_size, hidden_size=cfg.t_width, intermediate_size=cfg.t_mlp,
                         num_hidden_layers=cfg.t_layers, num_attention_heads=cfg.t_heads,
                         max_position_embeddings=cfg.context_length, projection_dim=cfg.projection_dim,
                         layer_norm_eps=cfg.layer_norm_eps, eos_token_id=cfg.eos_token_id,
                         bos_token_id=cfg.bos_token_id),
        vision_config=dict(hidden_size=cfg.v_width, intermediate_size=cfg.v_mlp,
                           num_hidden_layers=cfg.v_layers, num_attention_heads=cfg.v_heads,
                           image_size=cfg.image_size, patch_size=cfg.patch_size,
                           projection_dim=cfg.projection_dim, layer_norm_eps=cfg.layer_norm_eps),
        projection_dim=cfg.projection_dim, logit_scale_init_value=cfg.logit_scale_init)
    hc._attn_implementation = attn_implementation
    return hc


def build_model(cfg, state_dict, attn_implementation="sdpa", dtype=None):
    """HF CLIPModel carrying the given HF-format numpy state dict."""
    import torch
    from transformers import CLIPModel
    model = CLIPModel(hf_config(cfg, attn_implementation))
    sd = {k: torch.from_numpy(np.array(v, dtype=np.float32, copy=True)) for k, v in state_dict.items()}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    bad = [k for k in missing if "position_ids" not in k] + list(unexpected)
    if bad:
        raise RuntimeError(f"state dict does not fit CLIPModel: {bad[:6]}")
    model = model.eval()
    return model.to(dtype) if dtype is not None else model


def _tensor(out):
    # transformers 5.x returns BaseModelOutputWithPooling from get_*_features
    # (4.x, which plip.py was written for, returned the projected tensor).
    return out.pooler_output if hasattr(out, "pooler_output") else out


def run(model, pixels=None, ids=None, attention_mask=None, output_hidden_states=False):
    """Returns numpy outputs named like ``clip_oracle.clip_forward``."""
    import torch
    res = {}
    with torch.no_grad():
        tp = None if pixels is None else torch.from_numpy(np.asarray(pixels)).to(next(model.parameters()).dtype)
        ti = None if ids is None else torch.from_numpy(np.asarray(ids))
        tm = None if attention_mask is None else torch.from_numpy(np.asarray(attention_mask))
        if tp is not None:
            res["image_features"] = _tensor(model.get_image_features(pixel_values=tp)).float().numpy()
            if output_hidden_states:
                vo = model.vision_model(pixel_values=tp, output_hidden_states=True)
                res["vision_hidden"] = [h.float().numpy() for h in vo.hidden_states]
        if ti is not None:
            res["text_features"] = _tensor(model.get_text_features(input_ids=ti, attention_mask=tm)).float().numpy()
            if output_hidden_states:
                to = model.text_model(input_ids=ti, attention_mask=tm, output_hidden_states=True)
                res["text_hidden"] = [h.float().numpy() for h in to.hidden_states]
        if tp is not None and ti is not None:
            out = model(input_ids=ti, pixel_values=tp, attention_mask=tm)
            res["image_embeds"] = out.image_embeds.float().numpy()
            res["text_embeds"] = out.text_embeds.float().numpy()
            res["logits_per_image"] = out.logits_per_image.float().numpy()
            res["logits_per_text"] = out.logits_per_text.float().numpy()
    return res
