"""Generate tests/golden/*.npz from the reference's arithmetic -- TEST INFRASTRUCTURE.

Run in the build container (needs ``transformers``; no GPU):

    python -m oracle.make_golden

For every case the synthetic weights/inputs are re-derivable from seeds
(plip_amd.weights.synthetic_*: numpy RandomState), so only the *outputs* of
HuggingFace ``CLIPModel`` (CPU fp32, sdpa attention -- the HF default the
reference gets, modeling_clip.py:394) are stored, plus weight/input
fingerprints that prove a test regenerated the very same tensors.
"""
from __future__ import annotations

import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle import hf_reference as H  # noqa: E402
from plip_amd import weights as W  # noqa: E402
from plip_amd.config import LN100, get_config  # noqa: E402

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

# name -> (arch, batch, weight seed, pixel seed, ids seed, pad style, logit_scale override)
CASES = {
    "tiny_b6": ("tiny", 6, 0, 1, 2, "eos", None),
    "tiny_b5_zero_pad_ln100": ("tiny", 5, 3, 4, 5, "zero", LN100),
    "vitb32_b4": ("ViT-B/32", 4, 0, 1, 2, "eos", None),
    "vitb32_b3_zero_pad_ln100": ("ViT-B/32", 3, 7, 8, 9, "zero", LN100),
    "tinyp4_b3": ("tiny-p4", 3, 11, 12, 13, "eos", None),          # 257 vision tokens: chunked online-softmax attention
    # BASELINE.json configs[2] at its real size: the very batch bench.py times on rank 0 (weights seed 0, pixels seed
    # 1000, ids seed 2000), all 256 x 256 logits from HF itself.  Embeds + logits only (1.3 MB).
    "vitb32_b256": ("ViT-B/32", 256, 0, 1000, 2000, "eos", None),
    # trained-checkpoint-like statistics (weights.heavy_tailed_state_dict): outlier residual channels x30-100,
    # LayerNorm gains over two decades, logit_scale = ln 100
    "vitb32_b8_heavy": ("ViT-B/32", 8, 5, 31, 32, "eos", LN100),
    # the same checkpoint judged at the benchmark size: a maximum over 65 536 logits, like vitb32_b256 (slim)
    "vitb32_b256_heavy": ("ViT-B/32", 256, 5, 1031, 2031, "eos", LN100),
    # BASELINE.json configs[4]'s architecture end to end: 336 px / patch 14 = 577 vision tokens (chunked MFMA attention inside
    # a 24-layer tower), 588 -> 640 zero-padded patch rows, width 1024 / 16 heads, text width 768 / 12 heads, projection 768
    "vitl14_336_b2": ("ViT-L/14@336px", 2, 3, 41, 42, "eos", None),
    # ... and the first eight pairs of the ViT-L/14@336 share bench.py times (`vitl14_336_b64`: weights seed 3, pixels seed 6000, ids
    # seed 6001): the line's error field for that architecture is then against HF itself, not the numpy oracle
    "vitl14_336_b8": ("ViT-L/14@336px", 8, 3, 6000, 6001, "eos", None),
}
# cases whose fixtures hold embeddings and logits only (no features of every row / hidden states)
SLIM = {"vitb32_b256", "vitb32_b256_heavy", "vitl14_336_b8"}
# hidden-state rows are stored at these depths only (fractions of the tower depth) for the big architectures
HIDDEN_DEPTHS = {"vitl14_336_b2": (0, 1, 12, 24)}


def fingerprint(sd) -> np.ndarray:
    """Order-independent digest of a state dict: per-tensor (sum, sum of squares) in float64."""
    keys = sorted(sd)
    return np.array([[np.asarray(sd[k], dtype=np.float64).sum(),
                      (np.asarray(sd[k], dtype=np.float64) ** 2).sum()] for k in keys])


def case_inputs(name):
    arch, B, ws, ps, is_, pad, ls = CASES[name]
    cfg = get_config(arch)
    if not arch.startswith("tiny") and pad == "zero":
        cfg = cfg.replace(eos_token_id=2)       # OpenAI-clip / legacy-HF argmax pooling rule
    if name.endswith("_heavy"):
        sd = W.heavy_tailed_state_dict(cfg, ws, logit_scale=ls)
    else:
        sd = W.synthetic_state_dict(cfg, ws, logit_scale=ls)
    px = W.synthetic_pixels(cfg, B, ps)
    ids, mask = W.synthetic_ids(cfg.replace(eos_token_id=get_config(arch).eos_token_id), B, is_, pad=pad)
    return cfg, sd, px, ids, mask


def main(only=None):
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    for name in (only or CASES):
        cfg, sd, px, ids, mask = case_inputs(name)
        model = H.build_model(cfg, sd, "sdpa")
        # the OpenAI tokenizer gives no attention_mask: zero-pad cases run without one
        use_mask = None if "zero_pad" in name else mask
        slim = name in SLIM
        out = H.run(model, px, ids, use_mask, output_hidden_states=not slim)
        save = {k: out[k] for k in (("image_embeds", "text_embeds", "logits_per_image") if slim else
                                    ("image_features", "text_features", "image_embeds", "text_embeds",
                                     "logits_per_image", "logits_per_text"))}
        # hidden states: all of them for tiny; CLS / first-EOS rows for the full model (small files)
        vh, th = (None, None) if slim else (out["vision_hidden"], out["text_hidden"])
        if slim:
            pass
        elif cfg.v_width <= 128 and cfg.v_tokens <= 64:
            save["vision_hidden"] = np.stack(vh)
            save["text_hidden"] = np.stack(th)
        elif name in HIDDEN_DEPTHS:
            dv = HIDDEN_DEPTHS[name]
            dt = tuple(min(d, cfg.t_layers) for d in (0, 1, cfg.t_layers // 2, cfg.t_layers))
            eos = np.argmax(ids == get_config(CASES[name][0]).eos_token_id, axis=1)
            save["vision_depths"], save["text_depths"] = np.array(dv), np.array(dt)
            save["vision_hidden_cls"] = np.stack([vh[d][:, 0, :] for d in dv])
            save["vision_hidden_last_token"] = np.stack([vh[d][:, -1, :] for d in dv])
            save["text_hidden_bos"] = np.stack([th[d][:, 0, :] for d in dt])
            save["text_hidden_eos"] = np.stack([th[d][np.arange(len(ids)), eos, :] for d in dt])
        else:
            save["vision_hidden_cls"] = np.stack([h[:, 0, :] for h in vh])
            save["vision_hidden_last_token"] = np.stack([h[:, -1, :] for h in vh])
            save["text_hidden_bos"] = np.stack([h[:, 0, :] for h in th])
            save["text_hidden_tok1"] = np.stack([h[:, 1, :] for h in th])
        save["weights_fingerprint"] = fingerprint(sd)
        save["pixels_fingerprint"] = np.array([px.astype(np.float64).sum(), (px.astype(np.float64) ** 2).sum()])
        save["ids"] = ids
        save["mask"] = mask
        path = os.path.join(GOLDEN_DIR, name + ".npz")
        np.savez_compressed(path, **save)
        print(f"{name}: wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)  "
              f"|logits|max={np.abs(out['logits_per_image']).max():.4f}")


if __name__ == "__main__":
    main(sys.argv[1:] or None)
