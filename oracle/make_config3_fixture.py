"""tests/golden/config3_zero_shot.npz -- TEST INFRASTRUCTURE.  The reference's zero-shot head
(reproducibility/evaluation/zero_shot/zero_shot.py:12-13: ``score = image_embeddings.dot(text_embeddings.T)``, per-row
arg-max) on a sample of BASELINE.json configs[3]'s synthetic corpus, computed by HuggingFace ``CLIPModel`` (CPU fp32).

    python -m oracle.make_config3_fixture

On random-init weights ten arbitrary prompts send every tile to the same class ("top-1 100 %" proved
nothing).  So the ten class prompts are CHOSEN: from a pool of seeded candidate prompts, the ten whose mean scores over
the sample lie closest together -- then what separates the classes is the per-tile part of the score and the arg-max
spreads over the classes.  Stored: the chosen prompt ids, HF's [sample, 10] score matrix and its arg-max.  Inputs are
re-derivable from seeds (plip_amd.weights.synthetic_tiles / synthetic_ids / synthetic_state_dict)."""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle import hf_reference as H  # noqa: E402
from plip_amd import weights as W  # noqa: E402
from plip_amd.config import get_config  # noqa: E402
from plip_amd.preprocess import CLIP_MEAN, CLIP_STD  # noqa: E402

SAMPLE, POOL, CLASSES = 512, 384, 10
TILE_SEED, PROMPT_SEED, WEIGHT_SEED = 1000, 7, 0
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "config3_zero_shot.npz")


def tiles_to_pixels(u8):
    px = (u8.astype(np.float32) / np.float32(255.0) - np.asarray(CLIP_MEAN, np.float32)) / np.asarray(CLIP_STD, np.float32)
    return np.ascontiguousarray(px.transpose(0, 3, 1, 2))


def main():
    cfg = get_config("ViT-B/32")
    sd = W.synthetic_state_dict(cfg, WEIGHT_SEED)
    hf = H.build_model(cfg, sd, "sdpa")
    u8 = W.synthetic_tiles(cfg, SAMPLE, TILE_SEED)
    px = tiles_to_pixels(u8)
    cand, _ = W.synthetic_ids(cfg, POOL, seed=PROMPT_SEED)
    with torch.no_grad():
        fi = np.concatenate([H._tensor(hf.get_image_features(pixel_values=torch.from_numpy(px[s:s + 64]))).numpy()
                             for s in range(0, SAMPLE, 64)])
        ft = np.concatenate([H._tensor(hf.get_text_features(input_ids=torch.from_numpy(cand[s:s + 64]))).numpy()
                             for s in range(0, POOL, 64)])
    fi = fi / np.linalg.norm(fi, axis=1, keepdims=True)
    ft = ft / np.linalg.norm(ft, axis=1, keepdims=True)
    S = fi @ ft.T                                              # [SAMPLE, POOL]
    order = np.argsort(S.mean(axis=0))
    best = None
    for w0 in range(0, POOL - CLASSES + 1):                    # ten prompts adjacent in mean score: the most even histogram wins
        cols = order[w0:w0 + CLASSES]
        hist = np.bincount(S[:, cols].argmax(1), minlength=CLASSES)
        key = (int((hist >= SAMPLE // 25).sum()), -int(hist.max()))
        if best is None or key > best[0]:
            best = (key, cols, hist)
    _, cols, hist = best
    prompts = cand[cols]
    scores = S[:, cols].astype(np.float32)
    gaps = np.diff(np.sort(scores, axis=1), axis=1)
    print(f"chosen candidate prompts {cols.tolist()}; class histogram {hist.tolist()}; classes holding >= 4 % of the sample: "
          f"{int((hist >= SAMPLE // 25).sum())}; top-2 gap median {np.median(gaps[:, -1]):.2e}, rows with top-2 gap > 2e-3: "
          f"{int((gaps[:, -1] > 2e-3).sum())}, > 5e-4: {int((gaps[:, -1] > 5e-4).sum())}")
    np.savez_compressed(OUT, prompts=prompts, scores=scores, argmax=scores.argmax(1).astype(np.int32), candidate_index=cols,
                        tiles_fingerprint=np.array([u8.astype(np.float64).sum(), (u8.astype(np.float64) ** 2).sum()]),
                        seeds=np.array([TILE_SEED, PROMPT_SEED, WEIGHT_SEED, SAMPLE, POOL]))
    print(f"wrote {OUT} ({os.path.getsize(OUT) / 1024:.0f} KiB)")


if __name__ == "__main__":
    main()
