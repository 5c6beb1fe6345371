#!/usr/bin/env python
"""The reference's evaluation flow (reproducibility/scripts/zero_shot_evaluation.py + retrieval_evaluation.py) on the
MI355X engine, end to end on synthetic data:

    EmbedderFactory().factory(args) -> CLIPEmbedder -> image_embedder / text_embedder (cache files under
    $PC_CACHE_FOLDER, interchangeable with the reference's) -> ZeroShotClassifier / ImageRetrieval -> metrics

    python examples/reproducibility_eval.py --images 2048 --classes 8 --size 300x260

Without --checkpoint a synthetic OpenAI-format ViT-B/32 state dict is written to a temp dir first, so the factory's
checkpoint ingestion (factory.py:21-25) is exercised as well.  Images are random uint8 arrays of one size: they take
the GPU resize + crop + fused normalisation route (no host preprocessing at all).
"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from plip_amd import weights as W  # noqa: E402
from plip_amd.config import get_config  # noqa: E402
from plip_amd.reproducibility import EmbedderFactory, ImageRetrieval, ZeroShotClassifier  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=2048)
    ap.add_argument("--classes", type=int, default=8)
    ap.add_argument("--size", default="300x260", help="HxW of the synthetic images")
    ap.add_argument("--arch", default="ViT-B/32")
    ap.add_argument("--checkpoint", default=None, help="OpenAI-clip .pt state dict or HF directory")
    ap.add_argument("--dtype", default="bf16")
    args = ap.parse_args()
    h, w = (int(x) for x in args.size.split("x"))
    cfg = get_config(args.arch)
    tmp = tempfile.mkdtemp(prefix="plip_amd_eval_")
    os.environ.setdefault("PC_CACHE_FOLDER", os.path.join(tmp, "cache"))
    os.environ["PC_CLIP_ARCH"] = args.arch
    ck = args.checkpoint
    if ck is None:
        ck = os.path.join(tmp, "synthetic_openai_clip.pt")
        sd = W.synthetic_state_dict(cfg, 0)
        torch.save({k: torch.from_numpy(np.asarray(v)) for k, v in W.to_openai_state_dict(sd, cfg).items()}, ck)
    embedder = EmbedderFactory().factory(argparse.Namespace(model_name="plip", backbone=ck, dtype=args.dtype, max_batch=256))

    rng = np.random.RandomState(0)
    images = [rng.randint(0, 256, (h, w, 3), dtype=np.uint8) for _ in range(args.images)]
    ids, _ = W.synthetic_ids(cfg, args.images, 1, pad="zero")            # one synthetic caption per image (clip.tokenize-style)
    prompts, _ = W.synthetic_ids(cfg, args.classes, 2, pad="zero")       # class prompts
    labels = [f"class_{i}" for i in range(args.classes)]

    from plip_amd.plip import PLIP
    plip = PLIP(model=embedder.model)                                     # same engine, the PLIP-class surface
    t0 = time.perf_counter()
    raw = plip.encode_images(images, batch_size=256)                      # GPU resize + crop + normalise + tower
    img = raw / np.linalg.norm(raw, axis=1, keepdims=True)                # embedders/plip.py:53
    t1 = time.perf_counter()
    txt = embedder.text_embedder(ids, batch_size=256, additional_cache_name="captions")
    cls = embedder.text_embedder(prompts, batch_size=256, additional_cache_name="prompts")
    again = embedder.text_embedder(ids, batch_size=256, additional_cache_name="captions")   # cache hit
    assert np.array_equal(txt, again)
    print(f"embedded {args.images} {h}x{w} images in {t1 - t0:.2f} s ({args.images / (t1 - t0):.0f} img/s incl. host loop and H2D), "
          f"{args.images + args.classes} captions; cache folder {os.environ['PC_CACHE_FOLDER']}")

    target = [labels[i % args.classes] for i in range(args.images)]       # arbitrary targets: random weights know nothing
    _, zs = ZeroShotClassifier(embedder.model.engine).zero_shot_classification(img, cls, labels, target)
    _, rt = ImageRetrieval(embedder.model.engine).retrieval(img, txt)
    print("zero-shot :", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in zs.items() if k in ("Accuracy", "WF1", "mcc", "instances")})
    print("retrieval :", rt)


if __name__ == "__main__":
    main()
