"""Input contract of the path: CLIP image preprocessing and the tokenizer hook.

``_transform`` in reproducibility/embedders/transform.py:45-52 and HF
``CLIPImageProcessor`` (image_processing_pil_clip.py:23-34) are the same pipeline:
resize(shortest edge -> n_px, bicubic) -> center crop n_px -> RGB -> /255 ->
normalise with the CLIP mean/std -> CHW float32.  For tiles that are already
n_px x n_px it reduces exactly to ``(u8/255 - mean)/std``.
"""
from __future__ import annotations

from typing import Callable, List, Sequence, Union

import numpy as np

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)   # transform.py:50
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def preprocess_image(img, n_px: int = 224) -> np.ndarray:
    """PIL image / path / HWC uint8 array -> float32 [3, n_px, n_px]."""
    from PIL import Image
    if isinstance(img, str):
        img = Image.open(img)
    if isinstance(img, np.ndarray):
        img = Image.fromarray(img)
    img = img.convert("RGB")
    w, h = img.size
    if (w, h) != (n_px, n_px):
        short = min(w, h)
        # torchvision Resize(int): shortest edge -> n_px, long edge int(n_px * long / short)
        nw, nh = (n_px, int(n_px * h / w)) if w == short else (int(n_px * w / h), n_px)
        img = img.resize((nw, nh), resample=Image.BICUBIC)
        left, top = int(round((nw - n_px) / 2.0)), int(round((nh - n_px) / 2.0))
        img = img.crop((left, top, left + n_px, top + n_px))
    x = np.asarray(img, dtype=np.float32) / np.float32(255.0)
    x = (x - np.asarray(CLIP_MEAN, dtype=np.float32)) / np.asarray(CLIP_STD, dtype=np.float32)
    return np.ascontiguousarray(x.transpose(2, 0, 1))


def preprocess_images(images: Sequence, n_px: int = 224) -> np.ndarray:
    return np.stack([preprocess_image(i, n_px) for i in images]) if len(images) else \
        np.zeros((0, 3, n_px, n_px), np.float32)


def load_tokenizer(path: str) -> Callable[[List[str], int], "np.ndarray"]:
    """CLIP BPE tokenizer from a local HF model dir (vocab.json + merges.txt).  Returns
    ``fn(texts, context_length) -> (ids int64 [N,ctx], mask int64 [N,ctx])`` padded/truncated
    the way plip.py:57-58 asks (max_length=77, padding="max_length", truncation=True)."""
    from transformers import CLIPTokenizer
    tok = CLIPTokenizer.from_pretrained(path)

    def fn(texts, context_length=77):
        enc = tok(list(texts), return_tensors="np", max_length=context_length, padding="max_length",
                  truncation=True)
        return enc["input_ids"].astype(np.int64), enc["attention_mask"].astype(np.int64)

    return fn
