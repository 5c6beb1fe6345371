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


def crop_offset(extent: int, n_px: int, rule: str = "torchvision") -> int:
    """First row/column of the centre crop.  The two preprocessing front ends of the reference disagree by one
    pixel when the excess is odd: torchvision ``CenterCrop`` (OpenAI ``_transform``,
    reproducibility/embedders/transform.py:47) uses ``int(round((extent - n) / 2.0))`` (round-half-to-even), HF
    ``CLIPImageProcessor.center_crop`` behind ``self.preprocess`` (plip.py:27,35) uses ``(extent - n) // 2``."""
    if rule == "hf":
        return (extent - n_px) // 2
    if rule == "torchvision":
        return int(round((extent - n_px) / 2.0))
    raise ValueError(f"unknown crop rule {rule!r} (expected 'hf' or 'torchvision')")


def preprocess_image(img, n_px: int = 224, crop: str = "torchvision") -> np.ndarray:
    """PIL image / path / HWC uint8 array -> float32 [3, n_px, n_px].  ``crop``: see :func:`crop_offset`
    ("torchvision" = ``_transform`` of reproducibility/, "hf" = the ``CLIPProcessor`` of the top-level PLIP)."""
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
        left, top = crop_offset(nw, n_px, crop), crop_offset(nh, n_px, crop)
        img = img.crop((left, top, left + n_px, top + n_px))
    x = np.asarray(img, dtype=np.float32) / np.float32(255.0)
    x = (x - np.asarray(CLIP_MEAN, dtype=np.float32)) / np.asarray(CLIP_STD, dtype=np.float32)
    return np.ascontiguousarray(x.transpose(2, 0, 1))


# ---------------------------------------------------------------------------------------------------------------
# GPU-side resize + centre crop.  Pillow's ``Image.resize(BICUBIC)`` on 8-bit images is
# integer arithmetic once its coefficient tables exist (libImaging/Resample.c: precompute_coeffs,
# normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc / Vertical_8bpc): per output pixel a window [xmin, xmin+n) of
# input pixels, weights = bicubic(a = -0.5) sampled at the window's pixel centres, normalised, rounded to 22-bit fixed
# point; each pass accumulates int32 from 2^21 and shifts right by 22 with a clamp to 0..255, horizontal pass first,
# uint8 in between.  The tables are built here in float64 exactly as published; the two integer passes run in
# libplipmi.so (plipmi_resize_crop_u8) and are therefore bit-identical to Pillow (tests/test_host.py emulates the
# passes in numpy against Pillow itself, tests/test_gpu_api.py checks the kernel against Pillow).
# ---------------------------------------------------------------------------------------------------------------
_PRECISION_BITS = 32 - 8 - 2


def _bicubic(x: np.ndarray) -> np.ndarray:
    a = -0.5
    x = np.abs(x)
    return np.where(x < 1.0, ((a + 2.0) * x - (a + 3.0)) * x * x + 1.0,
                    np.where(x < 2.0, (((x - 5.0) * x + 8.0) * x - 4.0) * a, 0.0))


def resample_coeffs(in_size: int, out_size: int):
    """Pillow's 8-bit bicubic tables for one axis: ``bounds`` int32 [out, 2] = (first input index, tap count) and
    ``kk`` int32 [out, ksize] fixed-point weights (zero beyond the tap count)."""
    scale = float(np.float32(in_size) - np.float32(0.0)) / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)           # C (int) cast: truncation of a non-negative value
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = _bicubic((np.arange(xmax, dtype=np.float64) + xmin - center + 0.5) * ss)
        ww = w.sum()
        if ww != 0.0:
            w = w / ww
        fixed = np.where(w < 0, -0.5 + w * (1 << _PRECISION_BITS), 0.5 + w * (1 << _PRECISION_BITS))
        kk[xx, :xmax] = np.trunc(fixed).astype(np.int64).astype(np.int32)
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def resize_crop_plan(w: int, h: int, n_px: int = 224, crop: str = "torchvision"):
    """Everything ``plipmi_resize_crop_u8`` needs for [h, w, 3] uint8 images: torchvision ``Resize(n_px)`` geometry
    (shortest edge -> n_px, long edge ``int(n_px * long / short)``; HF's ``get_resize_output_image_size`` gives the
    same numbers), centre-crop offsets by ``crop`` rule (:func:`crop_offset`), and the two coefficient tables
    restricted to the crop window.  ``None`` entries mean that pass is an identity (Pillow skips it too)."""
    short = min(w, h)
    nw, nh = (n_px, int(n_px * h / w)) if w == short else (int(n_px * w / h), n_px)
    if nw < n_px or nh < n_px:
        raise ValueError(f"image {w}x{h} is too small to crop {n_px}x{n_px} after the resize")
    left, top = crop_offset(nw, n_px, crop), crop_offset(nh, n_px, crop)
    plan = dict(w=w, h=h, n_px=n_px, nw=nw, nh=nh, left=left, top=top, xb=None, xk=None, yb=None, yk=None)
    if nw != w:
        b, k = resample_coeffs(w, nw)
        plan["xb"], plan["xk"] = np.ascontiguousarray(b[left:left + n_px]), np.ascontiguousarray(k[left:left + n_px])
    if nh != h:
        b, k = resample_coeffs(h, nh)
        plan["yb"], plan["yk"] = np.ascontiguousarray(b[top:top + n_px]), np.ascontiguousarray(k[top:top + n_px])
    return plan


def resize_crop_reference(img_u8: np.ndarray, plan) -> np.ndarray:
    """numpy emulation of the two integer passes (what the HIP kernels do); [h, w, 3] uint8 -> [n, n, 3] uint8."""
    n = plan["n_px"]
    src = img_u8.astype(np.int64)
    half = 1 << (_PRECISION_BITS - 1)
    if plan["xb"] is not None:
        tmp = np.empty((src.shape[0], n, 3), np.int64)
        for x in range(n):
            x0, cnt = plan["xb"][x]
            acc = (src[:, x0:x0 + cnt, :] * plan["xk"][x, :cnt].astype(np.int64)[None, :, None]).sum(axis=1) + half
            tmp[:, x, :] = np.clip(acc >> _PRECISION_BITS, 0, 255)
    else:
        tmp = src[:, plan["left"]:plan["left"] + n, :]
    if plan["yb"] is not None:
        out = np.empty((n, n, 3), np.int64)
        for y in range(n):
            y0, cnt = plan["yb"][y]
            acc = (tmp[y0:y0 + cnt, :, :] * plan["yk"][y, :cnt].astype(np.int64)[:, None, None]).sum(axis=0) + half
            out[y] = np.clip(acc >> _PRECISION_BITS, 0, 255)
    else:
        out = tmp[plan["top"]:plan["top"] + n]
    return out.astype(np.uint8)


def preprocess_images(images: Sequence, n_px: int = 224, crop: str = "torchvision") -> np.ndarray:
    return np.stack([preprocess_image(i, n_px, crop) for i in images]) if len(images) else \
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
