"""``CLIPEmbedder`` / ``EmbedderFactory`` on the MI355X engine.

Mirrors reproducibility/embedders/plip.py:10-75 and reproducibility/embedders/factory.py:10-32: the same
constructor, the same four methods, the same cache behaviour (image side: raw-file-name scheme, text side:
sha256 scheme), the same return value -- ``np.ndarray [N, 512]`` with unit-norm rows.  Differences, all on the
fast side of the boundary: batches go through ``PlipModel.encode_image/encode_text`` (libplipmi.so), the row
normalisation runs on the GPU before the copy back (one fp32 ``x / ||x||`` per row, as :53/:73 do on the host),
and images are decoded on the host by ``preprocess`` exactly once per item (no DataLoader worker processes).
"""
from __future__ import annotations

import os
from typing import Callable, Optional, Sequence

import numpy as np
import torch

from ..model import PlipModel
from ..preprocess import preprocess_image
from . import cacher


class CLIPEmbedder:
    def __init__(self, model: PlipModel, preprocess: Optional[Callable] = None, name: str = "plip",
                 backbone: str = "", tokenizer: Optional[Callable] = None):
        self.model = model
        self.preprocess = preprocess if preprocess is not None else \
            (lambda img: preprocess_image(img, model.config.image_size))
        self.name = name
        self.backbone = backbone
        self.tokenizer = tokenizer      # fn(list[str], context_length) -> ids | (ids, mask); stands in for clip.tokenize

    # -- cached entry points (embedders/plip.py:18-38) ---------------------------------------------------
    def image_embedder(self, list_of_images, device="cuda", num_workers=1, batch_size=32, additional_cache_name=""):
        key = self.name + "img" + additional_cache_name
        hit = cacher.cache_hit_or_miss_raw_filename(key, self.backbone)
        if hit is not None:
            return hit
        hit = self.embed_images(list_of_images, device=device, num_workers=num_workers, batch_size=batch_size)
        cacher.cache_numpy_object_raw_filename(hit, key, self.backbone)
        return hit

    def text_embedder(self, list_of_labels, device="cuda", num_workers=1, batch_size=32, additional_cache_name=""):
        key = self.name + "txt" + additional_cache_name
        hit = cacher.cache_hit_or_miss(key, self.backbone)
        if hit is not None:
            return hit
        hit = self.embed_text(list_of_labels, device=device, num_workers=num_workers, batch_size=batch_size)
        cacher.cache_numpy_object(hit, key, self.backbone)
        return hit

    # -- the towers (embedders/plip.py:40-75) --------------------------------------------------------------
    def _chunks(self, n: int, batch_size: int):
        step = max(1, min(int(batch_size), self.model.engine.max_batch))
        return [(i, min(i + step, n)) for i in range(0, n, step)]

    @torch.no_grad()
    def embed_images(self, list_of_images, device="cuda", num_workers=1, batch_size=32) -> np.ndarray:
        eng = self.model.engine
        if num_workers and num_workers > 1 and not torch.is_tensor(list_of_images) and not (
                isinstance(list_of_images, np.ndarray) and list_of_images.dtype != object and list_of_images.ndim == 4):
            from ..pipeline import run_batches          # decode on a thread pool, H2D on a copy stream
            step = max(1, min(int(batch_size), eng.max_batch))
            outs = run_batches(list(list_of_images), step, lambda i: np.asarray(self.preprocess(i), dtype=np.float32),
                               lambda t, e=eng: e.encode_image(t, normalize=True), device=eng.device, num_workers=num_workers,
                               lanes=eng.lanes() if eng.use_lanes else None)     # consecutive batches on two engines / streams
            return torch.cat(outs).cpu().numpy() if outs else np.zeros((0, self.model.config.projection_dim), np.float32)
        out = []
        with eng.lane_loop() as run:       # consecutive batches alternate between the engine and a clone on a second stream
            for a, b in self._chunks(len(list_of_images), batch_size):
                items = list_of_images[a:b]
                if torch.is_tensor(items):
                    px = items.to(dtype=torch.float32)
                elif isinstance(items, np.ndarray) and items.dtype != object and items.ndim == 4 and items.shape[1] == 3:
                    px = torch.from_numpy(np.ascontiguousarray(items, dtype=np.float32))     # already preprocessed NCHW
                else:
                    px = torch.from_numpy(np.stack([np.asarray(self.preprocess(i), dtype=np.float32) for i in items]))
                out.append(run(lambda e, px=px: e.encode_image(px, normalize=True)))       # :48 encode_image + :53 row normalisation
        if not out:
            return np.zeros((0, self.model.config.projection_dim), np.float32)
        return torch.cat(out).cpu().numpy()

    def _tokenize(self, captions):
        if torch.is_tensor(captions):
            return captions.to(torch.int64)
        arr = np.asarray(captions)
        if arr.dtype.kind in "iu":
            return torch.from_numpy(arr.astype(np.int64))
        if self.tokenizer is None:
            raise RuntimeError("captions are strings but no tokenizer was given: pass tokenizer= (e.g. "
                               "plip_amd.preprocess.load_tokenizer(<dir with vocab.json/merges.txt>)) or token ids")
        ids = self.tokenizer(list(captions), self.model.config.context_length)
        if isinstance(ids, tuple):
            ids = ids[0]
        return torch.as_tensor(np.asarray(ids)).to(torch.int64)

    @torch.no_grad()
    def embed_text(self, list_of_labels, device="cuda", num_workers=1, batch_size=32) -> np.ndarray:
        eng = self.model.engine
        out = []
        with eng.lane_loop() as run:
            for a, b in self._chunks(len(list_of_labels), batch_size):
                ids = self._tokenize(list_of_labels[a:b])
                # clip.tokenize pads with 0 and the OpenAI model pools at argmax(ids): eos_token_id < 0 selects that rule
                out.append(run(lambda e, ids=ids: e.encode_text(ids, None, normalize=True, eos_token_id=-1)))      # :66 + :73
        if not out:
            return np.zeros((0, self.model.config.projection_dim), np.float32)
        return torch.cat(out).cpu().numpy()


class EmbedderFactory:
    """``EmbedderFactory().factory(args)`` with ``args.model_name`` in {"plip", "clip"} and ``args.backbone`` = a local
    checkpoint (OpenAI-clip ``.pt`` state dict, as factory.py:21-25 loads, or an HF directory / .safetensors).  The
    architecture comes from ``$PC_CLIP_ARCH`` (factory.py:21), default ViT-B/32.  Optional ``args.dtype`` ("bf16" |
    "fp32"), ``args.max_batch`` and ``args.tokenizer_dir`` are extensions."""

    def factory(self, args) -> CLIPEmbedder:
        name = args.model_name
        path = getattr(args, "backbone", None)
        arch = os.environ.get("PC_CLIP_ARCH", "ViT-B/32")
        if name in ("plip", "clip"):
            if not path or not os.path.exists(path):
                raise FileNotFoundError(
                    f"model_name={name!r} needs args.backbone to be a local checkpoint (got {path!r}); "
                    "this engine never downloads weights")
            model = PlipModel.from_pretrained(path, arch=arch, dtype=getattr(args, "dtype", "bf16"),
                                              max_batch=int(getattr(args, "max_batch", 256)))
            tok = None
            tok_dir = getattr(args, "tokenizer_dir", None) or (path if os.path.isdir(path) else None)
            if tok_dir and os.path.exists(os.path.join(tok_dir, "vocab.json")):
                from ..preprocess import load_tokenizer
                tok = load_tokenizer(tok_dir)
            return CLIPEmbedder(model.eval(), None, name, path, tokenizer=tok)
        if name == "mudipath":
            raise NotImplementedError("the MuDiPath DenseNet baseline (factory.py:34-47) is outside the PLIP hot path")
        raise ValueError(f"unknown model_name {name!r}")
