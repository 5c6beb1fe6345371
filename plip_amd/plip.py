"""Drop-in for the reference's top-level ``PLIP`` class (/root/reference/plip.py).

Same constructor and methods -- ``encode_images``, ``encode_text``,
``_cosine_similarity``, ``_nearest_neighbours``, ``zero_shot_classification``,
``retrieval`` -- with the model forward running in the MI355X engine.  Reference
quirks that are part of its observable behaviour are kept:
``encode_*`` return UN-normalised float32 [N,512] numpy arrays (plip.py:53,71) and
``_cosine_similarity`` normalises only the key side (plip.py:73-76).  Two latent bugs
are not reproduced: ``retrieval`` read a never-assigned ``self.image_vectors``
(plip.py:114) -- here ``index_images`` sets it -- and the batch loop no longer syncs
and copies to the host once per batch (plip.py:50): batches are enqueued back to back
and copied once.

``batch_size`` keeps its place in every signature, but it no longer sizes the ENGINE calls: the reference hard-codes
``batch_size=8`` in ``zero_shot_classification`` / ``retrieval`` (plip.py:90-91,112), a tenth of the engine's bs=256
rate.  A row's embedding is bit-identical whatever batch it travels in (tests/test_gpu_parity.py), so the host loops
prepare the caller's batches one by one (same routing, same preprocessing per batch) and hand them to the towers
``engine.max_batch`` rows at a time.  ``PLIP.coalesce = False`` restores one engine call per caller batch.
"""
from __future__ import annotations

import contextlib

from typing import Callable, List, Optional, Sequence, Union

import numpy as np
import torch

from .model import PlipModel
from .preprocess import load_tokenizer, preprocess_images

# The top-level reference class preprocesses with HF ``CLIPProcessor`` (plip.py:27,35), whose centre crop starts at
# ``(extent - n) // 2``; the reproducibility/ embedders use torchvision's rounding instead (preprocess.crop_offset).
_CROP = "hf"


def _native_u8_tiles(chunk, n_px):
    """uint8 HWC tiles that are already at the model resolution -> one [B,n,n,3] array, else None."""
    if torch.is_tensor(chunk):
        return None
    if isinstance(chunk, np.ndarray):
        ok = chunk.dtype == np.uint8 and chunk.ndim == 4 and chunk.shape[1:] == (n_px, n_px, 3)
        return np.ascontiguousarray(chunk) if ok else None
    arrs = []
    for im in chunk:
        if isinstance(im, np.ndarray) and im.dtype == np.uint8 and im.shape == (n_px, n_px, 3):
            arrs.append(im)
        elif hasattr(im, "size") and hasattr(im, "mode") and im.size == (n_px, n_px):   # PIL image
            arrs.append(np.asarray(im.convert("RGB"), dtype=np.uint8))
        else:
            return None
    return np.stack(arrs) if arrs else None


def _all_native_tiles(chunk, n_px) -> bool:
    """every element an n_px x n_px uint8 RGB array or PIL image (what _native_u8_tiles would stack)"""
    if len(chunk) == 0:
        return False
    for im in chunk:
        if isinstance(im, np.ndarray):
            if im.dtype != np.uint8 or im.shape != (n_px, n_px, 3):
                return False
        elif isinstance(im, str) or not (hasattr(im, "size") and hasattr(im, "mode") and im.size == (n_px, n_px)):
            return False
    return True


def _uniform_u8_images(chunk, n_px):
    """Equally sized uint8 RGB images (arrays or PIL) that still need resize/crop -> one [B,H,W,3] array, else None."""
    if torch.is_tensor(chunk) or isinstance(chunk, np.ndarray) or len(chunk) == 0:
        return None
    arrs, shape = [], None
    for im in chunk:
        if isinstance(im, np.ndarray) and im.dtype == np.uint8 and im.ndim == 3 and im.shape[2] == 3:
            a = im
        elif hasattr(im, "size") and hasattr(im, "mode") and not isinstance(im, str):      # PIL image
            a = np.asarray(im.convert("RGB"), dtype=np.uint8)
        else:
            return None
        if shape is None:
            shape = a.shape
        if a.shape != shape or min(a.shape[:2]) < 8:
            return None
        arrs.append(a)
    return np.stack(arrs)


@contextlib.contextmanager
def _lane_loop(eng):
    """``Engine.lane_loop`` where the engine has one (a PlipModel's); any other engine object: every call on it, in order."""
    loop = getattr(eng, "lane_loop", None)
    if loop is None:
        yield lambda fn: fn(eng)
    else:
        with loop() as run:
            yield run


class PLIP:
    coalesce = True          # engine calls carry up to engine.max_batch rows whatever ``batch_size`` says (module docstring)

    def __init__(self, model_name: str = None, auth_token=None, *, model: Optional[PlipModel] = None,
                 tokenizer: Optional[Callable] = None, tokenizer_dir: Optional[str] = None, dtype: str = "bf16",
                 max_batch: int = 256, device: str = "cuda:0", pack_captions: bool = False, text_f16: bool = False,
                 text_f16_layers: Optional[int] = None):
        """``model_name``: local HF directory (what ``CLIPModel/CLIPProcessor.from_pretrained`` take, plip.py:26-27)
        or an OpenAI-clip ``.pt`` state dict.  The tokenizer comes from ``tokenizer`` (a callable), else from
        ``tokenizer_dir`` / the model directory when it holds ``vocab.json`` + ``merges.txt``; with neither,
        ``encode_text`` still takes token ids.  ``pack_captions`` (extension, 16-bit engines): the text tower computes only
        the positions up to each caption's EOS token -- bit-identical embeddings, cost proportional to the caption lengths
        instead of the padded 77 (include/plipmi.h ``plipmi_set_text_packing``).  ``dtype``: "bf16" | "f16" | "f32";
        ``text_f16`` (bf16 engine): the text tower on IEEE-half operands (PLIPMI_FLAG_TEXT_TOWER_F16); ``text_f16_layers``
        (bf16 engine): only that many leading text blocks (plipmi_config.text_f16_layers; None = the default).  The engine's other
        per-handle options (ln_fold, pooled_last_block, mfma_attention, graph_batch) are reached by building the model
        explicitly -- ``PLIP(model=PlipModel.from_pretrained(path, **engine_options))``."""
        if not torch.cuda.is_available():
            raise RuntimeError("plip_amd.PLIP needs an MI355X (ROCm) GPU; there is no CPU path")
        self.device = device
        self.model_name = model_name
        if model is None and model_name is None:
            raise ValueError("give a local checkpoint path (HF dir or OpenAI .pt) or a PlipModel")
        if tokenizer is None:       # resolved BEFORE the engine is built: a bad tokenizer dir must not leak a handle
            import os
            cand = tokenizer_dir or (model_name if model_name and os.path.isdir(model_name) else None)
            if tokenizer_dir is not None and not os.path.exists(os.path.join(tokenizer_dir, "vocab.json")):
                raise FileNotFoundError(f"tokenizer_dir {tokenizer_dir!r} has no vocab.json / merges.txt")
            if cand and os.path.exists(os.path.join(cand, "vocab.json")) and os.path.exists(os.path.join(cand, "merges.txt")):
                tokenizer = load_tokenizer(cand)
        if model is None:
            model = PlipModel.from_pretrained(model_name, device=device, dtype=dtype, max_batch=max_batch, text_f16=text_f16,
                                              text_f16_layers=text_f16_layers)
        self.model = model.to(self.device)
        if pack_captions:
            self.model.engine.set_text_packing(True)
        self.tokenizer = tokenizer
        self.model_hash = hash            # the reference returns the builtin too (plip.py:29)
        self.image_vectors = None

    # -- plip.py:31-53 -------------------------------------------------------
    def encode_images(self, images: Union[List[str], list, np.ndarray, torch.Tensor], batch_size: int,
                      num_workers: int = 0):
        """``num_workers > 0`` (extension): decode / resize with a thread pool and stream batches through pinned
        double buffers so host work and H2D overlap the towers (plip_amd/pipeline.py); results are identical."""
        n_px = self.model.config.image_size
        if num_workers > 0 and not torch.is_tensor(images) and not isinstance(images, np.ndarray) and len(images):
            return self._encode_images_pipelined(list(images), batch_size, num_workers)
        eng = self.model.engine
        cap = max(int(batch_size), int(getattr(eng, "max_batch", batch_size))) if self.coalesce else int(batch_size)
        outs, pend, kind, rows = [], [], None, 0

        def flush():
            nonlocal pend, kind, rows
            if kind == "stage":            # native tiles: each copied ONCE, into the pinned staging rows; one H2D, one engine call
                stage = self._fill_stage(pend, n_px, cap)       # (the H2D copy inside the call is synchronous: the rows may be refilled)
                outs.append(run(lambda e: e.encode_image_u8(stage)))
            elif pend:
                if len(pend) > 1 and any(t.is_cuda for t in pend):
                    pend = [t.to(eng.device) for t in pend]
                t = pend[0] if len(pend) == 1 else torch.cat(pend)
                outs.append(run((lambda e: e.encode_image_u8(t)) if kind == "tiles" else (lambda e: e.encode_image(t, normalize=False))))
            pend, kind, rows = [], None, 0

        # consecutive engine calls of the loop alternate between the engine and a clone on a second stream (Engine.lane_loop): one
        # batch's launch boundaries and pooled tail run under the next batch's GEMMs; the rows' bits do not depend on the lane
        with torch.no_grad(), _lane_loop(eng) as run:
            for s in range(0, len(images), batch_size):
                chunk = images[s:s + batch_size]
                if isinstance(chunk, (list, tuple)) and any(isinstance(c, str) for c in chunk):
                    from PIL import Image                                  # plip.py:34 opens the paths of a batch
                    chunk = [Image.open(c) if isinstance(c, str) else c for c in chunk]
                if self.coalesce and isinstance(chunk, (list, tuple)) and len(chunk) <= cap:
                    # already n_px x n_px uint8 (normalised on the GPU, fused into the unfold): each tile is copied ONCE, into
                    # a pinned staging buffer of max_batch rows -- 32 small np.stack calls + a torch.cat + a pageable H2D
                    # cost more than the towers (profiles/r04_small_batch_latency.txt)
                    if _all_native_tiles(chunk, n_px):
                        if kind is not None and (kind != "stage" or rows + len(chunk) > cap):
                            flush()
                        pend.extend(chunk)
                        kind, rows = "stage", rows + len(chunk)
                        continue
                tiles = _native_u8_tiles(chunk, n_px)
                same = _uniform_u8_images(chunk, n_px) if tiles is None else None
                if tiles is not None:      # already n_px x n_px uint8: normalise on the GPU, fused into the unfold
                    k, t = "tiles", torch.from_numpy(tiles)
                elif same is not None:     # one size, not the model's: Pillow-exact resize + crop on the GPU as well
                    k, t = "tiles", eng.resize_crop_u8(torch.from_numpy(same), crop=_CROP)
                elif torch.is_tensor(chunk):
                    k, t = "pixels", chunk
                elif isinstance(chunk, np.ndarray) and chunk.dtype != np.uint8:
                    k, t = "pixels", torch.from_numpy(chunk)
                else:
                    k, t = "pixels", torch.from_numpy(preprocess_images(list(chunk), n_px, crop=_CROP))
                if kind is not None and (k != kind or t.dtype != pend[0].dtype or rows + t.shape[0] > cap):
                    flush()
                pend.append(t)
                kind, rows = k, rows + t.shape[0]
            flush()
        if not outs:
            return np.zeros((0, self.model.config.projection_dim), np.float32)
        return torch.cat(outs).detach().cpu().numpy()

    _stage = None            # pinned uint8 [max_batch, n, n, 3] staging rows of the coalescing host loop (allocated on first use)

    def _fill_stage(self, tiles, n_px, cap) -> torch.Tensor:
        """The native tiles of one engine call (arrays or PIL) -> the first len(tiles) rows of the pinned staging buffer.
        (Plain row copies: a thread pool was measured 3x slower -- numpy keeps the GIL for 150 KB copies.)"""
        if self._stage is None or self._stage.shape[0] < cap or self._stage.shape[1] != n_px:
            self._stage = torch.empty((cap, n_px, n_px, 3), dtype=torch.uint8, pin_memory=torch.cuda.is_available())
        dst = self._stage.numpy()
        for i, im in enumerate(tiles):
            dst[i] = im if isinstance(im, np.ndarray) else np.asarray(im.convert("RGB"), dtype=np.uint8)
        return self._stage[:len(tiles)]

    def _encode_images_pipelined(self, images: list, batch_size: int, num_workers: int):
        """Same routing per batch as the loop above (raw tiles / GPU resize / host Pillow), one batch ahead."""
        from .pipeline import run_batches
        from .preprocess import preprocess_image
        n_px = self.model.config.image_size
        eng = self.model.engine

        def decode(im):            # paths are opened here, i.e. on the worker threads
            if isinstance(im, str):
                from PIL import Image
                im = Image.open(im)
                im.load()
            return im

        def prepare_batch(chunk, pool):
            chunk = list(pool.map(decode, chunk)) if pool is not None else [decode(c) for c in chunk]
            tiles = _native_u8_tiles(chunk, n_px)
            if tiles is not None:
                return "tiles", tiles
            same = _uniform_u8_images(chunk, n_px)
            if same is not None:
                return "resize", same
            one = lambda im: preprocess_image(im, n_px, _CROP)
            arrs = list(pool.map(one, chunk)) if pool is not None else [one(c) for c in chunk]
            return "pixels", np.stack(arrs)

        def consume(tag, t, e=eng):
            if tag == "tiles":
                return e.encode_image_u8(t)
            if tag == "resize":
                return e.encode_image_u8(e.resize_crop_u8(t, crop=_CROP))
            return e.encode_image(t, normalize=False)

        bs = min(int(batch_size), eng.max_batch)
        lanes = eng.lanes() if getattr(eng, "use_lanes", False) and hasattr(eng, "lanes") else None
        with torch.no_grad():
            outs = run_batches(images, bs, None, consume, device=eng.device, num_workers=num_workers,
                               prepare_batch=prepare_batch, lanes=lanes)
        return torch.cat(outs).detach().cpu().numpy()

    # -- plip.py:55-71 ---------------------------------------------------------
    def encode_text(self, text: Union[List[str], np.ndarray, torch.Tensor], batch_size: int):
        ctx = self.model.config.context_length
        if torch.is_tensor(text) or isinstance(text, np.ndarray):
            ids, mask = torch.as_tensor(text), None      # already tokenised
        else:
            if self.tokenizer is None:
                raise RuntimeError("no tokenizer: pass tokenizer=... or token ids")
            ids, mask = self.tokenizer(list(text), ctx)
            ids, mask = torch.as_tensor(ids), (None if mask is None else torch.as_tensor(mask))
        outs = []
        step = int(batch_size)
        if self.coalesce:      # the captions are tokenised already: only the size of the engine calls changes
            step = max(step, int(getattr(self.model.engine, "max_batch", step)))
        with torch.no_grad(), _lane_loop(self.model.engine) as run:
            for s in range(0, len(ids), step):
                m = None if mask is None else mask[s:s + step]
                outs.append(run(lambda e, s=s, m=m: e.encode_text(ids[s:s + step], m, normalize=False)))
        if not outs:
            return np.zeros((0, self.model.config.projection_dim), np.float32)
        return torch.cat(outs).detach().cpu().numpy()

    # -- plip.py:73-76 -----------------------------------------------------------
    def _cosine_similarity(self, key_vectors: np.ndarray, space_vectors: np.ndarray, normalize=True):
        eng = self.model.engine
        k = torch.as_tensor(np.ascontiguousarray(key_vectors, dtype=np.float32)).to(eng.device)
        s = torch.as_tensor(np.ascontiguousarray(space_vectors, dtype=np.float32)).to(eng.device)
        if normalize:
            k = eng.l2_normalize_(k.clone())            # only the key side, as the reference does
        lpi, _, _ = eng.logits(k, s, scale=1.0, want_text=False)
        return lpi.cpu().numpy()

    # -- plip.py:78-87 -----------------------------------------------------------
    def _nearest_neighbours(self, k, key_vectors, space_vectors, normalize=True, debug=False):
        eng = self.model.engine
        key_vectors, space_vectors = np.asarray(key_vectors), np.asarray(space_vectors)
        # argsort()[:, -k:] (plip.py:84) is a SLICE: it hands back every column when k exceeds the corpus, every column for
        # k = 0 ([-0:] is [0:]), and for k < 0 it drops the |k| weakest columns ([|k|:] of the ascending order), i.e. keeps
        # the n - |k| best.  Same counts here instead of failing.
        n_space = space_vectors.shape[0]
        k = int(k)
        k = n_space if k == 0 else (max(n_space + k, 0) if k < 0 else min(k, n_space))
        if k == 0 or key_vectors.shape[0] == 0:
            return np.zeros((key_vectors.shape[0], k), np.int64)
        kv = torch.as_tensor(np.ascontiguousarray(key_vectors, dtype=np.float32)).to(eng.device)
        sv = torch.as_tensor(np.ascontiguousarray(space_vectors, dtype=np.float32)).to(eng.device)
        if normalize:
            kv = eng.l2_normalize_(kv.clone())          # only the key side, as the reference does (plip.py:74-76)
        if debug:
            print(self._cosine_similarity(key_vectors, space_vectors, normalize=normalize))
        if kv.shape[1] % 32 == 0 and k <= 1024:
            return eng.similarity_topk(kv, sv, k).cpu().numpy()     # fused: no [Nq, Ns] matrix
        lpi, _, _ = eng.logits(kv, sv, scale=1.0, want_text=False)
        return eng.topk(lpi, k).cpu().numpy()

    # -- plip.py:89-103 ----------------------------------------------------------
    def zero_shot_classification(self, images, text_labels: List[str], debug=False):
        text_vectors = self.encode_text(text_labels, batch_size=8)
        image_vectors = self.encode_images(images, batch_size=8)
        cosine_sim = self._cosine_similarity(image_vectors, text_vectors)
        if debug:
            print(cosine_sim)
        preds = np.argmax(cosine_sim, axis=-1)
        return [text_labels[idx] for idx in preds]

    def index_images(self, images, batch_size: int = 256):
        """Embed the retrieval corpus (the reference forgot to: plip.py:114 reads self.image_vectors)."""
        self.image_vectors = self.encode_images(images, batch_size=batch_size)
        return self.image_vectors

    # -- plip.py:105-114 -----------------------------------------------------------
    def retrieval(self, queries: List[str], top_k: int = 10):
        if self.image_vectors is None:
            raise RuntimeError("call index_images(images) before retrieval()")
        text_vectors = self.encode_text(queries, batch_size=8)
        return self._nearest_neighbours(k=top_k, key_vectors=text_vectors, space_vectors=self.image_vectors)
