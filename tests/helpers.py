"""Shared test scaffolding: a synthetic CLIP tokenizer / HF checkpoint directory written to tmp dirs, an
oracle-backed CPU stand-in for the engine (host-loop tests without a GPU), and the loader of the reference's own
``plip.py`` (only where /root/reference is mounted -- never on the GPU box).  TEST INFRASTRUCTURE."""
from __future__ import annotations

import importlib.util
import json
import os
import sys

import numpy as np
import torch

REFERENCE_ROOT = "/root/reference"


# ---------------------------------------------------------------------------------------------------------------
# tokenizer + checkpoint fixtures
# ---------------------------------------------------------------------------------------------------------------
def write_tokenizer_fixture(path, bos_id: int = 510, eos_id: int = 511) -> int:
    """``vocab.json`` + ``merges.txt`` of a ~100-entry byte-level BPE in CLIP's format (characters, characters with
    the ``</w>`` end-of-word suffix, a few merges, the two special tokens at the given ids).  No real CLIP vocabulary
    is on disk (no network); this makes ``CLIPTokenizer`` constructible offline.  Returns the vocabulary size."""
    chars = list("abcdefghijklmnopqrstuvwxyz0123456789.,!?'-")
    vocab = {}
    for c in chars:
        vocab[c] = len(vocab)
    for c in chars:
        vocab[c + "</w>"] = len(vocab)
    merges = []

    def add(a, b):
        merges.append(f"{a} {b}")
        vocab[a + b] = len(vocab)
    add("t", "u"); add("tu", "m"); add("o", "r</w>"); add("tum", "or</w>")          # tumor
    add("c", "e"); add("l", "l</w>"); add("ce", "ll</w>")                             # cell
    add("t", "h"); add("th", "e</w>"); add("o", "f</w>"); add("a", "n</w>")           # the, of, an
    add("i", "m"); add("a", "g"); add("im", "ag"); add("imag", "e</w>")               # image
    assert len(vocab) <= min(bos_id, eos_id)
    vocab["<|startoftext|>"] = bos_id
    vocab["<|endoftext|>"] = eos_id
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "vocab.json"), "w") as f:
        json.dump(vocab, f)
    with open(os.path.join(path, "merges.txt"), "w") as f:
        f.write("#version: 0.2\n" + "\n".join(merges) + "\n")
    return len(vocab)


def hf_config_dict(cfg) -> dict:
    return {"projection_dim": cfg.projection_dim, "logit_scale_init_value": cfg.logit_scale_init,
            "text_config": {"vocab_size": cfg.vocab_size, "hidden_size": cfg.t_width, "intermediate_size": cfg.t_mlp,
                            "num_hidden_layers": cfg.t_layers, "num_attention_heads": cfg.t_heads,
                            "max_position_embeddings": cfg.context_length, "eos_token_id": cfg.eos_token_id,
                            "bos_token_id": cfg.bos_token_id},
            "vision_config": {"hidden_size": cfg.v_width, "intermediate_size": cfg.v_mlp,
                              "num_hidden_layers": cfg.v_layers, "num_attention_heads": cfg.v_heads,
                              "image_size": cfg.image_size, "patch_size": cfg.patch_size,
                              "layer_norm_eps": cfg.layer_norm_eps}}


def write_hf_model_dir(path, cfg, sd, with_tokenizer: bool = True):
    """A local directory shaped like ``vinid/plip`` on the hub: config.json + model.safetensors (+ tokenizer files)."""
    from safetensors.numpy import save_file
    os.makedirs(path, exist_ok=True)
    save_file({k: np.ascontiguousarray(np.asarray(v, dtype=np.float32)) for k, v in sd.items()},
              os.path.join(path, "model.safetensors"))
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(hf_config_dict(cfg), f)
    if with_tokenizer:
        write_tokenizer_fixture(path, cfg.bos_token_id, cfg.eos_token_id)
    return str(path)


# ---------------------------------------------------------------------------------------------------------------
# oracle-backed engine stand-in (CPU): the same method surface plip_amd.engine.Engine offers to the host layer
# ---------------------------------------------------------------------------------------------------------------
class OracleEngine:
    """Computes with oracle/clip_oracle.py on the CPU what ``Engine`` computes in libplipmi.so, so the HOST logic
    (batch loops, routing of tiles / resizes / pixels, the reference's own loops) can be tested without a GPU."""

    def __init__(self, cfg, sd, max_batch: int = 8):
        from oracle import clip_oracle as O
        self.O, self.cfg, self.sd, self.max_batch = O, cfg, sd, max_batch
        self.device = torch.device("cpu")
        self.logit_scale = float(sd["logit_scale"])
        self.calls = []

    @property
    def logit_scale_exp(self):
        return float(np.exp(self.logit_scale))

    def _norm(self, e, normalize):
        return self.O.l2_normalize(e) if normalize else e

    def encode_image(self, pixels, normalize=False):
        self.calls.append(("encode_image", tuple(pixels.shape)))
        px = np.asarray(pixels.detach().cpu().numpy() if torch.is_tensor(pixels) else pixels, np.float32)
        if px.shape[0] == 0:
            return torch.zeros((0, self.cfg.projection_dim))
        return torch.from_numpy(self._norm(self.O.vision_tower(px, self.sd, self.cfg), normalize).astype(np.float32))

    def encode_image_u8(self, tiles, normalize=False):
        from plip_amd.preprocess import CLIP_MEAN, CLIP_STD
        self.calls.append(("encode_image_u8", tuple(tiles.shape)))
        x = tiles.numpy().astype(np.float32) / np.float32(255.0)
        x = (x - np.asarray(CLIP_MEAN, np.float32)) / np.asarray(CLIP_STD, np.float32)
        return self.encode_image(torch.from_numpy(np.ascontiguousarray(x.transpose(0, 3, 1, 2))), normalize)

    def resize_crop_u8(self, images_u8, crop="torchvision"):
        from plip_amd.preprocess import resize_crop_plan, resize_crop_reference
        self.calls.append(("resize_crop_u8", tuple(images_u8.shape), crop))
        a = images_u8.numpy()
        plan = resize_crop_plan(a.shape[2], a.shape[1], self.cfg.image_size, crop)
        return torch.from_numpy(np.stack([resize_crop_reference(im, plan) for im in a]))

    def encode_text(self, input_ids, attention_mask=None, normalize=False, eos_token_id=None):
        self.calls.append(("encode_text", tuple(input_ids.shape)))
        ids = np.asarray(input_ids.detach().cpu().numpy() if torch.is_tensor(input_ids) else input_ids)
        if ids.shape[0] == 0:
            return torch.zeros((0, self.cfg.projection_dim))
        cfg = self.cfg if eos_token_id is None else self.cfg.replace(eos_token_id=int(eos_token_id))
        m = None if attention_mask is None else np.asarray(attention_mask)
        return torch.from_numpy(self._norm(self.O.text_tower(ids, self.sd, cfg, m), normalize).astype(np.float32))

    def l2_normalize_(self, x):
        x /= x.norm(dim=-1, keepdim=True)
        return x

    def logits(self, a, b, scale=1.0, want_text=True, want_argmax=False):
        l = (scale * a.float() @ b.float().T)
        return l, (l.T.contiguous() if want_text else None), (l.argmax(1).int() if want_argmax else None)

    def similarity_topk(self, keys, space, k):
        return torch.argsort(-(keys @ space.T), dim=1, stable=True)[:, :k]

    def topk(self, scores, k):
        return torch.argsort(-scores, dim=1, stable=True)[:, :k]


def oracle_model(cfg, sd, max_batch: int = 8):
    """A real ``plip_amd.model.PlipModel`` (its host-side surface is what is under test) on an ``OracleEngine``."""
    from plip_amd.model import PlipModel
    m = object.__new__(PlipModel)
    m.config = cfg
    m.engine = OracleEngine(cfg, sd, max_batch)
    m.device = torch.device("cpu")
    m.dtype = torch.float32
    m.training = False
    m.logit_scale = torch.tensor(m.engine.logit_scale, dtype=torch.float32)
    m.to = lambda *a, **k: m            # the GPU class refuses non-cuda devices; the stand-in lives on the CPU
    return m


# ---------------------------------------------------------------------------------------------------------------
# the reference's own host code (read-only tree; absent on the GPU box)
# ---------------------------------------------------------------------------------------------------------------
def reference_available() -> bool:
    return os.path.exists(os.path.join(REFERENCE_ROOT, "plip.py"))


def load_reference_plip_module():
    """Import /root/reference/plip.py by path WITHOUT writing a __pycache__ into the read-only tree."""
    old = sys.dont_write_bytecode
    sys.dont_write_bytecode = True
    try:
        spec = importlib.util.spec_from_file_location("reference_plip", os.path.join(REFERENCE_ROOT, "plip.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    finally:
        sys.dont_write_bytecode = old


# ---------------------------------------------------------------------------------------------------------------
# the reference's host pattern, restated (the GPU box has no /root/reference)
# ---------------------------------------------------------------------------------------------------------------
class ReferenceHostLoops:
    """What /root/reference/plip.py:31-103 does on the host, restated so that it can run where the reference tree is not
    mounted: ``datasets.Dataset`` -> ``set_transform`` / ``map`` -> ``DataLoader`` -> ``self.model.get_*_features(**batch)``
    -> ``.detach().cpu().numpy()`` -> ``np.stack``; key-side-only normalisation (:73-76); arg-max labels (:89-103).
    tests/test_reference_loop.py asserts, on the CPU box where both exist, that this produces the very arrays the imported
    original produces on the same model -- so driving THIS against libplipmi.so on the GPU box exercises surface B1
     the way the reference's own loop does."""

    def __init__(self, model, processor, device):
        self.model, self.preprocess, self.device = model.to(device), processor, device

    def _embed(self, dataset, batch_size, features):
        from torch.utils.data import DataLoader
        rows = []
        with torch.no_grad():
            for batch in DataLoader(dataset, batch_size=batch_size):
                batch = {name: t.to(self.device) for name, t in batch.items()}        # kwargs by name, tensors on the device
                rows.extend(features(**batch).detach().cpu().numpy())                 # one synchronising D2H per batch
        return np.stack(rows)

    def encode_images(self, images, batch_size):            # plip.py:31-53
        from datasets import Dataset
        ds = Dataset.from_dict({"image": images})
        ds.set_format("torch")
        ds.set_transform(lambda el: self.preprocess(images=el["image"], return_tensors="pt"))
        return self._embed(ds, batch_size, self.model.get_image_features)

    def encode_text(self, text, batch_size):                # plip.py:55-71
        from datasets import Dataset
        ds = Dataset.from_dict({"text": text}).map(
            lambda el: self.preprocess(text=el["text"], return_tensors="pt", max_length=77, padding="max_length", truncation=True),
            batched=True, remove_columns=["text"])
        ds.set_format("torch")
        return self._embed(ds, batch_size, self.model.get_text_features)

    @staticmethod
    def cosine_similarity(keys, space):                     # plip.py:73-76: only the key side is normalised
        return (keys / np.linalg.norm(keys, ord=2, axis=-1, keepdims=True)) @ space.T

    def zero_shot_classification(self, images, labels):     # plip.py:89-103
        sim = self.cosine_similarity(self.encode_images(images, 8), self.encode_text(labels, 8))
        return [labels[i] for i in np.argmax(sim, axis=-1)]
