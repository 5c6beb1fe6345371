"""On-disk embedding cache, file-compatible with reproducibility/utils/cacher.py.

Two naming schemes exist in the reference and both are kept so that caches written by either
implementation are hits for the other:

* hashed   (cacher.py:6-44):  ``$PC_CACHE_FOLDER/sha256(name + path).hexdigest()``
* raw      (cacher.py:51-74): ``$PC_CACHE_FOLDER/<dataset>/<model>/<backbone>`` where ``name`` is
  ``<model>img<dataset>[.csv...]`` and, for model ``plip``, ``<backbone>`` is the checkpoint's basename.

Files hold ``np.save`` bytes (a .npy payload without the extension) of the ``[N, 512]`` row-normalised matrix.
"""
from __future__ import annotations

import hashlib
import os
from typing import Optional

import numpy as np


def _folder() -> str:
    try:
        return os.environ["PC_CACHE_FOLDER"]
    except KeyError:
        raise KeyError("PC_CACHE_FOLDER is not set (the reference reads it from the environment / .env)") from None


def get_cache_name(name: str, path: str) -> str:
    digest = hashlib.sha256((name + path).encode("utf-8")).hexdigest()
    return os.path.join(_folder(), digest)


def _load(save_path: str) -> Optional[np.ndarray]:
    return np.load(save_path) if os.path.exists(save_path) else None


def _store(npa, save_path: str) -> None:
    tmp = f"{save_path}.tmp.{os.getpid()}"
    with open(tmp, "wb") as f:      # write-then-rename: a killed run never leaves a half-written hit behind
        np.save(f, np.asarray(npa))
    os.replace(tmp, save_path)


def cache_hit_or_miss(name: str, path: str):
    return _load(get_cache_name(name, path))


def cache_numpy_object(npa, name: str, path: str) -> None:
    os.makedirs(_folder(), exist_ok=True)
    _store(npa, get_cache_name(name, path))


def get_savepath(name: str, path: str) -> str:
    modelname, dataset_name = name.split("img")       # same (strict) unpacking as the reference
    dataset_name = dataset_name.split(".csv")[0]
    sub = os.path.join(_folder(), dataset_name, modelname)
    os.makedirs(sub, exist_ok=True)
    if modelname == "plip":
        path = os.path.basename(path)
    return os.path.join(sub, path)


def cache_hit_or_miss_raw_filename(name: str, path: str):
    hit = _load(get_savepath(name, path))
    print("[CACHE] Found existed embedding." if hit is not None
          else "[CACHE] No existed embedding found. Need to generate embedding first.")
    return hit


def cache_numpy_object_raw_filename(npa, name: str, path: str) -> None:
    save_path = get_savepath(name, path)
    print(f"[CACHE] Saving embedding. Name: {name}, Path: {path}, Save path: {save_path}")
    _store(npa, save_path)
