"""MI355X stand-ins for the pieces of the reference's ``reproducibility/`` tree that sit on the
embedding hot path:

=====================================================  ==========================================
reference                                              here
=====================================================  ==========================================
reproducibility/embedders/plip.py   ``CLIPEmbedder``   :class:`embedders.CLIPEmbedder`
reproducibility/embedders/factory.py                   :class:`embedders.EmbedderFactory`
reproducibility/utils/cacher.py                        :mod:`cacher` (same file names and .npy bytes)
reproducibility/evaluation/zero_shot/zero_shot.py      :class:`evaluation.ZeroShotClassifier`
reproducibility/evaluation/retrieval/retrieval.py      :class:`evaluation.ImageRetrieval`
reproducibility/metrics.py                             :mod:`metrics`
=====================================================  ==========================================

The towers, the normalisation, the similarity product, the arg-max and the top-50 all run in
libplipmi.so on the GPU; only file I/O, label bookkeeping and the metric arithmetic stay on the host.
"""
from .cacher import (cache_hit_or_miss, cache_hit_or_miss_raw_filename, cache_numpy_object,  # noqa: F401
                     cache_numpy_object_raw_filename, get_cache_name, get_savepath)
from .embedders import CLIPEmbedder, EmbedderFactory  # noqa: F401
from .evaluation import ImageRetrieval, ZeroShotClassifier  # noqa: F401
from .metrics import eval_metrics, retrieval_metrics  # noqa: F401
