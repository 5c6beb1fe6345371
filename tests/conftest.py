import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container (gpu-marked tests run on the MI355X box)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))
    return load


@pytest.fixture(scope="session")
def engines():
    """Session cache of PlipModel instances keyed by (case name, dtype)."""
    cache = {}

    def get(case, dtype, max_batch=32):
        from oracle.make_golden import case_inputs
        from plip_amd.model import PlipModel
        key = (case, dtype, max_batch)
        if key not in cache:
            cfg, sd, px, ids, mask = case_inputs(case)
            cache[key] = (PlipModel(cfg, sd, dtype=dtype, max_batch=max_batch), cfg, sd, px, ids, mask)
        return cache[key]

    yield get
    for m, *_ in cache.values():
        m.engine.close()
