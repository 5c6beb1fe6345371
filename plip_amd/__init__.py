"""plip_amd -- MI355X-native engine for the PLIP (CLIP ViT-B/32) embedding hot path.

The package is a thin host-side mirror of the reference's interfaces
(``PLIP``, HF ``CLIPModel``-style and OpenAI-clip-style model objects) over the
C-ABI library ``csrc/libplipmi.so`` (hand-written HIP kernels for gfx950).
Importing it does not need a GPU; creating an engine does, and fails loudly otherwise.
"""
from .config import PRESETS, PlipConfig, get_config  # noqa: F401

__version__ = "0.1.0"
__all__ = ["PlipConfig", "get_config", "PRESETS", "PLIP", "PlipModel", "Engine"]


def __getattr__(name):  # lazy: torch is only imported when the engine classes are used
    if name == "PLIP":
        from .plip import PLIP
        return PLIP
    if name in ("PlipModel", "PlipOutput"):
        from . import model
        return getattr(model, name)
    if name == "Engine":
        from .engine import Engine
        return Engine
    raise AttributeError(name)
