"""Checkpoint ingestion and synthetic weights.

The canonical in-memory format is a ``dict[str, np.ndarray | torch.Tensor]``
with HuggingFace ``CLIPModel`` key names (separate q/k/v projections, the
``pre_layrnorm`` spelling included) because that is what the reference loads
at plip.py:26.  OpenAI-clip state dicts -- what
reproducibility/embedders/factory.py:21-25 loads (packed ``attn.in_proj_weight``,
``visual.proj`` stored [width, proj]) -- are converted to it.
"""
from __future__ import annotations

import math
import os
from typing import Dict, Mapping

import numpy as np

from .config import PlipConfig, from_hf_config, get_config

StateDict = Dict[str, np.ndarray]


# --------------------------------------------------------------------------
# key inventory
# --------------------------------------------------------------------------
def _layer_keys(prefix: str, i: int):
    p = f"{prefix}.encoder.layers.{i}"
    for proj in ("q_proj", "k_proj", "v_proj", "out_proj"):
        yield f"{p}.self_attn.{proj}.weight"
        yield f"{p}.self_attn.{proj}.bias"
    for ln in ("layer_norm1", "layer_norm2"):
        yield f"{p}.{ln}.weight"
        yield f"{p}.{ln}.bias"
    for fc in ("fc1", "fc2"):
        yield f"{p}.mlp.{fc}.weight"
        yield f"{p}.mlp.{fc}.bias"


def expected_shapes(cfg: PlipConfig) -> Dict[str, tuple]:
    """Every tensor of the HF-format state dict and its shape."""
    s: Dict[str, tuple] = {}
    Dv, Dt, P = cfg.v_width, cfg.t_width, cfg.projection_dim
    s["logit_scale"] = ()
    s["vision_model.embeddings.class_embedding"] = (Dv,)
    s["vision_model.embeddings.patch_embedding.weight"] = (Dv, 3, cfg.patch_size, cfg.patch_size)
    s["vision_model.embeddings.position_embedding.weight"] = (cfg.v_tokens, Dv)
    for ln in ("pre_layrnorm", "post_layernorm"):
        s[f"vision_model.{ln}.weight"] = (Dv,)
        s[f"vision_model.{ln}.bias"] = (Dv,)
    s["text_model.embeddings.token_embedding.weight"] = (cfg.vocab_size, Dt)
    s["text_model.embeddings.position_embedding.weight"] = (cfg.context_length, Dt)
    s["text_model.final_layer_norm.weight"] = (Dt,)
    s["text_model.final_layer_norm.bias"] = (Dt,)
    for prefix, D, F, L in (("vision_model", Dv, cfg.v_mlp, cfg.v_layers),
                            ("text_model", Dt, cfg.t_mlp, cfg.t_layers)):
        for i in range(L):
            for k in _layer_keys(prefix, i):
                if ".mlp.fc1.weight" in k:
                    s[k] = (F, D)
                elif ".mlp.fc1.bias" in k:
                    s[k] = (F,)
                elif ".mlp.fc2.weight" in k:
                    s[k] = (D, F)
                elif k.endswith("_proj.weight"):
                    s[k] = (D, D)
                else:
                    s[k] = (D,)
    s["visual_projection.weight"] = (P, Dv)
    s["text_projection.weight"] = (P, Dt)
    return s


# --------------------------------------------------------------------------
# synthetic weights (tests, golden fixtures, bench)
# --------------------------------------------------------------------------
def synthetic_state_dict(cfg: PlipConfig, seed: int = 0, logit_scale: float | None = None) -> StateDict:
    """Deterministic random weights, identical on every machine.

    ``numpy.random.RandomState`` (frozen MT19937 stream) is used instead of
    ``torch.manual_seed`` so the fixtures under tests/golden/ stay valid
    across torch builds.  Scales follow HF ``_init_weights``
    (modeling_clip.py:403-452) but biases and LayerNorm affine parameters are
    random too -- HF initialises them to 0/1, which would hide bias bugs.
    Keys are drawn in sorted order; each tensor is
    ``standard_normal(shape) * std (+ mean)`` in float32.
    """
    rs = np.random.RandomState(seed)
    shapes = expected_shapes(cfg)
    out: StateDict = {}
    for k in sorted(shapes):
        shp = shapes[k]
        if k == "logit_scale":
            out[k] = np.float32(cfg.logit_scale_init if logit_scale is None else logit_scale)
            continue
        tower = "vision" if k.startswith("vision") or k.startswith("visual") else "text"
        D = cfg.v_width if tower == "vision" else cfg.t_width
        L = cfg.v_layers if tower == "vision" else cfg.t_layers
        mean = 0.0
        if "norm" in k and k.endswith(".weight"):
            std, mean = 0.1, 1.0
        elif "norm" in k and k.endswith(".bias"):
            std = 0.1
        elif k.endswith(".bias"):
            std = 0.02
        elif "class_embedding" in k:
            std = D ** -0.5
        elif "embedding" in k:
            std = 0.02
        elif "q_proj" in k or "k_proj" in k or "v_proj" in k or "fc2" in k:
            std = D ** -0.5 * (2 * L) ** -0.5
        elif "out_proj" in k or "projection" in k:
            std = D ** -0.5
        elif "fc1" in k:
            std = (2 * D) ** -0.5
        else:  # pragma: no cover
            raise AssertionError(k)
        w = rs.standard_normal(shp).astype(np.float32)
        w *= np.float32(std)
        if mean:
            w += np.float32(mean)
        out[k] = w
    return out


def heavy_tailed_state_dict(cfg: PlipConfig, seed: int = 0, logit_scale: float | None = None,
                            n_outliers: int = 4) -> StateDict:
    """``synthetic_state_dict`` reshaped to look like a TRAINED CLIP/PLIP checkpoint where it hurts low precision:

    * ``n_outliers`` residual channels per tower carry values 30-100x the typical channel ("massive activations"):
      constant offsets injected by the embeddings / ``pre_layrnorm.bias`` plus token-dependent ones from up-scaled
      ``out_proj`` / ``fc2`` rows of those channels in the middle layers;
    * LayerNorm gains of the blocks are log-uniform over two decades (0.1 .. 10) instead of 1 +- 0.1, and the gains
      of the outlier channels are small (as trained models learn them), LayerNorm biases are 10x larger;
    * ``logit_scale`` defaults to ln 100 (the trained value CLIP clamps to).

    Deterministic (numpy ``RandomState``), generated THROUGH the same key order as ``synthetic_state_dict`` so the
    two share everything else.  Used by the outlier-robustness parity case (tests/golden/vitb32_b8_heavy.npz)."""
    sd = synthetic_state_dict(cfg, seed, math.log(100.0) if logit_scale is None else logit_scale)
    rs = np.random.RandomState(seed + 7919)
    for tower, D, L in (("vision_model", cfg.v_width, cfg.v_layers), ("text_model", cfg.t_width, cfg.t_layers)):
        ch = rs.choice(D, size=n_outliers, replace=False)
        amp = rs.uniform(30.0, 100.0, size=n_outliers).astype(np.float32) * rs.choice([-1.0, 1.0], size=n_outliers).astype(np.float32)
        if tower == "vision_model":     # the stream enters the blocks right after pre_layrnorm: unit-scale channels
            sd["vision_model.pre_layrnorm.bias"][ch] += amp
            unit = 1.0
        else:                            # token + position embeddings have std 0.02 each
            unit = 0.03
            sd["text_model.embeddings.position_embedding.weight"][:, ch] += amp * np.float32(unit)
        for i in range(L):
            p = f"{tower}.encoder.layers.{i}"
            for ln in ("layer_norm1", "layer_norm2"):
                g = np.exp(rs.uniform(np.log(0.1), np.log(10.0), size=D)).astype(np.float32)
                g[ch] = rs.uniform(0.02, 0.1, size=n_outliers).astype(np.float32)
                sd[f"{p}.{ln}.weight"] = g
                sd[f"{p}.{ln}.bias"] = sd[f"{p}.{ln}.bias"] * np.float32(10.0)
            if L // 3 <= i < 2 * L // 3 + 1:          # token-dependent outliers written by the middle blocks
                sd[f"{p}.self_attn.out_proj.weight"][ch, :] *= np.float32(30.0)
                sd[f"{p}.mlp.fc2.weight"][ch, :] *= np.float32(30.0)
            # keep the pre-softmax scores O(1) under the wide gains (a trained model's q/k weights are matched to them)
            for proj in ("q_proj", "k_proj", "v_proj"):
                sd[f"{p}.self_attn.{proj}.weight"] *= np.float32(0.3)
            sd[f"{p}.mlp.fc1.weight"] *= np.float32(0.3)
    return sd


def synthetic_pixels(cfg: PlipConfig, batch: int, seed: int = 1) -> np.ndarray:
    """fp32 [B,3,H,W] ~ N(0,1): the range of CLIP-normalised pixels."""
    rs = np.random.RandomState(seed)
    return rs.standard_normal((batch, 3, cfg.image_size, cfg.image_size)).astype(np.float32)


def synthetic_tiles(cfg: PlipConfig, batch: int, seed: int = 1000) -> np.ndarray:
    """uint8 [B,H,W,3] tiles with STRUCTURE (BASELINE.json configs[3]'s "synthetic corpus"): a stain-like base colour per
    tile, a low-frequency field per channel (a random 7 x 7 grid, nearest-upsampled) and pixel noise.  Pure-noise tiles
    all land on one embedding direction, so a zero-shot arg-max over class prompts would put the whole corpus in one
    class; these spread over several.  Deterministic (numpy ``RandomState``)."""
    rs = np.random.RandomState(seed)
    n = cfg.image_size
    out = np.empty((batch, n, n, 3), dtype=np.uint8)
    cell = -(-n // 7)
    for b in range(batch):
        base = rs.uniform(40.0, 215.0, size=3)
        amp = rs.uniform(10.0, 90.0, size=3)
        grid = rs.uniform(-1.0, 1.0, size=(7, 7, 3))
        field = np.repeat(np.repeat(grid, cell, axis=0), cell, axis=1)[:n, :n, :]
        noise = rs.uniform(-12.0, 12.0, size=(n, n, 3))
        out[b] = np.clip(base[None, None, :] + amp[None, None, :] * field + noise, 0.0, 255.0).astype(np.uint8)
    return out


def synthetic_ids(cfg: PlipConfig, batch: int, seed: int = 2, pad: str = "eos"):
    """int64 [B,ctx] token ids shaped like tokenizer output and the matching mask.

    Row = BOS, random body, EOS at position L (L ~ U{2..ctx-2}), then padding:
    ``pad='eos'`` repeats EOS (HF ``CLIPTokenizer``, pad_token=<|endoftext|>),
    ``pad='zero'`` pads with 0 (OpenAI ``clip.tokenize``,
    reproducibility/embedders/plip.py:65).
    """
    rs = np.random.RandomState(seed)
    S, V = cfg.context_length, cfg.vocab_size
    bos, eos = cfg.bos_token_id, cfg.eos_token_id
    ids = rs.randint(1, min(bos, eos) - 1, size=(batch, S)).astype(np.int64)
    lens = rs.randint(2, S - 1, size=batch)
    mask = np.zeros((batch, S), dtype=np.int64)
    ids[:, 0] = bos
    for b in range(batch):
        L = int(lens[b])
        ids[b, L] = eos
        ids[b, L + 1:] = eos if pad == "eos" else 0
        mask[b, :L + 1] = 1
    return ids, mask


# --------------------------------------------------------------------------
# real checkpoints
# --------------------------------------------------------------------------
def _to_numpy(v) -> np.ndarray:
    if isinstance(v, np.ndarray) or np.isscalar(v):
        return np.asarray(v, dtype=np.float32)
    return v.detach().to("cpu").float().numpy()  # torch tensor


def is_openai_state_dict(sd: Mapping[str, object]) -> bool:
    return "visual.conv1.weight" in sd or "visual.proj" in sd


def config_from_openai_state_dict(sd: Mapping[str, object]) -> PlipConfig:
    """Recover the architecture the way OpenAI ``clip.model.build_model`` does."""
    conv = sd["visual.conv1.weight"]
    v_width, _, patch, _ = conv.shape
    v_tokens = sd["visual.positional_embedding"].shape[0]
    grid = int(round(math.sqrt(v_tokens - 1)))
    v_layers = len({k.split(".")[3] for k in sd if k.startswith("visual.transformer.resblocks.")})
    t_layers = len({k.split(".")[2] for k in sd if k.startswith("transformer.resblocks.")})
    t_width = sd["ln_final.weight"].shape[0]
    ctx = sd["positional_embedding"].shape[0]
    vocab = sd["token_embedding.weight"].shape[0]
    proj = sd["text_projection"].shape[1]
    v_mlp = sd["visual.transformer.resblocks.0.mlp.c_fc.weight"].shape[0]
    t_mlp = sd["transformer.resblocks.0.mlp.c_fc.weight"].shape[0]
    return PlipConfig(image_size=grid * patch, patch_size=patch, v_width=v_width, v_layers=v_layers,
                      v_heads=v_width // 64, v_mlp=v_mlp, vocab_size=vocab, context_length=ctx,
                      t_width=t_width, t_layers=t_layers, t_heads=t_width // 64, t_mlp=t_mlp,
                      projection_dim=proj, eos_token_id=2)  # OpenAI pools at argmax(ids)


def convert_openai_state_dict(sd: Mapping[str, object]) -> StateDict:
    """OpenAI-clip names -> HF names (the inverse of what ``vinid/plip`` on the hub went through).

    Layout differences: ``attn.in_proj_weight`` is the
    packed [3D, D] q|k|v matrix; ``visual.proj`` / ``text_projection`` are stored
    [width, proj] and used as ``x @ proj`` (HF stores the Linear weight [proj, width]);
    ``visual.class_embedding`` etc. carry over unchanged.
    """
    out: StateDict = {}
    g = lambda k: _to_numpy(sd[k])
    out["logit_scale"] = np.float32(g("logit_scale"))
    out["vision_model.embeddings.class_embedding"] = g("visual.class_embedding")
    out["vision_model.embeddings.patch_embedding.weight"] = g("visual.conv1.weight")
    out["vision_model.embeddings.position_embedding.weight"] = g("visual.positional_embedding")
    out["vision_model.pre_layrnorm.weight"] = g("visual.ln_pre.weight")
    out["vision_model.pre_layrnorm.bias"] = g("visual.ln_pre.bias")
    out["vision_model.post_layernorm.weight"] = g("visual.ln_post.weight")
    out["vision_model.post_layernorm.bias"] = g("visual.ln_post.bias")
    out["visual_projection.weight"] = np.ascontiguousarray(g("visual.proj").T)
    out["text_model.embeddings.token_embedding.weight"] = g("token_embedding.weight")
    out["text_model.embeddings.position_embedding.weight"] = g("positional_embedding")
    out["text_model.final_layer_norm.weight"] = g("ln_final.weight")
    out["text_model.final_layer_norm.bias"] = g("ln_final.bias")
    out["text_projection.weight"] = np.ascontiguousarray(g("text_projection").T)
    for src, dst in (("visual.transformer", "vision_model"), ("transformer", "text_model")):
        i = 0
        while f"{src}.resblocks.{i}.ln_1.weight" in sd:
            s, d = f"{src}.resblocks.{i}", f"{dst}.encoder.layers.{i}"
            w, b = g(f"{s}.attn.in_proj_weight"), g(f"{s}.attn.in_proj_bias")
            D = w.shape[1]
            for j, name in enumerate(("q_proj", "k_proj", "v_proj")):
                out[f"{d}.self_attn.{name}.weight"] = np.ascontiguousarray(w[j * D:(j + 1) * D])
                out[f"{d}.self_attn.{name}.bias"] = np.ascontiguousarray(b[j * D:(j + 1) * D])
            out[f"{d}.self_attn.out_proj.weight"] = g(f"{s}.attn.out_proj.weight")
            out[f"{d}.self_attn.out_proj.bias"] = g(f"{s}.attn.out_proj.bias")
            out[f"{d}.layer_norm1.weight"] = g(f"{s}.ln_1.weight")
            out[f"{d}.layer_norm1.bias"] = g(f"{s}.ln_1.bias")
            out[f"{d}.layer_norm2.weight"] = g(f"{s}.ln_2.weight")
            out[f"{d}.layer_norm2.bias"] = g(f"{s}.ln_2.bias")
            out[f"{d}.mlp.fc1.weight"] = g(f"{s}.mlp.c_fc.weight")
            out[f"{d}.mlp.fc1.bias"] = g(f"{s}.mlp.c_fc.bias")
            out[f"{d}.mlp.fc2.weight"] = g(f"{s}.mlp.c_proj.weight")
            out[f"{d}.mlp.fc2.bias"] = g(f"{s}.mlp.c_proj.bias")
            i += 1
    return out


def to_openai_state_dict(sd: Mapping[str, np.ndarray], cfg: PlipConfig) -> StateDict:
    """HF names -> OpenAI-clip names (used by the tests to round-trip the converter)."""
    out: StateDict = {}
    out["logit_scale"] = np.float32(sd["logit_scale"])
    out["visual.class_embedding"] = sd["vision_model.embeddings.class_embedding"]
    out["visual.conv1.weight"] = sd["vision_model.embeddings.patch_embedding.weight"]
    out["visual.positional_embedding"] = sd["vision_model.embeddings.position_embedding.weight"]
    out["visual.ln_pre.weight"] = sd["vision_model.pre_layrnorm.weight"]
    out["visual.ln_pre.bias"] = sd["vision_model.pre_layrnorm.bias"]
    out["visual.ln_post.weight"] = sd["vision_model.post_layernorm.weight"]
    out["visual.ln_post.bias"] = sd["vision_model.post_layernorm.bias"]
    out["visual.proj"] = np.ascontiguousarray(sd["visual_projection.weight"].T)
    out["token_embedding.weight"] = sd["text_model.embeddings.token_embedding.weight"]
    out["positional_embedding"] = sd["text_model.embeddings.position_embedding.weight"]
    out["ln_final.weight"] = sd["text_model.final_layer_norm.weight"]
    out["ln_final.bias"] = sd["text_model.final_layer_norm.bias"]
    out["text_projection"] = np.ascontiguousarray(sd["text_projection.weight"].T)
    for dst, src, L in (("visual.transformer", "vision_model", cfg.v_layers),
                        ("transformer", "text_model", cfg.t_layers)):
        for i in range(L):
            s, d = f"{src}.encoder.layers.{i}", f"{dst}.resblocks.{i}"
            out[f"{d}.attn.in_proj_weight"] = np.concatenate(
                [sd[f"{s}.self_attn.{n}.weight"] for n in ("q_proj", "k_proj", "v_proj")], 0)
            out[f"{d}.attn.in_proj_bias"] = np.concatenate(
                [sd[f"{s}.self_attn.{n}.bias"] for n in ("q_proj", "k_proj", "v_proj")], 0)
            out[f"{d}.attn.out_proj.weight"] = sd[f"{s}.self_attn.out_proj.weight"]
            out[f"{d}.attn.out_proj.bias"] = sd[f"{s}.self_attn.out_proj.bias"]
            out[f"{d}.ln_1.weight"] = sd[f"{s}.layer_norm1.weight"]
            out[f"{d}.ln_1.bias"] = sd[f"{s}.layer_norm1.bias"]
            out[f"{d}.ln_2.weight"] = sd[f"{s}.layer_norm2.weight"]
            out[f"{d}.ln_2.bias"] = sd[f"{s}.layer_norm2.bias"]
            out[f"{d}.mlp.c_fc.weight"] = sd[f"{s}.mlp.fc1.weight"]
            out[f"{d}.mlp.c_fc.bias"] = sd[f"{s}.mlp.fc1.bias"]
            out[f"{d}.mlp.c_proj.weight"] = sd[f"{s}.mlp.fc2.weight"]
            out[f"{d}.mlp.c_proj.bias"] = sd[f"{s}.mlp.fc2.bias"]
    return out


def normalize_state_dict(sd: Mapping[str, object], cfg: PlipConfig | None = None):
    """Return ``(hf_format_numpy_state_dict, cfg)`` for either checkpoint flavour."""
    if is_openai_state_dict(sd):
        if cfg is None:
            cfg = config_from_openai_state_dict(sd)
        out = convert_openai_state_dict(sd)
    else:
        out = {k: _to_numpy(v) for k, v in sd.items()
               if not k.endswith("position_ids")}  # non-persistent buffer in old HF checkpoints
        if cfg is None:
            cfg = get_config("ViT-B/32")
    out["logit_scale"] = np.float32(np.asarray(out["logit_scale"], dtype=np.float32).reshape(-1)[0])
    check_state_dict(out, cfg)
    return out, cfg


def check_state_dict(sd: Mapping[str, np.ndarray], cfg: PlipConfig) -> None:
    want = expected_shapes(cfg)
    missing = sorted(set(want) - set(sd))
    extra = sorted(set(sd) - set(want))
    if missing or extra:
        raise KeyError(f"state dict mismatch: missing={missing[:5]} ({len(missing)}), "
                       f"unexpected={extra[:5]} ({len(extra)})")
    for k, shp in want.items():
        if tuple(np.shape(sd[k])) != tuple(shp):
            raise ValueError(f"{k}: shape {tuple(np.shape(sd[k]))}, expected {tuple(shp)}")


def load_checkpoint(path: str, arch: str | None = None, trust_pickle: bool = False):
    """Load real PLIP weights from disk: an HF directory (``config.json`` +
    ``model.safetensors`` / ``pytorch_model.bin``, what plip.py:26 pulls from the
    hub) or an OpenAI-clip ``.pt`` state dict (factory.py:23-25).

    ``.pt`` / ``.bin`` files are read with ``weights_only=True`` (tensors only: a plain state dict needs nothing
    more).  Whole-model or TorchScript pickles execute code while loading; they are refused unless
    ``trust_pickle=True`` (or ``PLIPMI_TRUST_PICKLE=1``) says the file comes from a trusted source."""
    import json

    import torch

    cfg = get_config(arch) if arch else None
    if os.path.isdir(path):
        cj = os.path.join(path, "config.json")
        if cfg is None and os.path.exists(cj):
            with open(cj) as f:
                cfg = from_hf_config(json.load(f))
        st = os.path.join(path, "model.safetensors")
        if os.path.exists(st):
            from safetensors.numpy import load_file
            sd = load_file(st)
        else:
            sd = torch.load(os.path.join(path, "pytorch_model.bin"), map_location="cpu",
                            weights_only=True)
    elif path.endswith(".safetensors"):
        from safetensors.numpy import load_file
        sd = load_file(path)
    else:
        import pickle
        import zipfile
        try:
            sd = torch.load(path, map_location="cpu", weights_only=True)
        except (OSError, TypeError):
            raise                                     # missing / unreadable file, or a torch without weights_only: not a pickle question
        except (EOFError, zipfile.BadZipFile) as e:
            # a damaged file is not a trust question: do not steer the user towards arbitrary-code unpickling for it
            raise RuntimeError(f"{path} is truncated or corrupt ({type(e).__name__}: {e}); re-download or re-save it") from e
        except (pickle.UnpicklingError, RuntimeError, AttributeError, ImportError, KeyError) as e:
            # a truncated torch ZIP checkpoint surfaces as RuntimeError("PytorchStreamReader failed reading zip archive: failed finding
            # central directory"): damaged, not untrusted
            if isinstance(e, RuntimeError) and any(k in str(e) for k in ("PytorchStreamReader", "central directory", "failed reading zip")):
                raise RuntimeError(f"{path} is truncated or corrupt ({str(e)[:160]}); re-download or re-save it") from e
            # what weights_only=True raises on a code-carrying pickle ("Unsupported global / class", UnpicklingError) -- and what
            # legacy pickles naming modules that are gone raise on their way there: these get the trust guidance below
            if isinstance(e, RuntimeError) and not any(k in str(e) for k in ("Unsupported", "weights_only", "pickle", "GLOBAL", "global")):
                raise                                 # some other runtime failure of torch.load: not a pickle-trust question
            if not (trust_pickle or os.environ.get("PLIPMI_TRUST_PICKLE") == "1"):
                raise RuntimeError(
                    f"{path} is not a plain tensor state dict ({type(e).__name__}: {str(e)[:200]}). Loading it needs full "
                    "unpickling, which can run arbitrary code: pass trust_pickle=True / PLIPMI_TRUST_PICKLE=1 only for a "
                    "file you trust, or re-save it as model.state_dict().") from e
            sd = torch.load(path, map_location="cpu", weights_only=False)
        if hasattr(sd, "state_dict"):  # a whole pickled / jit OpenAI model
            sd = sd.state_dict()
        if isinstance(sd, dict) and "state_dict" in sd and "logit_scale" not in sd:
            sd = sd["state_dict"]
    return normalize_state_dict(sd, cfg)
