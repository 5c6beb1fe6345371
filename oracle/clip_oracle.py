"""CPU restatement (numpy) of the arithmetic behind PLIP.encode_images /
PLIP.encode_text / CLIPModel.forward -- TEST INFRASTRUCTURE, not product code.

Where the algorithm lives
-------------------------
/root/reference/plip.py:50,68 delegate every FLOP to HuggingFace
``transformers`` ``CLIPModel`` -- a third-party dependency that is NOT vendored
in the reference tree and is NOT pinned there (requirements.txt is empty,
setup.py:3-4,33).  The version restated here is transformers **5.15.0**
(the one installed in this image); ``$HF`` below =
``transformers/models/clip/modeling_clip.py`` of that version.  The
reproducibility/ scripts call the OpenAI ``clip`` package for the same network
(reproducibility/embedders/plip.py:48,66), un-pinned and not installed.

Pinning
-------
The reference ships no tests, golden vectors or known-answer values for this
path -- at the level of the reference repository parity is
therefore "unpinned".  This oracle is instead pinned against the reference's
actual arithmetic run in this container: ``oracle/make_golden.py`` loads the
same synthetic weights into HF ``CLIPModel`` (CPU, fp32, eager and sdpa
attention) and stores its outputs under tests/golden/; tests/test_oracle.py
checks every function here against those fixtures (and against a live HF model
when ``transformers`` is importable).

All functions take/return numpy arrays; ``dtype`` selects float32 (the
reference precision) or float64 (error budgeting).
"""
from __future__ import annotations

import numpy as np


def _f(sd, key, dtype):
    return np.asarray(sd[key], dtype=dtype)


def layer_norm(x, weight, bias, eps):
    """``nn.LayerNorm`` over the last axis, biased variance ($HF:358,360,559,605,608)."""
    mu = x.mean(axis=-1, keepdims=True)
    xc = x - mu
    var = (xc * xc).mean(axis=-1, keepdims=True)
    return xc / np.sqrt(var + x.dtype.type(eps)) * weight + bias


def quick_gelu(x):
    """``x * sigmoid(1.702 x)`` -- hidden_act="quick_gelu"
    (transformers/activations.py:117-123, configuration_clip.py:54,105)."""
    return x / (1.0 + np.exp(x.dtype.type(-1.702) * x))


def linear(x, w, b=None):
    """``nn.Linear``: x @ w.T + b, w stored [out, in]."""
    y = x @ w.T
    return y if b is None else y + b


def attention(x, sd, prefix, heads, causal, key_mask, dtype):
    """CLIPAttention.forward ($HF:298-335) with eager_attention_forward ($HF:259-277).

    x [B,S,D]; ``causal`` adds the text tower's causal mask; ``key_mask`` is the
    tokenizer's attention_mask [B,S] (1 = keep) combined with it the way
    ``create_causal_mask`` does ($HF:543-548).  Softmax in the working dtype
    (HF computes it in float32, $HF:271).
    """
    B, S, D = x.shape
    dh = D // heads
    scale = dtype(dh ** -0.5)
    q = linear(x, _f(sd, f"{prefix}.q_proj.weight", dtype), _f(sd, f"{prefix}.q_proj.bias", dtype))
    k = linear(x, _f(sd, f"{prefix}.k_proj.weight", dtype), _f(sd, f"{prefix}.k_proj.bias", dtype))
    v = linear(x, _f(sd, f"{prefix}.v_proj.weight", dtype), _f(sd, f"{prefix}.v_proj.bias", dtype))
    q = q.reshape(B, S, heads, dh).transpose(0, 2, 1, 3)
    k = k.reshape(B, S, heads, dh).transpose(0, 2, 1, 3)
    v = v.reshape(B, S, heads, dh).transpose(0, 2, 1, 3)
    scores = (q @ k.transpose(0, 1, 3, 2)) * scale               # [B,H,S,S]
    neg = np.finfo(dtype).min
    if causal:
        keep = np.tril(np.ones((S, S), dtype=bool))
        scores = np.where(keep[None, None], scores, neg)
    if key_mask is not None:
        scores = np.where(np.asarray(key_mask, dtype=bool)[:, None, None, :], scores, neg)
    scores = scores - scores.max(axis=-1, keepdims=True)
    p = np.exp(scores)
    p = p / p.sum(axis=-1, keepdims=True)
    o = (p @ v).transpose(0, 2, 1, 3).reshape(B, S, D)
    return linear(o, _f(sd, f"{prefix}.out_proj.weight", dtype), _f(sd, f"{prefix}.out_proj.bias", dtype))


def encoder_layer(x, sd, p, heads, causal, key_mask, eps, dtype):
    """CLIPEncoderLayer.forward, pre-LN residual block ($HF:362-383) with
    CLIPMLP fc1 -> QuickGELU -> fc2 ($HF:346-350)."""
    h = layer_norm(x, _f(sd, f"{p}.layer_norm1.weight", dtype), _f(sd, f"{p}.layer_norm1.bias", dtype), eps)
    x = x + attention(h, sd, f"{p}.self_attn", heads, causal, key_mask, dtype)
    h = layer_norm(x, _f(sd, f"{p}.layer_norm2.weight", dtype), _f(sd, f"{p}.layer_norm2.bias", dtype), eps)
    h = quick_gelu(linear(h, _f(sd, f"{p}.mlp.fc1.weight", dtype), _f(sd, f"{p}.mlp.fc1.bias", dtype)))
    h = linear(h, _f(sd, f"{p}.mlp.fc2.weight", dtype), _f(sd, f"{p}.mlp.fc2.bias", dtype))
    return x + h


def unfold_patches(pixels, patch):
    """Conv2d(kernel=stride=patch) as a re-index: [B,3,H,W] -> [B, gh*gw, 3*patch*patch],
    row (i,j) row-major over the grid, column (c,u,v) -- the order of
    ``conv.weight.reshape(out, -1)`` ($HF:148-154,209-210)."""
    B, C, H, W = pixels.shape
    gh, gw = H // patch, W // patch
    x = pixels.reshape(B, C, gh, patch, gw, patch).transpose(0, 2, 4, 1, 3, 5)
    return x.reshape(B, gh * gw, C * patch * patch)


def vision_tower(pixels, sd, cfg, dtype=np.float32, return_hidden=False):
    """CLIPModel.get_image_features ($HF:719-753) = CLIPVisionModel.forward
    ($HF:613-656) + visual_projection ($HF:674,751).  Returns un-normalised
    [B,P] embeddings (what plip.py:50 hands back)."""
    pixels = np.asarray(pixels, dtype=dtype)
    B = pixels.shape[0]
    Dv = cfg.v_width
    w = _f(sd, "vision_model.embeddings.patch_embedding.weight", dtype).reshape(Dv, -1)
    patches = unfold_patches(pixels, cfg.patch_size) @ w.T                      # [B,Np,Dv], bias=False
    cls = np.broadcast_to(_f(sd, "vision_model.embeddings.class_embedding", dtype), (B, 1, Dv))
    x = np.concatenate([cls, patches], axis=1)
    x = x + _f(sd, "vision_model.embeddings.position_embedding.weight", dtype)[None]   # $HF:212-217
    x = layer_norm(x, _f(sd, "vision_model.pre_layrnorm.weight", dtype),
                   _f(sd, "vision_model.pre_layrnorm.bias", dtype), cfg.layer_norm_eps)  # $HF:642
    hidden = [x]
    for i in range(cfg.v_layers):
        x = encoder_layer(x, sd, f"vision_model.encoder.layers.{i}", cfg.v_heads, False, None,
                          cfg.layer_norm_eps, dtype)
        hidden.append(x)
    pooled = layer_norm(x[:, 0, :], _f(sd, "vision_model.post_layernorm.weight", dtype),
                        _f(sd, "vision_model.post_layernorm.bias", dtype), cfg.layer_norm_eps)  # $HF:650-651
    emb = pooled @ _f(sd, "visual_projection.weight", dtype).T
    return (emb, hidden) if return_hidden else emb


def eos_positions(ids, eos_token_id):
    """Pooled-row rule of CLIPTextModel.forward ($HF:561-581): ``eos_token_id == 2``
    (legacy configs, and the OpenAI ``text.argmax(dim=-1)`` rule) -> first
    arg-max of the ids; otherwise the first position equal to ``eos_token_id``
    (0 when absent, as ``(ids == eos).int().argmax()`` gives)."""
    ids = np.asarray(ids)
    if eos_token_id == 2 or eos_token_id < 0:
        return ids.argmax(axis=-1)
    return (ids == eos_token_id).astype(np.int32).argmax(axis=-1)


def text_tower(ids, sd, cfg, attention_mask=None, dtype=np.float32, return_hidden=False):
    """CLIPModel.get_text_features ($HF:683-715) = CLIPTextModel.forward
    ($HF:513-586) + text_projection ($HF:675,713)."""
    ids = np.asarray(ids)
    B, S = ids.shape
    x = _f(sd, "text_model.embeddings.token_embedding.weight", dtype)[ids]       # $HF:251
    x = x + _f(sd, "text_model.embeddings.position_embedding.weight", dtype)[None, :S]  # $HF:253-254
    hidden = [x]
    for i in range(cfg.t_layers):
        x = encoder_layer(x, sd, f"text_model.encoder.layers.{i}", cfg.t_heads, True, attention_mask,
                          cfg.layer_norm_eps, dtype)
        hidden.append(x)
    x = layer_norm(x, _f(sd, "text_model.final_layer_norm.weight", dtype),
                   _f(sd, "text_model.final_layer_norm.bias", dtype), cfg.layer_norm_eps)  # $HF:559
    pooled = x[np.arange(B), eos_positions(ids, cfg.eos_token_id)]
    emb = pooled @ _f(sd, "text_projection.weight", dtype).T
    return (emb, hidden) if return_hidden else emb


def l2_normalize(x):
    """``x / sqrt(sum(x^2))`` with no epsilon ($HF:57-65,810-811; plip.py:75)."""
    return x / np.sqrt((x * x).sum(axis=-1, keepdims=True))


def clip_forward(pixels, ids, sd, cfg, attention_mask=None, dtype=np.float32):
    """CLIPModel.forward ($HF:757-831): both towers, L2 normalise both sides,
    ``logits_per_text = text @ image.T * exp(logit_scale)``, ``logits_per_image`` its transpose."""
    img_raw = vision_tower(pixels, sd, cfg, dtype)
    txt_raw = text_tower(ids, sd, cfg, attention_mask, dtype)
    img, txt = l2_normalize(img_raw), l2_normalize(txt_raw)
    scale = np.exp(dtype(sd["logit_scale"]))
    lpt = (txt @ img.T) * scale
    return {"image_features": img_raw, "text_features": txt_raw, "image_embeds": img,
            "text_embeds": txt, "logits_per_text": lpt, "logits_per_image": lpt.T.copy()}


def plip_cosine_similarity(key_vectors, space_vectors, normalize=True):
    """PLIP._cosine_similarity (plip.py:73-76): only the KEY side is normalised."""
    if normalize:
        key_vectors = key_vectors / np.linalg.norm(key_vectors, ord=2, axis=-1, keepdims=True)
    return key_vectors @ space_vectors.T
