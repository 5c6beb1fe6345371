"""Numpy emulation of the 16-bit engines' PRECISION PLAN -- TEST INFRASTRUCTURE, not product code.

``clip_oracle`` restates the reference's arithmetic (HF ``CLIPModel``, fp32).  This module restates the same
network with bf16 roundings inserted where libplipmi.so's bf16 engine rounds (bf16 MFMA
operands, fp32 accumulation / residual stream / LayerNorm + softmax statistics / pooled head / logits), so the
error budget of a precision plan can be costed on the CPU -- against the HF golden vectors -- before GPU time is
spent on it, and so the tests can tell "the kernel is wrong" from "the plan is this inexact".

Two plans for the LayerNorm -> Linear pairs (modeling_clip.py:370-381: layer_norm1 -> q/k/v, layer_norm2 -> fc1):

* ``"round_ln"``  h = bf16(LN(x) * g + b);  y = h @ bf16(W)^T + bias                      (round 1 engine)
* ``"folded"``    y = rstd * (bf16(x) @ W'^T) + c2,   W' = bf16(W * g - rowmean_k(W * g)),  c2 = W @ b + bias (fp32)
                                                                                             (round 2 engine)
  -- LayerNorm's gain AND its mean subtraction are folded into the weights (a row of W' sums to zero, so
  ``x @ W'^T == (x - mean(x)) @ (W * g)^T``), its bias into the Linear's bias, and only the row's rstd enters in the
  GEMM epilogue: the normalised activations never exist in memory; algebraically identical to the reference's
  ``linear(layer_norm(x))``.

GEMM accumulation is emulated in float64 (the MFMA accumulates fp32: its error is far below one bf16 ulp of the
operands).

``dtype`` selects the operand type -- "bf16" (8 significand bits) or "f16" (IEEE half, 11 bits, the PLIPMI_F16 engine) --
and ``sites`` which of the six operand roundings are applied, so the error of a plan can be DECOMPOSED:
``xa`` (A operand of q/k/v and fc1), ``w`` (every Linear weight), ``qkv`` (the attention inputs), ``p`` (softmax
probabilities), ``att`` (attention output = A operand of out_proj), ``mlp`` (fc1 output = A operand of fc2).
What that decomposition says for the text tower of the bs=256 fixture (tests/test_oracle.py): bf16 WEIGHT rounding alone
puts text_embeds 8.9e-4 from HF (it is coherent across a caption's tokens, so attention does not average it out), all six
bf16 roundings 1.2e-3, all six in f16 1.4e-4.
"""
from __future__ import annotations

import numpy as np

from . import clip_oracle as O


def bf16(x):
    """Round-to-nearest-even to bfloat16, returned as float32 (v_cvt_pk_bf16_f32 semantics)."""
    a = np.ascontiguousarray(x, dtype=np.float32)
    u = a.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32).reshape(a.shape)


def f16(x):
    """Round-to-nearest-even to IEEE half (saturating like the engine's from_f32<f16_t>), returned as float32."""
    a = np.clip(np.ascontiguousarray(x, dtype=np.float32), -65504.0, 65504.0)
    return a.astype(np.float16).astype(np.float32)


ALL_SITES = frozenset(("xa", "w", "qkv", "p", "att", "mlp"))


class Rounding:
    """site -> rounding function for one (dtype, sites) choice"""
    def __init__(self, dtype="bf16", sites=ALL_SITES):
        self.fn = {"bf16": bf16, "f16": f16}[dtype]
        self.sites = frozenset(sites)
        assert self.sites <= ALL_SITES, self.sites - ALL_SITES

    def __call__(self, site, x):
        return self.fn(x) if site in self.sites else np.asarray(x, dtype=np.float32)


_DEFAULT = Rounding()


def _mm(a, w):  # a [.., K] @ w[N, K]^T with wide accumulation, fp32 result
    return (a.astype(np.float64) @ w.astype(np.float64).T).astype(np.float32)


def _f(sd, k):
    return np.asarray(sd[k], dtype=np.float32)


def _ln_stats(x, eps):
    mu = x.mean(axis=-1, keepdims=True, dtype=np.float64)
    var = ((x - mu) ** 2).mean(axis=-1, keepdims=True, dtype=np.float64)
    return mu.astype(np.float32), (1.0 / np.sqrt(var + eps)).astype(np.float32)


def ln_linear(x, g, b, W, bias, eps, plan, pre=1.0, rnd=_DEFAULT):
    """Linear(LayerNorm(x)) under ``plan``; ``pre`` scales the output rows (the 1/8 folded into W_q, b_q)."""
    mu, rstd = _ln_stats(x, eps)
    if plan == "round_ln":
        h = rnd("xa", (x - mu) * rstd * g + b)
        return _mm(h, rnd("w", W * np.float32(pre))) + bias * np.float32(pre)
    if plan == "folded":
        Wg = (W * g[None, :]).astype(np.float32)
        Wg = rnd("w", np.float32(pre) * (Wg - Wg.astype(np.float64).mean(axis=1, keepdims=True).astype(np.float32)))
        c2 = ((W.astype(np.float64) @ b.astype(np.float64)) * pre).astype(np.float32) + bias * np.float32(pre)
        return rstd * _mm(rnd("xa", x), Wg) + c2
    raise ValueError(plan)


def _attention(qkv, B, S, H, causal, key_mask, rnd=_DEFAULT):
    D = H * 64
    q, k, v = (qkv[..., i * D:(i + 1) * D].reshape(B, S, H, 64).transpose(0, 2, 1, 3) for i in range(3))
    s = (q.astype(np.float64) @ k.astype(np.float64).transpose(0, 1, 3, 2)).astype(np.float32)   # scale folded into q
    neg = np.float32(-1e30)
    if causal:
        s = np.where(np.tril(np.ones((S, S), bool))[None, None], s, neg)
    if key_mask is not None:
        s = np.where(np.asarray(key_mask, bool)[:, None, None, :], s, neg)
    p = np.exp(s - s.max(-1, keepdims=True))
    o = (rnd("p", p).astype(np.float64) @ v.astype(np.float64)).astype(np.float32) / p.sum(-1, keepdims=True)
    return rnd("att", o.transpose(0, 2, 1, 3).reshape(B, S, D))


def _layers(x, sd, prefix, L, H, causal, key_mask, eps, plan, hidden, rnd=_DEFAULT, lead=0):
    """``lead``: the first ``lead`` blocks round their operands to f16 whatever ``rnd`` says for the rest
    (plipmi_config.text_f16_layers: a bf16 text tower whose leading blocks run on IEEE-half operands)."""
    B, S, D = x.shape
    rnd_rest, rnd_lead = rnd, Rounding("f16", rnd.sites)
    for i in range(L):
        rnd = rnd_lead if i < lead else rnd_rest
        p = f"{prefix}.encoder.layers.{i}"
        g1, b1 = _f(sd, f"{p}.layer_norm1.weight"), _f(sd, f"{p}.layer_norm1.bias")
        parts = []
        for name, pre in (("q_proj", 0.125), ("k_proj", 1.0), ("v_proj", 1.0)):
            parts.append(ln_linear(x, g1, b1, _f(sd, f"{p}.self_attn.{name}.weight"), _f(sd, f"{p}.self_attn.{name}.bias"),
                                   eps, plan, pre, rnd))
        qkv = rnd("qkv", np.concatenate(parts, axis=-1))
        att = _attention(qkv, B, S, H, causal, key_mask, rnd)
        x = x + (_mm(att, rnd("w", _f(sd, f"{p}.self_attn.out_proj.weight"))) + _f(sd, f"{p}.self_attn.out_proj.bias"))
        g2, b2 = _f(sd, f"{p}.layer_norm2.weight"), _f(sd, f"{p}.layer_norm2.bias")
        m = rnd("mlp", O.quick_gelu(ln_linear(x, g2, b2, _f(sd, f"{p}.mlp.fc1.weight"), _f(sd, f"{p}.mlp.fc1.bias"), eps, plan, 1.0, rnd)))
        x = x + (_mm(m, rnd("w", _f(sd, f"{p}.mlp.fc2.weight"))) + _f(sd, f"{p}.mlp.fc2.bias"))
        hidden.append(x)
    return x


def vision_tower(pixels, sd, cfg, plan="folded", return_hidden=False, dtype="bf16", sites=ALL_SITES):
    rnd = Rounding(dtype, sites)
    pixels = np.asarray(pixels, np.float32)
    B, Dv = pixels.shape[0], cfg.v_width
    w = _f(sd, "vision_model.embeddings.patch_embedding.weight").reshape(Dv, -1)
    patches = _mm(rnd("xa", O.unfold_patches(pixels, cfg.patch_size)), rnd("w", w))
    cls = np.broadcast_to(_f(sd, "vision_model.embeddings.class_embedding"), (B, 1, Dv))
    x = np.concatenate([cls, patches], axis=1) + _f(sd, "vision_model.embeddings.position_embedding.weight")[None]
    x = O.layer_norm(x, _f(sd, "vision_model.pre_layrnorm.weight"), _f(sd, "vision_model.pre_layrnorm.bias"),
                     cfg.layer_norm_eps)
    hidden = [x]
    x = _layers(x, sd, "vision_model", cfg.v_layers, cfg.v_heads, False, None, cfg.layer_norm_eps, plan, hidden, rnd)
    pooled = O.layer_norm(x[:, 0, :], _f(sd, "vision_model.post_layernorm.weight"),
                          _f(sd, "vision_model.post_layernorm.bias"), cfg.layer_norm_eps)
    emb = pooled @ _f(sd, "visual_projection.weight").T
    return (emb, hidden) if return_hidden else emb


def text_tower(ids, sd, cfg, attention_mask=None, plan="folded", return_hidden=False, dtype="bf16", sites=ALL_SITES,
               lead_f16=0):
    rnd = Rounding(dtype, sites)
    ids = np.asarray(ids)
    B, S = ids.shape
    x = _f(sd, "text_model.embeddings.token_embedding.weight")[ids] + \
        _f(sd, "text_model.embeddings.position_embedding.weight")[None, :S]
    hidden = [x]
    x = _layers(x, sd, "text_model", cfg.t_layers, cfg.t_heads, True, attention_mask, cfg.layer_norm_eps, plan, hidden, rnd,
                lead_f16)
    x = O.layer_norm(x, _f(sd, "text_model.final_layer_norm.weight"), _f(sd, "text_model.final_layer_norm.bias"),
                     cfg.layer_norm_eps)
    pooled = x[np.arange(B), O.eos_positions(ids, cfg.eos_token_id)]
    emb = pooled @ _f(sd, "text_projection.weight").T
    return (emb, hidden) if return_hidden else emb


def clip_forward(pixels, ids, sd, cfg, attention_mask=None, plan="folded", dtype="bf16", sites=ALL_SITES):
    img_raw = vision_tower(pixels, sd, cfg, plan, dtype=dtype, sites=sites)
    txt_raw = text_tower(ids, sd, cfg, attention_mask, plan, dtype=dtype, sites=sites)
    img, txt = O.l2_normalize(img_raw), O.l2_normalize(txt_raw)
    lpt = (txt @ img.T) * np.exp(np.float32(sd["logit_scale"]))
    return {"image_features": img_raw, "text_features": txt_raw, "image_embeds": img, "text_embeds": txt,
            "logits_per_text": lpt, "logits_per_image": lpt.T.copy()}
