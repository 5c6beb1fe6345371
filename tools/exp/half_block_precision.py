"""Which HALF of a text block carries the bf16 operand-rounding error -- the attention half (q/k/v, attention, out-proj) or the MLP
half (fc1, fc2)?  Numpy precision model with the operand type chosen per half block in the leading N blocks, 48 captions of the
bs=256 fixture against HF's text_embeds.  CPU only.   python tools/exp/half_block_precision.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import clip_oracle as O  # noqa: E402
from oracle import precision_model as P  # noqa: E402
from plip_amd import weights as W  # noqa: E402
from plip_amd.config import get_config  # noqa: E402


class HalfRounding:
    """per layer call order of precision_model._layers: w xa w xa w xa (q k v) | qkv p att w (out) | w xa (fc1) mlp w (fc2)"""
    sites = P.ALL_SITES

    def __init__(self, attn, mlp):
        self.fa, self.fm = {"bf16": P.bf16, "f16": P.f16}[attn], {"bf16": P.bf16, "f16": P.f16}[mlp]
        self.n = 0

    def __call__(self, site, x):
        i = self.n
        self.n += 1
        return (self.fa if i < 10 else self.fm)(x)   # calls 0..9 belong to the attention half, 10..13 to the MLP half


def text_tower(ids, sd, cfg, mask, halves):
    B, S = ids.shape
    x = P._f(sd, "text_model.embeddings.token_embedding.weight")[ids] + P._f(sd, "text_model.embeddings.position_embedding.weight")[None, :S]
    for i in range(cfg.t_layers):
        rnd = HalfRounding(*halves[i])
        sub = {k.replace(f"text_model.encoder.layers.{i}.", "text_model.encoder.layers.0."): v for k, v in sd.items()
               if k.startswith(f"text_model.encoder.layers.{i}.")}
        x = P._layers(x, sub, "text_model", 1, cfg.t_heads, True, mask, cfg.layer_norm_eps, "folded", [], rnd)
        assert rnd.n == 14, rnd.n
    x = O.layer_norm(x, P._f(sd, "text_model.final_layer_norm.weight"), P._f(sd, "text_model.final_layer_norm.bias"), cfg.layer_norm_eps)
    pooled = x[np.arange(B), O.eos_positions(ids, cfg.eos_token_id)]
    return O.l2_normalize(pooled @ P._f(sd, "text_projection.weight").T)


def main():
    cfg = get_config("ViT-B/32")
    sd = W.synthetic_state_dict(cfg, 0)
    g = np.load(os.path.join(ROOT, "tests", "golden", "vitb32_b256.npz"))
    n = int(os.environ.get("NCAP", "48"))
    ids, mask, want = g["ids"][:n], g["attention_mask"][:n] if "attention_mask" in g else None, g["text_embeds"][:n]
    L = cfg.t_layers
    b, h = ("bf16", "bf16"), ("f16", "f16")
    arms = [("all bf16", [b] * L)]
    for k in (4, 8, 12):
        arms.append((f"first {k}: both halves f16", [h] * k + [b] * (L - k)))
        arms.append((f"first {k}: attention half f16", [("f16", "bf16")] * k + [b] * (L - k)))
        arms.append((f"first {k}: MLP half f16", [("bf16", "f16")] * k + [b] * (L - k)))
    for name, hv in arms:
        t0 = time.time()
        got = text_tower(ids, sd, cfg, mask, hv)
        e = np.abs(got - want)
        print(f"{name:32s} text_embeds max err {e.max():.3e}  rms {np.sqrt((e ** 2).mean()):.3e}   ({time.time() - t0:.0f} s)", flush=True)


if __name__ == "__main__":
    main()
