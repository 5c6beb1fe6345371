"""Which text-tower layers carry the bf16 engine's text_embeds error?  ('f16 only where the error is made')
Numpy precision model (oracle/precision_model.py) with the operand type chosen PER LAYER, 48 captions of the bs=256 fixture
against HF's text_embeds.  CPU only.   python tools/exp/layer_precision.py"""
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


def text_tower(ids, sd, cfg, mask, layer_dtypes):
    B, S = ids.shape
    x = P._f(sd, "text_model.embeddings.token_embedding.weight")[ids] + P._f(sd, "text_model.embeddings.position_embedding.weight")[None, :S]
    for i in range(cfg.t_layers):
        rnd = P.Rounding(layer_dtypes[i])
        sub = {k.replace(f"text_model.encoder.layers.{i}.", "text_model.encoder.layers.0."): v for k, v in sd.items()
               if k.startswith(f"text_model.encoder.layers.{i}.")}
        x = P._layers(x, sub, "text_model", 1, cfg.t_heads, True, mask, cfg.layer_norm_eps, "folded", [], rnd)
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
    arms = [("all bf16", ["bf16"] * L), ("all f16", ["f16"] * L)]
    for k in (2, 4, 6, 8):
        arms.append((f"f16 in the LAST {k}", ["bf16"] * (L - k) + ["f16"] * k))
        arms.append((f"f16 in the FIRST {k}", ["f16"] * k + ["bf16"] * (L - k)))
    for i in (0, 5, 11):
        arms.append((f"f16 in layer {i} only", ["f16" if j == i else "bf16" for j in range(L)]))
    for name, dts in arms:
        t0 = time.time()
        got = text_tower(ids, sd, cfg, mask, dts)
        e = np.abs(got - want)
        print(f"{name:24s} text_embeds max err {e.max():.3e}  rms {np.sqrt((e ** 2).mean()):.3e}   ({time.time() - t0:.0f} s)", flush=True)


if __name__ == "__main__":
    main()
