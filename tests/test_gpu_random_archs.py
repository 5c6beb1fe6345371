"""Randomised architectures against the numpy oracle, through the C ABI.

The fixtures pin a handful of named configurations; this sweeps the dimensions the engine is parameterised by -- widths,
depths, patch / image sizes (5 ... 145 vision tokens: every attention kernel), context lengths (odd ones, longer than 128),
projection widths (with and without the MFMA head), MLP ratios, batch sizes -- on seeded random models small enough for
the oracle to finish in a second, in all three compute dtypes and with the engine's A/B forms switched at random.
Reference arithmetic: oracle/clip_oracle.py (HF modeling_clip.py:138-831)."""
import numpy as np
import pytest
import torch

from oracle import clip_oracle as O
from plip_amd import weights as W
from plip_amd.config import PlipConfig

pytestmark = pytest.mark.gpu

N_CASES = 48


def _random_case(seed):
    rs = np.random.RandomState(7000 + seed)
    vw, tw = (int(rs.choice([128, 256, 384])) for _ in range(2))
    patch = int(rs.choice([4, 8, 14, 16, 32]))
    grid = int(rs.randint(2, 10)) if seed % 5 else 12            # every fifth case: 145 tokens (chunked attention)
    ctx = int(rs.randint(5, 100)) if seed % 7 else int(rs.randint(129, 160))
    vocab = int(rs.randint(300, 2000))
    cfg = PlipConfig(image_size=patch * grid, patch_size=patch, v_width=vw, v_layers=int(rs.randint(1, 4)), v_heads=vw // 64,
                     v_mlp=128 * int(rs.randint(1, 6)), vocab_size=vocab, context_length=ctx, t_width=tw,
                     t_layers=int(rs.randint(1, 4)), t_heads=tw // 64, t_mlp=128 * int(rs.randint(1, 6)),
                     projection_dim=int(rs.choice([32, 64, 80, 96, 200, 256])), eos_token_id=vocab - 1, bos_token_id=vocab - 2)
    batch = int(rs.randint(1, 10))
    opts = {}
    if rs.rand() < 0.3:
        opts["ln_fold"] = False
    if rs.rand() < 0.3:
        opts["pooled_last_block"] = False
    if rs.rand() < 0.3:
        opts["graph_batch"] = 0
    if rs.rand() < 0.3:
        opts["text_f16"] = True                                  # bf16 engine only
    if rs.rand() < 0.4 and "ln_fold" not in opts and "pooled_last_block" not in opts:
        opts["pack_captions"] = True                             # 16-bit engines with the pooled last block
    return cfg, batch, opts, int(rs.randint(0, 1 << 30))


# cosine-logit tolerances of small random models (64 ... 256-d unit vectors have larger components than ViT-B/32's 512-d
# ones: the TINY rows of tests/test_gpu_parity.py)
COS = {"f32": 2e-5, "bf16": 4e-3, "f16": 1e-3}


@pytest.mark.parametrize("seed", range(N_CASES))
def test_random_architecture_matches_the_oracle(seed):
    from plip_amd.model import PlipModel
    cfg, B, opts, s = _random_case(seed)
    cfg.validate()
    sd = W.synthetic_state_dict(cfg, s)
    px = W.synthetic_pixels(cfg, B, s + 1)
    ids, mask = W.synthetic_ids(cfg, B, s + 2, pad="eos" if seed % 2 else "zero")
    use_mask = mask if seed % 3 else None
    ref = O.clip_forward(px, ids, sd, cfg, use_mask)
    scale = np.exp(np.float64(sd["logit_scale"]))
    for dtype in ("f32", "bf16", "f16"):
        kw = dict(opts) if dtype != "f32" else {k: v for k, v in opts.items() if k == "graph_batch"}
        if dtype != "bf16":
            kw.pop("text_f16", None)
        model = PlipModel(cfg, sd, dtype=dtype, max_batch=B, **kw)
        try:
            for rep in range(3):      # call 1 eager, call 2 captures a hipGraph (small batches), call 3 replays it
                out = model(input_ids=torch.from_numpy(ids), pixel_values=torch.from_numpy(px),
                            attention_mask=None if use_mask is None else torch.from_numpy(mask))
                got = out.logits_per_image.cpu().numpy()
                assert got.shape == (B, B) and np.isfinite(got).all()
                err = np.abs(got - ref["logits_per_image"]).max() / scale
                assert err < COS[dtype], (seed, dtype, rep, cfg, B, opts, err)
                assert torch.equal(out.logits_per_image, out.logits_per_text.T.contiguous())
            # the un-normalised features PLIP.encode_* return, separately (text without the pixel path and vice versa)
            img = model.get_image_features(pixel_values=torch.from_numpy(px)).cpu().numpy()
            txt = model.get_text_features(input_ids=torch.from_numpy(ids)).cpu().numpy()
            for a, b in ((img, ref["image_features"]), (txt, ref["text_features"])):
                rel = np.abs(a - b).max() / max(np.abs(b).max(), 1e-6)
                assert rel < {"f32": 5e-5, "bf16": 4e-2, "f16": 6e-3}[dtype], (seed, dtype, rel)
        finally:
            model.engine.close()
