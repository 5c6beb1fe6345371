"""The CPU oracle (oracle/clip_oracle.py) pinned against the golden vectors produced by the
reference's arithmetic (HF CLIPModel, oracle/make_golden.py).  CPU only."""
import numpy as np
import pytest

from oracle import clip_oracle as O
from oracle import hf_reference as H
from oracle.make_golden import CASES, SLIM, case_inputs, fingerprint

FULL_CASES = [c for c in CASES if c not in SLIM]

FP32_FEAT_TOL = 2e-5     # |features| ~ 3-5; HF sdpa vs eager already differ by ~2e-6
FP32_COS_TOL = 2e-6


@pytest.mark.parametrize("name", list(CASES))
def test_fixture_inputs_regenerate(name, golden):
    """Seeds -> exactly the tensors the fixtures were generated from (RandomState is frozen)."""
    g = golden(name)
    cfg, sd, px, ids, mask = case_inputs(name)
    np.testing.assert_array_equal(g["ids"], ids)
    np.testing.assert_array_equal(g["mask"], mask)
    np.testing.assert_allclose(fingerprint(sd), g["weights_fingerprint"], rtol=0, atol=0)
    fp = np.array([px.astype(np.float64).sum(), (px.astype(np.float64) ** 2).sum()])
    np.testing.assert_allclose(fp, g["pixels_fingerprint"], rtol=0, atol=0)


def test_oracle_matches_golden_bs256_rows(golden):
    """BASELINE configs[2] size: the oracle on eight rows of the 256-pair batch against HF's own 256 x 256 logits
    (samples are independent, so a row subset is exact; the whole batch would cost the CPU suite a minute)."""
    g = golden("vitb32_b256")
    cfg, sd, px, ids, mask = case_inputs("vitb32_b256")
    rows = [0, 1, 37, 100, 128, 199, 254, 255]
    o = O.clip_forward(px[rows], ids[rows], sd, cfg, mask[rows])
    assert np.abs(o["image_embeds"] - g["image_embeds"][rows]).max() < FP32_COS_TOL
    assert np.abs(o["text_embeds"] - g["text_embeds"][rows]).max() < FP32_COS_TOL
    scale = np.exp(np.float64(sd["logit_scale"]))
    assert np.abs(o["logits_per_image"] - g["logits_per_image"][rows][:, rows]).max() / scale < FP32_COS_TOL
    # the fixture is self-consistent: logits = scale * img @ txt^T over all 65 536 entries
    full = (g["image_embeds"].astype(np.float64) @ g["text_embeds"].astype(np.float64).T) * scale
    assert np.abs(full - g["logits_per_image"]).max() / scale < FP32_COS_TOL


@pytest.mark.parametrize("name", FULL_CASES)
def test_oracle_matches_golden(name, golden):
    g = golden(name)
    cfg, sd, px, ids, mask = case_inputs(name)
    use_mask = None if "zero_pad" in name else mask
    o = O.clip_forward(px, ids, sd, cfg, use_mask)
    for k in ("image_features", "text_features"):
        assert np.abs(o[k] - g[k]).max() < FP32_FEAT_TOL, k
    for k in ("image_embeds", "text_embeds"):
        assert np.abs(o[k] - g[k]).max() < FP32_COS_TOL, k
    scale = np.exp(np.float64(sd["logit_scale"]))
    assert np.abs(o["logits_per_image"] - g["logits_per_image"]).max() / scale < FP32_COS_TOL
    np.testing.assert_array_equal(o["logits_per_image"], o["logits_per_text"].T)
    np.testing.assert_array_equal(o["logits_per_image"].argmax(1), g["logits_per_image"].argmax(1))


def test_oracle_hidden_states_tiny(golden):
    g = golden("tiny_b6")
    cfg, sd, px, ids, mask = case_inputs("tiny_b6")
    _, vh = O.vision_tower(px, sd, cfg, return_hidden=True)
    _, th = O.text_tower(ids, sd, cfg, mask, return_hidden=True)
    assert len(vh) == cfg.v_layers + 1 and len(th) == cfg.t_layers + 1
    assert np.abs(np.stack(vh) - g["vision_hidden"]).max() < 2e-5
    # rows after the EOS see padding keys only through the mask; compare the attended rows
    m = mask[None, :, :, None].astype(np.float32)
    assert np.abs((np.stack(th) - g["text_hidden"]) * m).max() < 2e-5


def test_oracle_hidden_states_vitb32(golden):
    g = golden("vitb32_b4")
    cfg, sd, px, ids, mask = case_inputs("vitb32_b4")
    _, vh = O.vision_tower(px, sd, cfg, return_hidden=True)
    _, th = O.text_tower(ids, sd, cfg, mask, return_hidden=True)
    assert np.abs(np.stack([h[:, 0] for h in vh]) - g["vision_hidden_cls"]).max() < 1e-4
    assert np.abs(np.stack([h[:, -1] for h in vh]) - g["vision_hidden_last_token"]).max() < 1e-4
    assert np.abs(np.stack([h[:, 0] for h in th]) - g["text_hidden_bos"]).max() < 1e-4
    assert np.abs(np.stack([h[:, 1] for h in th]) - g["text_hidden_tok1"]).max() < 1e-4


def test_oracle_fp64_error_budget():
    """fp32 vs fp64 restatement: the reference-precision noise floor the GPU tolerances sit above."""
    cfg, sd, px, ids, mask = case_inputs("tiny_b6")
    o32 = O.clip_forward(px, ids, sd, cfg, mask, dtype=np.float32)
    o64 = O.clip_forward(px, ids, sd, cfg, mask, dtype=np.float64)
    assert np.abs(o32["image_embeds"] - o64["image_embeds"]).max() < 1e-6
    assert np.abs(o32["text_embeds"] - o64["text_embeds"]).max() < 1e-6


def test_padding_mask_is_irrelevant_to_pooled_output():
    """Causal attention + EOS pooling: the tokenizer's padding mask cannot change the pooled
    embedding -- which is why the OpenAI path can run without one."""
    cfg, sd, px, ids, mask = case_inputs("tiny_b6")
    a = O.text_tower(ids, sd, cfg, mask)
    b = O.text_tower(ids, sd, cfg, None)
    np.testing.assert_array_equal(a, b)


def test_eos_rules():
    ids = np.array([[5, 9, 7, 9, 0], [0, 1, 2, 3, 4], [9, 9, 9, 9, 9]])
    np.testing.assert_array_equal(O.eos_positions(ids, 9), [1, 0, 0])      # first eos, 0 if absent
    np.testing.assert_array_equal(O.eos_positions(ids, 2), [1, 4, 0])      # legacy: first arg-max
    np.testing.assert_array_equal(O.eos_positions(ids, -1), [1, 4, 0])


def test_plip_cosine_similarity_quirk():
    """plip.py:73-76 normalises only the key vectors."""
    rs = np.random.RandomState(0)
    k, s = rs.randn(4, 8).astype(np.float32), rs.randn(3, 8).astype(np.float32)
    got = O.plip_cosine_similarity(k, s)
    want = (k / np.linalg.norm(k, axis=-1, keepdims=True)) @ s.T
    np.testing.assert_allclose(got, want, rtol=1e-6)


@pytest.mark.skipif(not H.available(), reason="transformers not importable")
def test_oracle_matches_live_hf_tiny():
    """Live run of the reference's third-party forward (both attention implementations)."""
    cfg, sd, px, ids, mask = case_inputs("tiny_b5_zero_pad_ln100")
    o = O.clip_forward(px, ids, sd, cfg, None)
    for impl in ("sdpa", "eager"):
        h = H.run(H.build_model(cfg, sd, impl), px, ids, None)
        assert np.abs(o["image_features"] - h["image_features"]).max() < FP32_FEAT_TOL
        assert np.abs(o["text_features"] - h["text_features"]).max() < FP32_FEAT_TOL
        assert np.abs(o["logits_per_image"] - h["logits_per_image"]).max() / 100.0 < FP32_COS_TOL


# ---- the bf16 engine's precision plan, costed on the CPU (oracle/precision_model.py) ----------------------------
@pytest.mark.parametrize("name", ["vitb32_b4", "vitb32_b8_heavy"])
@pytest.mark.parametrize("plan", ["round_ln", "folded"])
def test_bf16_precision_plans_hold_the_cosine_bar(name, plan, golden):
    """bf16 MFMA operands + fp32 everything else, emulated in numpy against the HF golden vectors: both ways of
    pairing LayerNorm with the following Linear (normalise-then-round, and LayerNorm folded into the weights with
    the statistics applied in the GEMM epilogue) stay inside the north-star 1e-3 cosine bar -- also on the
    heavy-tailed checkpoint (outlier channels, LayerNorm gains over two decades, logit_scale ln 100)."""
    from oracle import precision_model as P
    g = golden(name)
    cfg, sd, px, ids, mask = case_inputs(name)
    n = 4
    o = P.clip_forward(px[:n], ids[:n], sd, cfg, mask[:n], plan)
    scale = np.exp(np.float64(sd["logit_scale"]))
    err = np.abs(o["logits_per_image"] - g["logits_per_image"][:n, :n]).max() / scale
    assert err < 1e-3, (name, plan, err)
    assert np.abs(o["image_embeds"] - g["image_embeds"][:n]).max() < 2e-3
    assert np.abs(o["text_embeds"] - g["text_embeds"][:n]).max() < 2e-3


def test_operand_rounding_floor_of_the_text_tower(golden):
    """WHY the bf16 engine's text_embeds sit 3x further from HF than its image_embeds at the benchmark size,
    costed on the CPU: 48 captions of the bs=256 fixture through the numpy emulation with ONE class of operand rounding
    at a time, everything else exact.  bf16 weight rounding alone already exceeds 6e-4 (the perturbation is the same for
    every token of a caption, so attention does not average it out); all six bf16 roundings give the 1.2e-3 the engine
    measures -- it is the operand type, not a kernel -- and the same plan in IEEE half (PLIPMI_F16, same MFMA rate) is
    8x closer."""
    from oracle import precision_model as P
    g = golden("vitb32_b256")
    cfg, sd, px, ids, mask = case_inputs("vitb32_b256")
    n = 48
    want = g["text_embeds"][:n]

    def err(dtype, sites):
        e = O.l2_normalize(P.text_tower(ids[:n], sd, cfg, mask[:n], "folded", dtype=dtype, sites=sites))
        d = np.abs(e - want)
        return float(d.max()), float(np.sqrt((d.astype(np.float64) ** 2).mean()))
    exact = err("bf16", ())
    w_only = err("bf16", ("w",))
    acts = err("bf16", ("xa", "qkv", "p", "att", "mlp"))
    full = err("bf16", P.ALL_SITES)
    half = err("f16", P.ALL_SITES)
    print(f"text_embeds max / rms vs HF: exact {exact}, bf16 weights only {w_only}, bf16 activations only {acts}, "
          f"bf16 all {full}, f16 all {half}")
    assert exact[0] < 2e-6
    assert w_only[0] > 6e-4 and w_only[1] > acts[1]          # the weights are the larger half, and alone past 6e-4
    assert 8e-4 < full[0] < 2e-3
    assert abs(full[1] ** 2 - (w_only[1] ** 2 + acts[1] ** 2)) < 0.25 * full[1] ** 2     # the two halves add in quadrature
    assert half[0] < 2.5e-4 and half[1] < full[1] / 6


def test_the_first_text_blocks_carry_the_bf16_error(golden):
    """WHERE in the text tower bf16's operand rounding costs text_embeds, costed on the CPU: the same
    emulation with the operand type chosen per block.  f16 in the FIRST two blocks removes a third of the error, in the
    LAST two nothing -- the residual stream is small at the bottom of the tower, so a block's rounding error is large
    against it and every later LayerNorm carries it along.  Hence plipmi_config.text_f16_layers (leading blocks), default 4."""
    from oracle import precision_model as P
    g = golden("vitb32_b256")
    cfg, sd, px, ids, mask = case_inputs("vitb32_b256")
    n = 24
    want = g["text_embeds"][:n]

    def err(lead=0, dtype="bf16"):
        e = O.l2_normalize(P.text_tower(ids[:n], sd, cfg, mask[:n], "folded", dtype=dtype, lead_f16=lead))
        return float(np.sqrt((np.abs(e - want).astype(np.float64) ** 2).mean()))
    pure, lead2, lead4, allf16 = err(0), err(2), err(4), err(cfg.t_layers)
    # "f16 in the last two blocks": an f16 tower whose first ten blocks are... not expressible as a lead; emulate by hand
    rnd_b, rnd_h = P.Rounding("bf16"), P.Rounding("f16")
    x = P._f(sd, "text_model.embeddings.token_embedding.weight")[ids[:n]] + P._f(sd, "text_model.embeddings.position_embedding.weight")[None, :ids.shape[1]]
    for i in range(cfg.t_layers):
        sub = {k.replace(f".layers.{i}.", ".layers.0."): v for k, v in sd.items() if k.startswith(f"text_model.encoder.layers.{i}.")}
        x = P._layers(x, sub, "text_model", 1, cfg.t_heads, True, mask[:n], cfg.layer_norm_eps, "folded", [],
                      rnd_h if i >= cfg.t_layers - 2 else rnd_b)
    x = O.layer_norm(x, P._f(sd, "text_model.final_layer_norm.weight"), P._f(sd, "text_model.final_layer_norm.bias"), cfg.layer_norm_eps)
    e = O.l2_normalize(x[np.arange(n), O.eos_positions(ids[:n], cfg.eos_token_id)] @ P._f(sd, "text_projection.weight").T)
    last2 = float(np.sqrt((np.abs(e - want).astype(np.float64) ** 2).mean()))
    print(f"text_embeds rms vs HF: pure bf16 {pure:.2e}, f16 in the first 2 / 4 blocks {lead2:.2e} / {lead4:.2e}, in the last 2 {last2:.2e}, "
          f"in all {allf16:.2e}")
    assert allf16 < lead4 < lead2 < pure
    assert lead2 < 0.8 * pure and last2 > 0.93 * pure


@pytest.mark.parametrize("name", ["vitb32_b4", "vitb32_b8_heavy"])
def test_f16_precision_plan(name, golden):
    """The PLIPMI_F16 engine's plan (IEEE-half MFMA operands, LayerNorm folded) against the HF golden vectors, emulated:
    a quarter of the cosine bar, also on the heavy-tailed checkpoint whose residual stream reaches |x| ~ 90."""
    from oracle import precision_model as P
    g = golden(name)
    cfg, sd, px, ids, mask = case_inputs(name)
    n = 4
    o = P.clip_forward(px[:n], ids[:n], sd, cfg, mask[:n], "folded", dtype="f16")
    scale = np.exp(np.float64(sd["logit_scale"]))
    err = np.abs(o["logits_per_image"] - g["logits_per_image"][:n, :n]).max() / scale
    assert err < 2.5e-4, (name, err)
    assert np.abs(o["image_embeds"] - g["image_embeds"][:n]).max() < 4e-4
    assert np.abs(o["text_embeds"] - g["text_embeds"][:n]).max() < 4e-4


def test_bf16_rounding_helper_is_rne():
    import torch
    from oracle.precision_model import bf16
    x = np.random.RandomState(0).standard_normal(4096).astype(np.float32) * np.float32(37.0)
    x[:4] = [1.00390625, 1.01171875, -1.00390625, 3.3895314e38]       # ties (to even) and a value near the top
    np.testing.assert_array_equal(bf16(x), torch.from_numpy(x).bfloat16().float().numpy())


# ---- the OpenAI-clip checkpoint layout, pinned independently of the converter's own inverse ------------------------
def test_openai_layout_converter_against_independent_openai_style_model():
    """reproducibility/embedders/factory.py:21-25 loads OpenAI-clip state dicts (packed ``attn.in_proj_weight``,
    ``visual.proj [width, proj]`` applied as ``x @ proj``, ``text_projection [width, proj]``).  oracle/openai_clip_ref.py
    is an OpenAI-STYLE model built from torch.nn.MultiheadAttention whose ``state_dict()`` carries that layout by
    construction; its own forward is the expected output.  Its state dict goes through
    ``convert_openai_state_dict`` into (a) HF ``CLIPModel`` and (b) the numpy oracle: both must reproduce it."""
    import torch

    from oracle import openai_clip_ref as R
    from plip_amd import weights as W
    from plip_amd.config import get_config
    cfg0 = get_config("tiny")
    ref = R.build_random(cfg0, seed=3)
    oa = R.numpy_state_dict(ref)
    assert oa["visual.transformer.resblocks.0.attn.in_proj_weight"].shape == (3 * cfg0.v_width, cfg0.v_width)
    assert oa["visual.proj"].shape == (cfg0.v_width, cfg0.projection_dim)
    assert W.is_openai_state_dict(oa)
    sd, cfg = W.normalize_state_dict(oa)
    assert cfg.eos_token_id == 2 and cfg.v_layers == cfg0.v_layers and cfg.t_heads == cfg0.t_heads
    rs = np.random.RandomState(4)
    px = rs.standard_normal((5, 3, cfg.image_size, cfg.image_size)).astype(np.float32)
    ids, _ = W.synthetic_ids(cfg0, 5, seed=6, pad="zero")           # clip.tokenize pads with 0, EOT = highest id
    with torch.no_grad():
        want_i = ref.encode_image(torch.from_numpy(px)).numpy()
        want_t = ref.encode_text(torch.from_numpy(ids)).numpy()
        want_lpi, want_lpt = (t.numpy() for t in ref(torch.from_numpy(px), torch.from_numpy(ids)))
    o = O.clip_forward(px, ids, sd, cfg, None)
    assert np.abs(o["image_features"] - want_i).max() < 2e-5
    assert np.abs(o["text_features"] - want_t).max() < 2e-5
    scale = float(np.exp(np.float64(sd["logit_scale"])))
    assert np.abs(o["logits_per_image"] - want_lpi).max() / scale < FP32_COS_TOL
    np.testing.assert_array_equal(want_lpi, want_lpt.T)
    if H.available():
        h = H.run(H.build_model(cfg, sd, "sdpa"), px, ids, None)
        assert np.abs(h["image_features"] - want_i).max() < 2e-5
        assert np.abs(h["text_features"] - want_t).max() < 2e-5
        assert np.abs(h["logits_per_image"] - want_lpi).max() / scale < FP32_COS_TOL
    # a converter that mis-ordered q|k|v or forgot a transpose must NOT pass: scramble and expect a visible miss
    bad = dict(oa)
    w = bad["visual.transformer.resblocks.0.attn.in_proj_weight"]
    D = w.shape[1]
    bad["visual.transformer.resblocks.0.attn.in_proj_weight"] = np.concatenate([w[D:2 * D], w[:D], w[2 * D:]], 0)
    sd_bad, _ = W.normalize_state_dict(bad)
    assert np.abs(O.vision_tower(px, sd_bad, cfg) - want_i).max() > 1e-3


def test_config3_zero_shot_fixture_is_the_reference_head(golden):
    """tests/golden/config3_zero_shot.npz (HF CLIPModel, oracle/make_config3_fixture.py): inputs regenerate from seeds, the
    oracle reproduces HF's scores on a few tiles, and the reference's head (zero_shot.py:12-13: dot product, per-row
    arg-max) populates at least five of the ten classes -- the property VERDICT r2 found missing."""
    from oracle.make_config3_fixture import tiles_to_pixels
    from plip_amd import weights as W
    from plip_amd.config import get_config
    g = golden("config3_zero_shot")
    tile_seed, prompt_seed, weight_seed, n, pool = (int(v) for v in g["seeds"])
    cfg = get_config("ViT-B/32")
    sd = W.synthetic_state_dict(cfg, weight_seed)
    u8 = W.synthetic_tiles(cfg, n, tile_seed)
    np.testing.assert_allclose([u8.astype(np.float64).sum(), (u8.astype(np.float64) ** 2).sum()], g["tiles_fingerprint"],
                               rtol=0, atol=0)
    cand, _ = W.synthetic_ids(cfg, pool, seed=prompt_seed)
    np.testing.assert_array_equal(cand[g["candidate_index"]], g["prompts"])
    rows = [0, 100, 511]
    img = O.l2_normalize(O.vision_tower(tiles_to_pixels(u8[rows]), sd, cfg))
    txt = O.l2_normalize(O.text_tower(g["prompts"], sd, cfg, None))
    assert np.abs(img @ txt.T - g["scores"][rows]).max() < FP32_COS_TOL
    np.testing.assert_array_equal(g["scores"].argmax(1), g["argmax"])
    hist = np.bincount(g["argmax"], minlength=10)
    assert (hist >= n // 25).sum() >= 5, hist
