"""The CPU oracle (oracle/clip_oracle.py) pinned against the golden vectors produced by the
reference's arithmetic (HF CLIPModel, oracle/make_golden.py).  CPU only."""
import numpy as np
import pytest

from oracle import clip_oracle as O
from oracle import hf_reference as H
from oracle.make_golden import CASES, case_inputs, fingerprint

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


@pytest.mark.parametrize("name", list(CASES))
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
