"""The reference-facing surfaces on the GPU: PLIP class (plip.py), HF-style and
OpenAI-clip-style model objects, the similarity / top-k heads, error behaviour."""
import numpy as np
import pytest
import torch

from oracle import clip_oracle as O

pytestmark = pytest.mark.gpu


def fake_tokenizer(cfg):
    """Deterministic stand-in for CLIPTokenizer (no vocab files offline): hash words to ids."""
    def fn(texts, context_length):
        ids = np.full((len(texts), context_length), cfg.eos_token_id, dtype=np.int64)
        mask = np.zeros_like(ids)
        for i, t in enumerate(texts):
            toks = [cfg.bos_token_id] + [1 + (sum(map(ord, w)) * 31 + len(w)) % (cfg.bos_token_id - 2)
                                         for w in t.lower().split()][: context_length - 2]
            toks.append(cfg.eos_token_id)
            ids[i, : len(toks)] = toks
            mask[i, : len(toks)] = 1
        return ids, mask
    return fn


def test_plip_class_matches_oracle(engines):
    from plip_amd.plip import PLIP
    model, cfg, sd, px, ids, mask = engines("tiny_b6", "f32")
    plip = PLIP(model=model, tokenizer=fake_tokenizer(cfg))
    rs = np.random.RandomState(5)
    tiles = [rs.randint(0, 256, size=(cfg.image_size, cfg.image_size, 3), dtype=np.uint8) for _ in range(7)]
    from plip_amd.preprocess import CLIP_MEAN, CLIP_STD
    emb = plip.encode_images(tiles, batch_size=3)                      # uint8 HWC tiles -> processor -> engine
    assert emb.shape == (7, cfg.projection_dim) and emb.dtype == np.float32
    pxs = np.stack([((t.astype(np.float32) / 255 - np.float32(CLIP_MEAN)) / np.float32(CLIP_STD)).transpose(2, 0, 1)
                    for t in tiles])
    want = O.vision_tower(pxs, sd, cfg)
    assert np.abs(emb - want).max() < 2e-4                                # UN-normalised (plip.py:53)
    labels = ["an h&e image of tumor", "an h&e image of stroma", "an h&e image of lymphocytes"]
    temb = plip.encode_text(labels, batch_size=2)
    tids, tmask = fake_tokenizer(cfg)(labels, cfg.context_length)
    want_t = O.text_tower(tids, sd, cfg, tmask)
    assert np.abs(temb - want_t).max() < 2e-4
    sim = plip._cosine_similarity(emb, temb)                              # key side normalised only
    assert np.abs(sim - O.plip_cosine_similarity(want, want_t)).max() < 2e-4
    preds = plip.zero_shot_classification(tiles, labels)
    assert preds == [labels[i] for i in O.plip_cosine_similarity(want, want_t).argmax(-1)]
    plip.index_images(tiles, batch_size=4)
    nn = plip.retrieval(labels, top_k=5)
    want_nn = np.argsort(-O.plip_cosine_similarity(want_t, want), axis=1, kind="stable")[:, :5]
    np.testing.assert_array_equal(nn, want_nn)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_fused_u8_preprocessing_matches_processor_path(dtype, engines):
    """plipmi_encode_image_u8 == CLIPImageProcessor arithmetic on native tiles + plipmi_encode_image."""
    from plip_amd.preprocess import preprocess_images
    model, cfg, sd, *_ = engines("tiny_b6", dtype)
    rs = np.random.RandomState(9)
    tiles = rs.randint(0, 256, size=(11, cfg.image_size, cfg.image_size, 3), dtype=np.uint8)
    got = model.engine.encode_image_u8(torch.from_numpy(tiles)).cpu().numpy()
    px = preprocess_images(list(tiles), cfg.image_size)
    want = model.engine.encode_image(torch.from_numpy(px)).cpu().numpy()
    # identical op order in fp32; differences only from the (x-mean)*(1/std) vs /std rounding before the bf16 cast
    assert np.abs(got - want).max() < (2e-5 if dtype == "f32" else 3e-2)
    ref = O.vision_tower(px, sd, cfg)
    assert np.abs(got - ref).max() < (2e-4 if dtype == "f32" else 6e-2)
    with pytest.raises(ValueError):
        model.engine.encode_image_u8(torch.zeros(1, 8, 8, 3, dtype=torch.uint8))


def test_openai_style_surface(engines):
    """encode_image / encode_text / model(images, tokens) of reproducibility/embedders/plip.py:48,66."""
    model, cfg, sd, px, ids, mask = engines("tiny_b5_zero_pad_ln100", "f32")
    img = model.encode_image(torch.from_numpy(px))
    txt = model.encode_text(torch.from_numpy(ids).int())                  # clip.tokenize gives int32, 0-padded
    want_t = O.text_tower(ids, sd, cfg.replace(eos_token_id=2), None)      # OpenAI pools at ids.argmax(-1)
    assert np.abs(txt.cpu().numpy() - want_t).max() < 2e-4
    assert np.abs(img.cpu().numpy() - O.vision_tower(px, sd, cfg)).max() < 2e-4
    lpi, lpt = model(torch.from_numpy(px), torch.from_numpy(ids))
    assert lpi.shape == (5, 5) and torch.equal(lpi, lpt.T.contiguous())
    assert abs(float(model.logit_scale) - float(sd["logit_scale"])) < 1e-6
    assert model.eval() is model and model.to("cuda") is model


def test_logits_heads_and_topk(engines):
    model, cfg, *_ = engines("tiny_b6", "f32")
    eng = model.engine
    g = torch.Generator().manual_seed(3)
    img = torch.randn(130, 64, generator=g)
    txt = torch.randn(10, 64, generator=g)                                  # 10 class prompts (config 4)
    lpi, lpt, am = eng.logits(img, txt, scale=2.5, want_argmax=True)
    ref = 2.5 * img.double() @ txt.double().T
    assert (lpi.cpu().double() - ref).abs().max().item() < 1e-4
    assert torch.equal(lpt.cpu(), lpi.cpu().T.contiguous())
    np.testing.assert_array_equal(am.cpu().numpy(), lpi.cpu().numpy().argmax(1))
    sc = torch.randn(9, 300, generator=g)
    sc[0, 5] = sc[0, 17] = 9.0                                              # tie -> lower index first
    idx = eng.topk(sc, 50).cpu().numpy()
    want = np.argsort(-sc.numpy(), axis=1, kind="stable")[:, :50]
    np.testing.assert_array_equal(idx, want)
    x = torch.randn(33, 64, generator=g).cuda()
    y = eng.l2_normalize_(x.clone())
    assert (y - x / x.norm(dim=-1, keepdim=True)).abs().max().item() < 1e-6


def test_error_behaviour(engines):
    model, cfg, sd, px, ids, mask = engines("tiny_b6", "f32")
    with pytest.raises(ValueError):                                         # HF raises ValueError on a wrong image size
        model.get_image_features(pixel_values=torch.zeros(1, 3, cfg.image_size + 16, cfg.image_size + 16))
    with pytest.raises(ValueError):
        model.get_text_features(input_ids=torch.zeros(2, cfg.context_length + 1, dtype=torch.long))
    with pytest.raises(IndexError):                                         # embedding lookup out of range
        model.get_text_features(input_ids=torch.full((1, cfg.context_length), cfg.vocab_size, dtype=torch.long))
    with pytest.raises(ValueError):
        model.get_image_features()
    assert model.get_image_features(pixel_values=torch.zeros(0, 3, cfg.image_size, cfg.image_size)).shape == (0, cfg.projection_dim)


def test_native_library_is_what_runs():
    """The extension must be the in-tree libplipmi.so (the driver records loaded .so files)."""
    from plip_amd import _lib
    lib = _lib.load()
    assert lib.plipmi_version() >= 100
    with open("/proc/self/maps") as f:
        assert "plip_amd/csrc/libplipmi.so" in f.read()
