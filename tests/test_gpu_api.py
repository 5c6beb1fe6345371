"""The reference-facing surfaces on the GPU: PLIP class (plip.py), HF-style and
OpenAI-clip-style model objects, the similarity / top-k heads, error behaviour."""
import numpy as np
import pytest
import torch

from oracle import clip_oracle as O
from oracle.make_golden import case_inputs

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


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16"])
def test_plip_host_loops_coalesce_engine_calls_bit_identically(engines, dtype):
    """The reference drives its heads at batch_size=8 (plip.py:90-91,112); PLIP hands the caller's batches to the towers
    engine.max_batch rows at a time.  A row's embedding does not depend on the batch it travels in, so the coalesced
    loops return the very bits one-engine-call-per-caller-batch does, and zero_shot_classification the same labels."""
    from plip_amd.plip import PLIP
    model, cfg, sd, px, ids, mask = engines("tiny_b6", dtype)
    plip = PLIP(model=model, tokenizer=fake_tokenizer(cfg))
    rs = np.random.RandomState(11)
    tiles = [rs.randint(0, 256, size=(cfg.image_size, cfg.image_size, 3), dtype=np.uint8) for _ in range(21)]
    odd = [rs.randint(0, 256, size=(cfg.image_size + 9, cfg.image_size + 4, 3), dtype=np.uint8) for _ in range(5)]
    labels = [f"an h&e image of tissue class {i} " + "x " * i for i in range(10)]
    pix = torch.from_numpy(px)
    got = {}
    for co in (True, False):
        plip.coalesce = co
        got[co] = (plip.encode_images(tiles, batch_size=2), plip.encode_images(tiles[:3] + odd + tiles[3:6], batch_size=3),
                   plip.encode_images(pix, batch_size=4), plip.encode_text(labels, batch_size=3),
                   plip.zero_shot_classification(tiles, labels))
    del plip.coalesce
    for a, b in zip(got[True][:4], got[False][:4]):
        assert a.shape == b.shape and np.array_equal(a, b)
    assert got[True][4] == got[False][4]


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_coalesced_host_loops_cross_the_tile_boundary_on_vitb32(engines, dtype):
    """The same invariant where it is NOT trivially one tile: on ViT-B/32 a caller batch of 8 images (400 rows) runs on the
    128x128 tile (v_mfma 32x32x16), the coalesced 40 images / 30 captions (2000 / 2310 rows) on the 16x16x32 tiles the cost
    model picks above 1024 rows -- the embeddings must still be the same bits."""
    from plip_amd.plip import PLIP
    model, cfg, sd, *_ = engines("vitb32_b4", dtype, 256)
    plip = PLIP(model=model, tokenizer=fake_tokenizer(cfg))
    rs = np.random.RandomState(21)
    tiles = [rs.randint(0, 256, size=(cfg.image_size, cfg.image_size, 3), dtype=np.uint8) for _ in range(40)]
    labels = [f"an h&e image of tissue class {i} " + "x " * (i % 9) for i in range(30)]
    got = {}
    for co in (True, False):
        plip.coalesce = co
        got[co] = (plip.encode_images(tiles, batch_size=8), plip.encode_text(labels, batch_size=8))
    del plip.coalesce
    for a, b in zip(got[True], got[False]):
        assert a.shape == b.shape and np.array_equal(a, b)


@pytest.mark.parametrize("h,w", [(300, 500), (512, 512), (64, 200), (96, 96), (71, 64), (1000, 700)])
def test_gpu_resize_crop_is_pillow_exact(engines, h, w):
    """plipmi_resize_crop_u8 == Image.resize(BICUBIC) + centre crop, bit for bit, and PLIP.encode_images takes that
    route for equally sized images with the same embeddings as the host preprocessing path."""
    from PIL import Image
    from plip_amd.plip import PLIP
    from plip_amd.preprocess import preprocess_images, resize_crop_plan
    model, cfg, sd, *_ = engines("tiny_b6", "f32")
    n = cfg.image_size
    rs = np.random.RandomState(h + w)
    imgs = rs.randint(0, 256, (5, h, w, 3), dtype=np.uint8)
    for rule in ("torchvision", "hf"):          # OpenAI _transform vs HF CLIPImageProcessor centre crop (odd excess)
        got = model.engine.resize_crop_u8(torch.from_numpy(imgs), crop=rule).cpu().numpy()
        plan = resize_crop_plan(w, h, n, crop=rule)
        for i in range(5):
            im = Image.fromarray(imgs[i]).resize((plan["nw"], plan["nh"]), resample=Image.BICUBIC)
            want = np.asarray(im.crop((plan["left"], plan["top"], plan["left"] + n, plan["top"] + n)))
            np.testing.assert_array_equal(got[i], want)
    plip = PLIP(model=model, tokenizer=fake_tokenizer(cfg))
    a = plip.encode_images(list(imgs), batch_size=3)                                   # GPU resize + fused normalise
    b = model.engine.encode_image(torch.from_numpy(preprocess_images(list(imgs), n, crop="hf"))).cpu().numpy()   # host Pillow path, PLIP's (HF) crop rule
    assert np.abs(a - b).max() < 2e-5
    c = plip.encode_images([Image.fromarray(x) for x in imgs], batch_size=5)
    np.testing.assert_array_equal(a, c)


def test_pipelined_encode_images_is_identical(engines):
    """num_workers > 0: thread-pool decode + pinned double buffers + copy stream must not change a single bit."""
    from PIL import Image
    from plip_amd.plip import PLIP
    from plip_amd.reproducibility import CLIPEmbedder
    model, cfg, *_ = engines("tiny_b6", "f32")
    plip = PLIP(model=model, tokenizer=fake_tokenizer(cfg))
    rs = np.random.RandomState(11)
    native = [rs.randint(0, 256, size=(cfg.image_size, cfg.image_size, 3), dtype=np.uint8) for _ in range(13)]
    odd = [Image.fromarray(rs.randint(0, 256, size=(80 + 3 * i, 100, 3), dtype=np.uint8)) for i in range(9)]   # need resize + crop
    for imgs in (native, [Image.fromarray(t) for t in native], odd):
        a = plip.encode_images(imgs, batch_size=4)
        b = plip.encode_images(imgs, batch_size=4, num_workers=3)
        np.testing.assert_array_equal(a, b)
    emb = CLIPEmbedder(model)
    np.testing.assert_array_equal(emb.embed_images(odd, batch_size=4, num_workers=1),
                                  emb.embed_images(odd, batch_size=4, num_workers=4))


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


def _check_topk(idx, vals, q, sp, k):
    """idx/vals must be a valid descending top-k of q @ sp.T: exact-score comparison with an fp32-ulp allowance
    (the GPU sums the 512 products in MFMA order, numpy in BLAS order, so near-ties may legally swap)."""
    ref = q.astype(np.float64) @ sp.astype(np.float64).T
    got = np.take_along_axis(ref, idx, axis=1)
    want = -np.sort(-ref, axis=1)[:, :k]
    tol = 1e-5 * max(1.0, float(np.abs(ref).max()))
    assert np.abs(got - want).max() < tol                                   # same score profile as the true top-k
    assert (np.diff(got, axis=1) <= tol).all()                              # descending
    assert all(len(set(r)) == k for r in idx.tolist())                      # no duplicates
    assert idx.min() >= 0 and idx.max() < sp.shape[0]
    if vals is not None:
        assert np.abs(vals - got).max() < tol


@pytest.mark.parametrize("nq,ns,d,k", [(7, 10, 64, 3), (130, 300, 64, 50), (33, 8192 + 777, 64, 50),
                                        (4100, 20000, 512, 10), (5, 9000, 512, 1024)])
def test_similarity_topk_streaming(engines, nq, ns, d, k):
    """Fused similarity + top-k (retrieval.py:13-18, plip.py:78-87) without the [Nq,Ns] matrix: several panels,
    ragged tail panel, more queries than one query block, k up to the list capacity."""
    model, *_ = engines("tiny_b6", "f32")
    eng = model.engine
    rng = np.random.RandomState(nq + ns)
    q = rng.randn(nq, d).astype(np.float32)
    sp = rng.randn(ns, d).astype(np.float32)
    idx, vals = eng.similarity_topk(torch.from_numpy(q), torch.from_numpy(sp), k, return_values=True)
    _check_topk(idx.cpu().numpy(), vals.cpu().numpy(), q, sp, k)
    idx2 = eng.similarity_topk(torch.from_numpy(q), torch.from_numpy(sp), k)
    assert torch.equal(idx, idx2)                                           # deterministic, vals optional


def test_similarity_topk_ties_and_errors(engines):
    model, *_ = engines("tiny_b6", "f32")
    eng = model.engine
    sp = np.zeros((600, 64), np.float32)
    sp[:, 0] = 1.0                                                          # every score identical -> index order
    sp[500, 0] = 2.0
    q = np.zeros((3, 64), np.float32)
    q[:, 0] = 1.0
    idx = eng.similarity_topk(torch.from_numpy(q), torch.from_numpy(sp), 5).cpu().numpy()
    np.testing.assert_array_equal(idx, np.tile(np.array([500, 0, 1, 2, 3]), (3, 1)))
    sp[7] = np.nan                                                          # NaN scores sort last
    idx = eng.similarity_topk(torch.from_numpy(q), torch.from_numpy(sp), 600).cpu().numpy()
    assert idx[0, -1] == 7 and idx[0, 0] == 500
    from plip_amd._lib import PlipmiError
    with pytest.raises(PlipmiError):
        eng.similarity_topk(torch.from_numpy(q), torch.from_numpy(sp), 601)    # k > Ns
    with pytest.raises(PlipmiError):
        eng.similarity_topk(torch.zeros(2, 48), torch.zeros(9, 48), 2)         # D % 32
    assert eng.similarity_topk(torch.zeros(0, 64), torch.from_numpy(sp), 4).shape == (0, 4)


def test_clip_embedder_cache_and_eval_heads(engines, tmp_path, monkeypatch):
    """reproducibility/embedders/plip.py + evaluation heads on the engine: normalised rows, both cache schemes,
    zero-shot arg-max and retrieval top-50 equal to the reference's numpy formulation on the oracle's embeddings."""
    import argparse
    from oracle import clip_oracle as O
    from plip_amd import weights as W
    from plip_amd.reproducibility import CLIPEmbedder, EmbedderFactory, ImageRetrieval, ZeroShotClassifier
    model, cfg, sd, px, ids, mask = engines("tiny_b5_zero_pad_ln100", "f32")
    monkeypatch.setenv("PC_CACHE_FOLDER", str(tmp_path))
    emb = CLIPEmbedder(model, None, "plip", "/ckpts/tiny.pt")
    img = emb.image_embedder(px, batch_size=2, additional_cache_name="unit_test.csv")     # NCHW float array input
    txt = emb.text_embedder(ids, batch_size=3)                                             # token ids (0-padded)
    want_img = O.l2_normalize(O.vision_tower(px, sd, cfg))
    want_txt = O.l2_normalize(O.text_tower(ids, sd, cfg))
    assert img.dtype == np.float32 and np.abs(img - want_img).max() < 1e-5 and np.abs(txt - want_txt).max() < 1e-5
    assert (tmp_path / "unit_test" / "plip" / "tiny.pt").exists()
    emb.model = None                                                                        # second call must be a cache hit
    np.testing.assert_array_equal(emb.image_embedder(px, additional_cache_name="unit_test.csv"), img)
    np.testing.assert_array_equal(emb.text_embedder(ids), txt)
    with pytest.raises(RuntimeError):
        CLIPEmbedder(model).embed_text(["a caption"])                                       # strings need a tokenizer
    # PIL / uint8 HWC items go through the preprocess contract (transform.py:45-52)
    u8 = np.random.RandomState(0).randint(0, 256, (3, cfg.image_size, cfg.image_size, 3), dtype=np.uint8)
    from plip_amd.preprocess import preprocess_images
    got = CLIPEmbedder(model).embed_images(list(u8), batch_size=2)
    assert np.abs(got - O.l2_normalize(O.vision_tower(preprocess_images(list(u8), cfg.image_size), sd, cfg))).max() < 1e-5

    # factory: OpenAI-format .pt on disk, arch from $PC_CLIP_ARCH (factory.py:21-25)
    ck = tmp_path / "tiny_openai.pt"
    torch.save({k: torch.from_numpy(np.asarray(v)) for k, v in W.to_openai_state_dict(sd, cfg).items()}, ck)
    monkeypatch.setenv("PC_CLIP_ARCH", "tiny")
    e2 = EmbedderFactory().factory(argparse.Namespace(model_name="plip", backbone=str(ck), dtype="fp32", max_batch=8))
    assert np.abs(e2.embed_images(px) - want_img).max() < 1e-5 and e2.backbone == str(ck)
    e2.model.engine.close()
    with pytest.raises(FileNotFoundError):
        EmbedderFactory().factory(argparse.Namespace(model_name="clip", backbone=str(tmp_path / "missing.pt")))
    with pytest.raises(NotImplementedError):
        EmbedderFactory().factory(argparse.Namespace(model_name="mudipath", backbone=None))

    # evaluation heads
    rng = np.random.RandomState(5)
    I = rng.randn(300, 512).astype(np.float32); I /= np.linalg.norm(I, axis=1, keepdims=True)
    T = I + 0.4 * rng.randn(300, 512).astype(np.float32); T /= np.linalg.norm(T, axis=1, keepdims=True)
    labels = ["adipose", "lymphocytes", "mucus", "tumor"]
    C = T[:4]
    zs = ZeroShotClassifier()
    pred = zs.predict(I, C, labels)
    assert pred == [labels[int(np.argmax(r))] for r in I.dot(C.T)]                         # zero_shot.py:12-13
    target = [labels[i % 4] for i in range(300)]
    tr, te = zs.zero_shot_classification(I, C, labels, target, pickle_path=str(tmp_path / "pickle.pkl"))
    assert tr["split"] == "train" and te["split"] == "test" and te["instances"] == 300
    assert abs(te["Accuracy"] - np.mean([a == b for a, b in zip(pred, target)])) < 1e-12
    import pickle
    assert pickle.load(open(tmp_path / "pickle.pkl", "rb"))["predictions"] == pred
    ir = ImageRetrieval()
    best = ir.best_scores(I, T)
    ref_best = np.stack([t.dot(I.T).argsort()[-50:][::-1] for t in T])                    # retrieval.py:13-16
    _check_topk(best, None, T, I, 50)
    assert (best[:, 0] == ref_best[:, 0]).mean() > 0.99
    tr, te = ir.retrieval(I, T)
    p10 = np.mean([i in ref_best[i, :10] for i in range(300)]); p50 = np.mean([i in ref_best[i] for i in range(300)])
    assert abs(te["p@10"] - p10) < 1e-9 and abs(te["p@50"] - p50) < 1e-9 and tr["split"] == "train"


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


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_large_batches_run_as_equal_passes_with_the_same_bits(engines, dtype):
    """plipmi_config.pass_batch (throughput must not fall with the caller's batch): a call of B >= 2 * pass_batch
    samples runs as ceil(B / pass_batch) equal back-to-back passes, and the embeddings are the bits of the one-pass engine --
    fp32 pixels, uint8 tiles, captions with a mask; a call under 2 * pass_batch stays one pass; 0 resolves to "no splitting" on the
    tiny towers' 1 GB-per-several-thousand-samples working set and to 256 on ViT-B/32."""
    from plip_amd.model import PlipModel
    from plip_amd import weights as W
    _, cfg, sd, *_ = engines("tiny_b6", dtype)
    B = 41                                                     # three passes of 14 / 14 / 13 at pass_batch 16
    px = torch.from_numpy(W.synthetic_pixels(cfg, B, seed=5))
    ids_np, mask_np = W.synthetic_ids(cfg, B, seed=6)
    ids, mask = torch.from_numpy(ids_np), torch.from_numpy(mask_np)
    rs = np.random.RandomState(3)
    tiles = torch.from_numpy(rs.randint(0, 256, size=(B, cfg.image_size, cfg.image_size, 3), dtype=np.uint8))
    one = PlipModel(cfg, sd, dtype=dtype, max_batch=64, pass_batch=-1)
    split = PlipModel(cfg, sd, dtype=dtype, max_batch=64, pass_batch=16)
    try:
        for m in (one, split):
            m.got = (m.engine.encode_image(px, True), m.engine.encode_image_u8(tiles, False), m.engine.encode_text(ids, mask, True),
                     m.engine.encode_image(px[:31], False), m.engine.encode_text(ids[:31], None, False))
            torch.cuda.synchronize()
        for a, b in zip(one.got, split.got):
            assert a.shape == b.shape and torch.equal(a, b)
        # both towers on two streams (Engine.encode_pair): the host layer cuts the batch itself, pass by pass with the towers joined
        assert split.engine.pass_batch == 16 and one.engine.pass_batch == 0
        pa, pb = one.engine.encode_pair(px, ids, mask, normalize=True, overlap=True), split.engine.encode_pair(px, ids, mask, normalize=True, overlap=True)
        torch.cuda.synchronize()
        assert torch.equal(pa[0], pb[0]) and torch.equal(pa[1], pb[1]) and torch.equal(pa[0], one.got[0]) and torch.equal(pa[1], one.got[2])
        if dtype == "bf16":                                    # packed captions: every pass packs its own captions; same bits again
            for m in (one, split):
                m.engine.set_text_packing(True)
                m.packed = m.engine.encode_text(ids, mask, True)
                m.engine.set_text_packing(False)
            torch.cuda.synchronize()
            assert torch.equal(one.packed, split.packed) and torch.equal(one.packed, one.got[2])
        # the automatic rule on the benchmark model: 256 on the 16-bit engines, never on the fp32 engine (only 128 samples' activations
        # fit the cache, and passes that small cost the GEMMs more than the cache returns: 10.1 k instead of 12.5 k img/s at bs = 256)
        assert engines("vitb32_b4", dtype, 256)[0].engine.pass_batch == (256 if dtype == "bf16" else 0)
        rows = []
        with split.engine.profile(rows):                       # 41 samples: 3 passes -> 3 patch GEMMs; 31 < 2 * 16: one
            split.engine.encode_image(px, True)
        assert sum(r["calls"] for r in rows if "patch_embed" in r["name"]) == 3
        rows = []
        with split.engine.profile(rows):
            split.engine.encode_image(px[:31], True)
        assert sum(r["calls"] for r in rows if "patch_embed" in r["name"]) == 1
    finally:
        one.engine.close()
        split.engine.close()


def test_pair_stream_is_measured_to_run_beside_the_callers_stream(engines):
    """HIP streams share hardware queues, and two streams on one queue run in order: Engine.encode_pair's second stream
    is chosen by plipmi_streams_overlap, per caller's stream.  A stream against itself measures ~2 (in order), the chosen one ~1;
    whatever the process did to the stream -> queue mapping before (here: many streams created first), a caller's stream gets a
    partner that overlaps with it, and the pair on it has the bits of the one-stream pair."""
    model, cfg, sd, px, ids, mask = engines("tiny_b6", "bf16")
    eng = model.engine
    dev = eng.device
    main = torch.cuda.current_stream(dev)
    assert eng.streams_overlap(main, main) > 1.7
    side = eng.pair_stream(main)
    assert side.cuda_stream != main.cuda_stream and eng.pair_stream(main) is side             # measured once, then kept
    assert eng.streams_overlap(main, side) < 1.5
    held = [torch.cuda.Stream(device=dev) for _ in range(11)]                                 # shifts where the next streams land
    ratios = [eng.streams_overlap(held[0], s) for s in held[1:]]
    assert min(ratios) < 1.5                                                                  # some pairs overlap ...
    mine = held[0]
    partner = eng.pair_stream(mine)
    assert partner.cuda_stream != mine.cuda_stream and eng.streams_overlap(mine, partner) < 1.5
    pxd, idd, md = (torch.as_tensor(a).to(dev) for a in (px, ids, mask))
    ref = eng.encode_pair(pxd, idd, md, normalize=True, overlap=False)
    mine.wait_stream(main)
    with torch.cuda.stream(mine):
        got = eng.encode_pair(pxd, idd, md, normalize=True, overlap=True)
    main.wait_stream(mine)
    torch.cuda.synchronize()
    assert torch.equal(ref[0], got[0]) and torch.equal(ref[1], got[1])
    aliased = [r for r in ratios if r > 1.7]
    print("overlap ratios of held[0] against ten later streams:", [round(r, 2) for r in ratios], "sharing a queue:", len(aliased))


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_clone_shares_the_weights_and_lanes_keep_the_bits(dtype):
    """plipmi_clone: a second handle on the same packed weights with a workspace of its own.  Its embeddings are the source's bits; the
    weights outlive the source (destroyed first here); calls of more than max_batch rows run their chunks alternately on the engine
    and its clone on two streams (Engine.lanes) with the bits of the one-lane call -- images, uint8 tiles, captions --; the setters
    reach the clone; a token id out of range that a CLONE's embedding kernel met is reported by the engine; Engine.lane_loop and the
    PLIP host loops give the plain loops' results."""
    from plip_amd.model import PlipModel
    from plip_amd.plip import PLIP
    from plip_amd import weights as W
    cfg, sd, *_ = case_inputs("tiny_b6")
    B = 37                                                     # max_batch 8: five chunks, the last one ragged
    px = torch.from_numpy(W.synthetic_pixels(cfg, B, seed=15))
    ids_np, mask_np = W.synthetic_ids(cfg, B, seed=16)
    ids, mask = torch.from_numpy(ids_np), torch.from_numpy(mask_np)
    tiles = torch.from_numpy(np.random.RandomState(4).randint(0, 256, size=(B, cfg.image_size, cfg.image_size, 3), dtype=np.uint8))
    model = PlipModel(cfg, sd, dtype=dtype, max_batch=8)
    eng = model.engine
    try:
        eng.use_lanes = False
        want = (eng.encode_image(px, True), eng.encode_image_u8(tiles, False), eng.encode_text(ids, mask, True), eng.encode_text(ids, None, False))
        torch.cuda.synchronize()
        assert getattr(eng, "_lanes", None) is None            # nothing was cloned for the one-lane calls
        # a clone by itself, used after its source is gone
        src = PlipModel(cfg, sd, dtype=dtype, max_batch=8).engine
        twin = src.clone()
        src.close()
        twin.use_lanes = False
        assert torch.equal(twin.encode_image(px, True), want[0]) and torch.equal(twin.encode_text(ids, mask, True), want[2])
        twin.close()
        # chunks on two lanes
        eng.use_lanes = True
        got = (eng.encode_image(px, True), eng.encode_image_u8(tiles, False), eng.encode_text(ids, mask, True), eng.encode_text(ids, None, False))
        torch.cuda.synchronize()
        assert eng._lanes is not None and len(eng._lanes) == 1
        for a, b in zip(want, got):
            assert torch.equal(a, b)
        rows = []
        with eng.profile(rows):                                # profiling keeps every chunk on the profiled handle
            eng.encode_image(px, True)
        assert sum(r["calls"] for r in rows if "patch_embed" in r["name"]) == 5
        # a bad id in the FOURTH chunk (rows 24..31 -> the clone's lane, its last chunk of the call) is the engine's to report, once
        bad = ids.to(eng.device).clone()
        bad[25, 2] = cfg.vocab_size
        eng.encode_text(bad, None)
        with pytest.raises(IndexError):
            eng.check_async()
        eng.check_async()
        eng.encode_text(bad, None)
        torch.cuda.synchronize()
        with pytest.raises(IndexError):
            eng.encode_text(ids, None)
        assert torch.equal(eng.encode_text(ids, None, False), want[3])
        # setters reach the clone: packed captions (bit-identical by construction) and the latency path (tolerance) on every lane
        if dtype == "bf16":
            eng.set_text_packing(True)
            assert torch.equal(eng.encode_text(ids, mask, True), want[2])
            eng.set_text_packing(False)
        # the host loop form
        with eng.lane_loop() as run:
            parts = [run(lambda e, a=a: e.encode_image_u8(tiles[a:a + 8], False)) for a in range(0, B, 8)]
        assert torch.equal(torch.cat(parts), want[1])
        plip = PLIP(model=model, tokenizer=None)
        imgs = [t.numpy() for t in tiles]
        two = (plip.encode_images(imgs, batch_size=8), plip.encode_text(ids_np, batch_size=8), plip.encode_images(imgs, batch_size=8, num_workers=2))
        eng.use_lanes = False
        one = (plip.encode_images(imgs, batch_size=8), plip.encode_text(ids_np, batch_size=8), plip.encode_images(imgs, batch_size=8, num_workers=2))
        for a, b in zip(one, two):
            assert np.array_equal(a, b)
        assert np.array_equal(one[0], one[2]) and np.array_equal(one[0], want[1].cpu().numpy())
    finally:
        eng.close()


def test_config_struct_size_lets_the_struct_grow(engines):
    """plipmi_config starts with its own size: a caller compiled against an OLDER, shorter header -- one that
    ends at max_batch -- gets every later member as 0 (the product defaults), not whatever lies behind its struct; sizes
    that cannot be a plipmi_config are rejected."""
    import ctypes as C
    from plip_amd import _lib
    from plip_amd._lib import PlipmiError
    from plip_amd.model import PlipModel
    model, cfg, sd, px, ids, mask = engines("tiny_b6", "bf16")
    short = _lib.Config.max_batch.offset + 4
    assert short < C.sizeof(_lib.Config)
    # flags = 0, graph_batch = 0 (default replay), text_f16_layers = 0 (pure bf16) is what the short struct must mean --
    # although the binding FILLS the tail with other values (text_f16_layers = 2), which the library must not read
    old_caller = PlipModel(cfg, sd, dtype="bf16", max_batch=32, text_f16_layers=2, _config_struct_size=short)
    same = PlipModel(cfg, sd, dtype="bf16", max_batch=32, text_f16_layers=0)
    mixed = PlipModel(cfg, sd, dtype="bf16", max_batch=32, text_f16_layers=2)
    t = [m.get_text_features(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask)) for m in (old_caller, same, mixed)]
    assert torch.equal(t[0], t[1]) and not torch.equal(t[0], t[2])
    for m in (old_caller, same, mixed):
        m.engine.close()
    for bad in (0, 8, short - 4, C.sizeof(_lib.Config) + 4):
        with pytest.raises(PlipmiError, match="struct_size"):
            PlipModel(cfg, sd, dtype="bf16", max_batch=32, _config_struct_size=bad)


def test_native_library_is_what_runs():
    """The extension must be the in-tree libplipmi.so (the driver records loaded .so files)."""
    from plip_amd import _lib
    lib = _lib.load()
    assert lib.plipmi_version() >= 300
    with open("/proc/self/maps") as f:
        assert "plip_amd/csrc/libplipmi.so" in f.read()


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16"])
def test_small_batch_graph_replay_is_bit_identical(dtype, engines):
    """plipmi_set_graph_batch: encode calls of <= 32 samples replay a captured hipGraph from the third call of a shape on
    (call 1 eager, call 2 capture + launch).  Same kernels in the same order: results must be BIT-identical to the eager
    path, for fresh inputs at fresh addresses on every call, for every input kind, with and without a mask."""
    model, cfg, sd, px, ids, mask = engines("tiny_b6", dtype)
    eng = model.engine
    rs = np.random.RandomState(3)
    for B in (1, 5):
        want, got = [], []
        for rep in range(4):                        # eager reference first (graphs off), fresh tensors each time
            p = torch.from_numpy(rs.standard_normal((B, 3, cfg.image_size, cfg.image_size)).astype(np.float32))
            u = torch.from_numpy(rs.randint(0, 256, (B, cfg.image_size, cfg.image_size, 3), dtype=np.uint8))
            t = torch.from_numpy(np.ascontiguousarray(ids[(rep + np.arange(B)) % len(ids)]))
            m = torch.from_numpy(np.ascontiguousarray(mask[(rep + np.arange(B)) % len(ids)]))
            want.append((p, u, t, m))
        eng.set_graph_batch(0)
        ref = [(eng.encode_image(p, True).clone(), eng.encode_image_u8(u).clone(), eng.encode_text(t, m, True).clone(),
                eng.encode_text(t, None, False, eos_token_id=-1).clone()) for (p, u, t, m) in want]
        eng.set_graph_batch(32)
        for (p, u, t, m) in want:
            got.append((eng.encode_image(p.clone(), True).clone(), eng.encode_image_u8(u.clone()).clone(),
                        eng.encode_text(t.clone(), m.clone(), True).clone(),
                        eng.encode_text(t.clone(), None, False, eos_token_id=-1).clone()))
        torch.cuda.synchronize()
        for r, g in zip(ref, got):
            for a, b in zip(r, g):
                assert torch.equal(a, b)
    # profiling switches replay off for the bracketed calls (events cannot sit inside a replayed graph), results unchanged
    rows = []
    p = want[0][0]
    with eng.profile(rows):
        a = eng.encode_image(p, True).clone()
    assert any(r["name"].startswith("gemm_nt") for r in rows) and torch.equal(a, eng.encode_image(p, True))
    # a batch above the threshold is untouched by all this
    eng.set_graph_batch(2)
    big = torch.from_numpy(px)
    assert torch.equal(eng.encode_image(big), eng.encode_image(big))
    eng.set_graph_batch(32)


@pytest.mark.parametrize("half", ["bf16", "f16"])
@pytest.mark.parametrize("case,max_batch", [("tiny_b6", 32), ("vitb32_b4", 32), ("vitb32_b4", 256)])
def test_packed_captions_are_bit_identical(case, max_batch, half, engines):
    """plipmi_set_text_packing: the text tower runs on rows 0 .. EOS of every caption only (causal attention + EOS pooling:
    the padding behind EOS cannot reach text_embeds).  Lengths, row offsets and the live-row count never leave the device.
    The embeddings must equal the padded computation BIT FOR BIT -- for both EOS rules (modeling_clip.py:561-581), both
    padding conventions (HF: EOS, OpenAI clip.tokenize: 0), with and without the tokenizer's mask, for captions of one
    token and of the full context, and when a captured graph is replayed on captions of other lengths."""
    from plip_amd import weights as W
    model, cfg, sd, px, ids0, mask0 = engines(case, half, max_batch)
    eng = model.engine
    S = cfg.context_length
    B = 256 if max_batch == 256 else 8
    sets = []
    for seed, pad in ((11, "eos"), (12, "zero"), (13, "eos")):
        ids, mask = W.synthetic_ids(cfg, B, seed=seed, pad=pad)
        sets.append((ids, mask))
    # hand-made edge rows: EOS right after BOS, EOS in the last position, no EOS at all (explicit rule pools row 0)
    ids, mask = W.synthetic_ids(cfg, B, seed=14)
    ids[0, :] = cfg.eos_token_id; ids[0, 0] = cfg.bos_token_id; mask[0, :] = 0; mask[0, :2] = 1
    ids[1, 1:S - 1] = 5; ids[1, S - 1] = cfg.eos_token_id; mask[1, :] = 1
    if B > 2:
        ids[2, 1:] = 7; mask[2, :] = 1
    sets.append((ids, mask))
    try:
        for gb in (0, 32):
            eng.set_graph_batch(gb)
            for rule in (None, -1):                               # config's eos_token_id / the legacy argmax rule
                want = []
                eng.set_text_packing(False)
                for ids, mask in sets:
                    t, m = torch.from_numpy(ids), torch.from_numpy(mask)
                    want.append((eng.encode_text(t, m, True, eos_token_id=rule).clone(),
                                 eng.encode_text(t, None, False, eos_token_id=rule).clone()))
                eng.set_text_packing(True)
                for (ids, mask), (w_mask, w_plain) in zip(sets, want):
                    t, m = torch.from_numpy(ids), torch.from_numpy(mask)
                    g_mask = eng.encode_text(t, m, True, eos_token_id=rule)
                    g_plain = eng.encode_text(t, None, False, eos_token_id=rule)
                    torch.cuda.synchronize()
                    assert torch.isfinite(g_mask).all() and torch.isfinite(g_plain).all()
                    assert torch.equal(g_mask, w_mask), (case, gb, rule)
                    assert torch.equal(g_plain, w_plain), (case, gb, rule)
        # both towers of a step on two streams, as bench.py times them
        ids, mask = sets[0]
        eng.set_text_packing(False)
        i0, t0 = eng.encode_pair(torch.from_numpy(px[:B] if len(px) >= B else np.resize(px, (B,) + px.shape[1:])),
                                 torch.from_numpy(ids), torch.from_numpy(mask), True, overlap=True)
        eng.set_text_packing(True)
        i1, t1 = eng.encode_pair(torch.from_numpy(px[:B] if len(px) >= B else np.resize(px, (B,) + px.shape[1:])),
                                 torch.from_numpy(ids), torch.from_numpy(mask), True, overlap=True)
        torch.cuda.synchronize()
        assert torch.equal(i0, i1) and torch.equal(t0, t1)
    finally:
        eng.set_text_packing(False)
        eng.set_graph_batch(32)


def test_text_packing_needs_a_pooled_16bit_engine(engines):
    model, *_ = engines("tiny_b6", "f32")
    with pytest.raises(RuntimeError):
        model.engine.set_text_packing(True)
    model.engine.set_text_packing(False)                         # switching it off is always allowed


def test_plip_class_pack_captions_is_bit_identical(engines):
    """PLIP(..., pack_captions=True) (extension of plip.py:17-29): same text embeddings as the padded engine, bit for bit,
    through the reference's own encode_text flow (tokenise -> batches -> get_text_features)."""
    from plip_amd.plip import PLIP
    model, cfg, sd, px, ids, mask = engines("tiny_b6", "bf16")
    texts = ["an h&e image of tumor", "lymphocytes", "a", "normal colon mucosa with crypts and goblet cells " * 3]
    try:
        want = PLIP(model=model, tokenizer=fake_tokenizer(cfg)).encode_text(texts, batch_size=3)
        got = PLIP(model=model, tokenizer=fake_tokenizer(cfg), pack_captions=True).encode_text(texts, batch_size=3)
    finally:
        model.engine.set_text_packing(False)
    assert np.array_equal(np.asarray(want), np.asarray(got))


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16"])
def test_device_resident_out_of_range_ids_surface_an_error(dtype, engines):
    """plip.py:68 -> HF nn.Embedding raises on a token id outside the vocabulary (on a GPU: a device-side assert that
    surfaces at the next synchronisation).  Ids that already live on the device are range-checked BY the embedding kernel;
    the error surfaces at check_async() / the next encode_text call, exactly once (never from an image-side call), and a CPU
    tensor still raises at once."""
    model, cfg, sd, px, ids, mask = engines("tiny_b6", dtype)
    eng = model.engine
    good = torch.from_numpy(ids).to(eng.device)
    want = eng.encode_text(good, None).clone()
    eng.check_async()                                             # nothing pending
    for bad_value in (cfg.vocab_size, -1, 2 ** 40):
        bad = good.clone()
        bad[2, 3] = bad_value
        eng.encode_text(bad, None)                                # enqueues; the ids are device memory
        with pytest.raises(IndexError):
            eng.check_async()
        eng.check_async()                                         # reported once
        eng.encode_text(bad, None)
        torch.cuda.synchronize()
        img = eng.encode_image(torch.from_numpy(px))              # an unrelated image call is NOT failed by the bad caption
        assert bool(torch.isfinite(img).all())
        with pytest.raises(IndexError):                           # ... the next encode_text on the handle reports it, once
            eng.encode_text(good, None)
        assert torch.equal(eng.encode_text(good, None), want)    # the handle keeps working
        eng.check_async()
    with pytest.raises(IndexError):
        eng.encode_text(torch.from_numpy(ids).clamp(min=cfg.vocab_size), None)    # host-resident ids: checked up front


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_latency_path_small_batches_on_split_k_gemms(dtype, golden):
    """plipmi_set_latency_batch: batches of at most n samples run every GEMM of their towers -- patch
    embedding, q/k/v, out_proj, fc1, fc2 with the engine's own epilogues -- on the split-K small-M kernel.  Same arithmetic:
    inside the parity bar against HF; a row's embedding independent of its batch INSIDE the regime; against the big-tile
    regime the other summation order moves 16-bit roundings (bounded here), which is why the path is opt-in."""
    from plip_amd.model import PlipModel
    g = golden("vitb32_b4")
    cfg, sd, px, ids, mask = case_inputs("vitb32_b4")
    px, ids, mask = torch.from_numpy(px), torch.from_numpy(ids), torch.from_numpy(mask)
    big = PlipModel(cfg, sd, dtype=dtype, max_batch=32)
    fast = PlipModel(cfg, sd, dtype=dtype, max_batch=32, latency_batch=8)
    try:
        ob = big(input_ids=ids, pixel_values=px, attention_mask=mask)
        of = fast(input_ids=ids, pixel_values=px, attention_mask=mask)
        scale = np.exp(np.float64(sd["logit_scale"]))
        cos = {"bf16": 1e-3, "f16": 2.5e-4}[dtype]
        for o in (ob, of):
            assert np.abs(o.logits_per_image.cpu().numpy() - g["logits_per_image"]).max() / scale < cos
        d = max(float((ob.image_embeds - of.image_embeds).abs().max()), float((ob.text_embeds - of.text_embeds).abs().max()))
        assert 0 < d < {"bf16": 2e-3, "f16": 4e-4}[dtype]                     # another summation order, not another result
        # inside the regime: rows do not depend on their batch (1 of 4, 2 of 4), eager == graph replay (calls 2, 3 replay)
        one = fast.get_image_features(pixel_values=px[2:3])
        full = fast.get_image_features(pixel_values=px)
        assert torch.equal(one, full[2:3]) and torch.equal(fast.get_image_features(pixel_values=px[2:3]), one)
        t_full = fast.get_text_features(input_ids=ids, attention_mask=mask)
        assert torch.equal(fast.get_text_features(input_ids=ids[1:3], attention_mask=mask[1:3]), t_full[1:3])
        assert torch.equal(fast.get_text_features(input_ids=ids, attention_mask=mask), t_full)
        # above the threshold the engine is the big-tile engine, bit for bit
        rs = np.random.RandomState(3)
        px20 = torch.from_numpy(rs.randn(20, 3, cfg.image_size, cfg.image_size).astype(np.float32))
        assert torch.equal(fast.get_image_features(pixel_values=px20), big.get_image_features(pixel_values=px20))
        # hidden states of the latency path against the big path's: a 16-bit rounding apart at most
        hb, hf = big.engine.hidden("text", cfg.t_layers, ids), fast.engine.hidden("text", cfg.t_layers, ids)
        assert float((hb - hf).abs().max()) < {"bf16": 1.5e-1, "f16": 4e-2}[dtype]
    finally:
        big.engine.close()
        fast.engine.close()


def test_engine_switches_are_constructor_arguments_not_environment(engines, monkeypatch):
    """Every behavioural switch of a handle is plipmi_config.flags / a setter: the library reads no environment variable, so
    the variables earlier versions honoured must change nothing."""
    from plip_amd.model import PlipModel
    model, cfg, sd, px, ids, mask = engines("tiny_b6", "bf16")
    tpx = torch.from_numpy(px)
    want = model.get_image_features(pixel_values=tpx)
    for var in ("PLIPMI_LN_FOLD", "PLIPMI_POOLED_LAST_BLOCK", "PLIPMI_ATTENTION", "PLIPMI_GRAPH_BATCH", "PLIPMI_TEXT_PACKING"):
        monkeypatch.setenv(var, "0")
    same = PlipModel(cfg, sd, dtype="bf16", max_batch=8)
    other = PlipModel(cfg, sd, dtype="bf16", max_batch=8, ln_fold=False, mfma_attention=False, graph_batch=0)
    try:
        assert torch.equal(same.get_image_features(pixel_values=tpx), want)
        rows = []
        with other.engine.profile(rows):
            got = other.get_image_features(pixel_values=tpx)
        names = " ".join(r["name"] for r in rows)
        assert "attention_valu" in names and "ln_bias" not in names and sum(r["calls"] for r in rows if r["name"] == "layernorm") > 2
        assert (got - want).abs().max().item() < 6e-2           # another rounding plan of the same network
    finally:
        same.engine.close(); other.engine.close()
