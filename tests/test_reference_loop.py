"""The reference's OWN host loops driven against the drop-in (SURVEY.md section 4 item 4, section 8b surface B1).

/root/reference/plip.py:31-71 (``encode_images`` / ``encode_text``: ``datasets.Dataset`` -> ``DataLoader`` ->
``self.preprocess(...)`` -> ``self.model.get_*_features(**batch).detach().cpu().numpy()``) is imported by path and run
unmodified via ``object.__new__(PLIP)`` with ``self.model`` := ``plip_amd.model.PlipModel`` and ``self.preprocess`` :=
``CLIPProcessor(CLIPImageProcessor(), CLIPTokenizer(<synthetic vocab>))`` (both construct offline).
``PLIP.__init__`` itself cannot run under transformers 5.x (``use_auth_token`` kwarg, plip.py:26).

* CPU variant (runs in the build container): the model is a real ``PlipModel`` whose engine is the oracle-backed
  stand-in of tests/helpers.py, so what is under test is the HOST surface -- kwargs by name, ``.to(device)``, tensor
  results, batch loop -- and ``plip_amd.PLIP``'s own loops against the reference's on identical inputs, including the
  centre-crop rule on a 227 x 224 image.
* GPU variant: the GPU box does not mount /root/reference, so there the loop is ``tests.helpers.ReferenceHostLoops`` -- a
  restatement of plip.py:31-103's host pattern which the CPU variant pins to the imported original (same arrays on the
  same model) -- driven against the MI355X engine through libplipmi.so, in all three compute dtypes.  Where the reference
  tree IS mounted next to a GPU, the original itself runs too.
The reference's ``reproducibility`` ``CLIPEmbedder`` cannot be driven this way at all: it imports the OpenAI ``clip``
package, which is not installed.
"""
import numpy as np
import pytest
import torch

from tests import helpers as Hh

pytest.importorskip("datasets")
needs_reference = pytest.mark.skipif(not Hh.reference_available(), reason="/root/reference is not mounted on this box")


def _cfg():
    from plip_amd.config import get_config
    # what the reference hard-codes: 224-pixel processor output, max_length=77 (plip.py:58)
    return get_config("tiny").replace(image_size=224, patch_size=32, context_length=77)


def _images():
    from PIL import Image
    rs = np.random.RandomState(7)
    sizes = [(224, 224), (227, 224), (224, 231), (300, 225), (224, 224)]      # (w, h); 227 / 231 hit the crop-rule seam
    return [Image.fromarray(rs.randint(0, 256, (h, w, 3), dtype=np.uint8)) for (w, h) in sizes]


def _processor(tmp_path):
    from transformers import CLIPImageProcessor, CLIPProcessor, CLIPTokenizer
    Hh.write_tokenizer_fixture(str(tmp_path), 510, 511)
    tok = CLIPTokenizer.from_pretrained(str(tmp_path))
    return CLIPProcessor(image_processor=CLIPImageProcessor(), tokenizer=tok), tok


CAPTIONS = ["an image of the tumor cell", "the cell", "tumor", "an image of an image of the tumor"]


def _reference_instance(model, processor, device):
    ref_mod = Hh.load_reference_plip_module()
    ref = object.__new__(ref_mod.PLIP)           # plip.py:14-18 without the hub download
    ref.device = device
    ref.model_name = "local"
    ref.model, ref.preprocess, ref.model_hash = model, processor, hash
    ref.model = ref.model.to(ref.device)         # plip.py:18
    return ref


@needs_reference
def test_reference_host_loops_on_the_drop_in_model_cpu(tmp_path):
    from oracle import clip_oracle as O
    from plip_amd import weights as W
    from plip_amd.plip import PLIP
    from plip_amd.preprocess import load_tokenizer
    cfg = _cfg()
    sd = W.synthetic_state_dict(cfg, 2)
    model = Hh.oracle_model(cfg, sd)
    processor, tok = _processor(tmp_path)
    ref = _reference_instance(model, processor, "cpu")
    images = _images()

    # --- the reference's loops, unmodified, on the drop-in model --------------------------------------------------
    got_img = ref.encode_images(images, batch_size=2)                       # plip.py:31-53
    got_txt = ref.encode_text(CAPTIONS, batch_size=3)                       # plip.py:55-71
    assert got_img.shape == (5, cfg.projection_dim) and got_img.dtype == np.float32
    assert got_txt.shape == (4, cfg.projection_dim) and got_txt.dtype == np.float32
    # what those loops must have computed: HF preprocessing -> the towers, un-normalised
    px = processor(images=images, return_tensors="np")["pixel_values"]
    enc = tok(CAPTIONS, return_tensors="np", max_length=77, padding="max_length", truncation=True)
    np.testing.assert_allclose(got_img, O.vision_tower(px, sd, cfg), rtol=0, atol=1e-6)
    np.testing.assert_allclose(got_txt, O.text_tower(enc["input_ids"], sd, cfg, enc["attention_mask"]), rtol=0, atol=1e-6)
    labels = CAPTIONS[:3]
    ref_pred = ref.zero_shot_classification(images, labels)                 # plip.py:89-103
    assert len(ref_pred) == 5 and set(ref_pred) <= set(labels)

    # --- plip_amd.PLIP's loops on the same model + inputs: identical arrays, identical predictions -------------------
    ours = object.__new__(PLIP)                  # the constructor insists on a GPU; the host loops do not need one
    ours.device, ours.model_name, ours.model = "cpu", "local", model
    ours.tokenizer = load_tokenizer(str(tmp_path))
    ours.model_hash, ours.image_vectors = hash, None
    model.engine.calls.clear()
    mine_img = ours.encode_images(images, batch_size=2)
    mine_txt = ours.encode_text(CAPTIONS, batch_size=3)
    np.testing.assert_allclose(mine_img, got_img, rtol=0, atol=2e-6)        # incl. the 227 x 224 / 224 x 231 crops
    np.testing.assert_allclose(mine_txt, got_txt, rtol=0, atol=1e-6)
    assert ours.zero_shot_classification(images, labels) == ref_pred
    kinds = {c[0] for c in model.engine.calls}
    assert "encode_image_u8" in kinds            # native 224 x 224 tiles took the fused-normalisation route
    np.testing.assert_allclose(ours._cosine_similarity(mine_img, mine_txt), ref._cosine_similarity(got_img, got_txt),
                                rtol=0, atol=1e-5)
    ours.image_vectors, ref.image_vectors = mine_img, got_img               # (the reference never assigns it: plip.py:114)
    np.testing.assert_array_equal(ours.retrieval(CAPTIONS, top_k=3), ref.retrieval(CAPTIONS, top_k=3))
    # k beyond the corpus: the reference's argsort()[:, -k:] hands back every column
    assert ours.retrieval(CAPTIONS, top_k=10).shape == ref.retrieval(CAPTIONS, top_k=10).shape == (4, 5)
    # ... and for k = 0 too ([:, -0:] is [:, 0:]): ADVICE r2
    np.testing.assert_array_equal(ours.retrieval(CAPTIONS, top_k=0), ref.retrieval(CAPTIONS, top_k=0))
    # ... and k < 0 drops the |k| weakest columns ([:, -k:] == [:, |k|:] of the ascending order): ADVICE r3
    np.testing.assert_array_equal(ours.retrieval(CAPTIONS, top_k=-2), ref.retrieval(CAPTIONS, top_k=-2))
    assert ours.retrieval(CAPTIONS, top_k=-2).shape == (4, 3) and ours.retrieval(CAPTIONS, top_k=-9).shape == (4, 0)

    # --- batch_size no longer sizes the ENGINE calls: the caller's batches of 2 / 3 reach the towers
    #     max_batch rows at a time, same arrays; coalesce = False is one engine call per caller batch --------------------
    def calls_of(fn):
        model.engine.calls.clear()
        out = fn()
        return out, [c for c in model.engine.calls if c[0].startswith("encode")]
    u8 = lambda calls: [c for c in calls if c[0] != "encode_image"]      # (the stand-in's u8 route logs its inner call too)
    tiles = [np.asarray(im.convert("RGB").resize((cfg.image_size, cfg.image_size))) for im in images] * 4      # 20 native tiles
    ids = np.asarray(tok(CAPTIONS * 5, return_tensors="np", max_length=77, padding="max_length", truncation=True)["input_ids"])
    assert model.engine.max_batch == 8
    a_img, c_img = calls_of(lambda: ours.encode_images(tiles, batch_size=2))
    a_txt, c_txt = calls_of(lambda: ours.encode_text(ids, batch_size=3))
    assert [c[1][0] for c in u8(c_img)] == [8, 8, 4] and [c[1][0] for c in c_txt] == [8, 8, 4]
    ours.coalesce = False
    b_img, d_img = calls_of(lambda: ours.encode_images(tiles, batch_size=2))
    b_txt, d_txt = calls_of(lambda: ours.encode_text(ids, batch_size=3))
    del ours.coalesce
    assert [c[1][0] for c in u8(d_img)] == [2] * 10 and [c[1][0] for c in d_txt] == [3] * 6 + [2]
    np.testing.assert_allclose(a_img, b_img, rtol=0, atol=2e-6)
    np.testing.assert_allclose(a_txt, b_txt, rtol=0, atol=2e-6)
    # mixed routes keep their per-batch routing and their order: 2 native tiles, 2 odd-sized images, 2 native tiles
    mixed = tiles[:2] + images[:2] + tiles[2:4]
    m_img, c_mix = calls_of(lambda: ours.encode_images(mixed, batch_size=2))
    assert [c[0] for c in c_mix] == ["encode_image_u8", "encode_image", "encode_image", "encode_image_u8", "encode_image"]
    np.testing.assert_allclose(m_img[:2], a_img[:2], rtol=0, atol=2e-6)
    np.testing.assert_allclose(m_img[4:6], a_img[2:4], rtol=0, atol=2e-6)

    # --- the restatement the GPU box runs (no /root/reference there) IS the reference's loop: identical arrays -------------
    restated = Hh.ReferenceHostLoops(model, processor, "cpu")
    np.testing.assert_array_equal(restated.encode_images(images, batch_size=2), got_img)
    np.testing.assert_array_equal(restated.encode_text(CAPTIONS, batch_size=3), got_txt)
    np.testing.assert_array_equal(restated.cosine_similarity(got_img, got_txt), ref._cosine_similarity(got_img, got_txt))
    assert restated.zero_shot_classification(images, labels) == ref_pred


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [("f32", 2e-4), ("bf16", 6e-2), ("f16", 1.5e-2)])
def test_reference_host_loops_on_the_mi355x_engine(dtype, tol, tmp_path):
    """Surface B1 through libplipmi.so: the reference's host loop (restated; the original too where the tree is mounted)
    feeds ``PlipModel`` on the GPU -- ``**batch`` kwargs, device tensors in, ``.detach().cpu().numpy()`` out."""
    from oracle import clip_oracle as O
    from plip_amd import weights as W
    from plip_amd.model import PlipModel
    cfg = _cfg()
    sd = W.synthetic_state_dict(cfg, 2)
    model = PlipModel(cfg, sd, dtype=dtype, max_batch=8)
    try:
        processor, tok = _processor(tmp_path)
        images = _images()
        px = processor(images=images, return_tensors="np")["pixel_values"]
        enc = tok(CAPTIONS, return_tensors="np", max_length=77, padding="max_length", truncation=True)
        want_img, want_txt = O.vision_tower(px, sd, cfg), O.text_tower(enc["input_ids"], sd, cfg, enc["attention_mask"])
        loops = [Hh.ReferenceHostLoops(model, processor, "cuda")]
        if Hh.reference_available():
            loops.append(_reference_instance(model, processor, "cuda"))
        outs = []
        for loop in loops:
            got_img = loop.encode_images(images, batch_size=2)
            got_txt = loop.encode_text(CAPTIONS, batch_size=3)
            assert got_img.shape == (5, cfg.projection_dim) and got_img.dtype == np.float32
            assert np.abs(got_img - want_img).max() < tol and np.abs(got_txt - want_txt).max() < tol
            pred = loop.zero_shot_classification(images, CAPTIONS[:3])
            assert len(pred) == 5 and set(pred) <= set(CAPTIONS[:3])
            outs.append((got_img, got_txt, pred))
        if len(outs) == 2:                       # original and restatement drive the engine identically
            np.testing.assert_array_equal(outs[0][0], outs[1][0])
            np.testing.assert_array_equal(outs[0][1], outs[1][1])
            assert outs[0][2] == outs[1][2]
    finally:
        model.engine.close()
