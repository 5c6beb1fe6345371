"""Drop-in legs that round 1 never executed, on the MI355X through the C ABI:
the genuine OpenAI-clip checkpoint layout, the ``PLIP(model_name=<local dir>)`` constructor with its tokenizer, and
string captions end to end."""
import argparse

import numpy as np
import pytest
import torch

from tests import helpers as Hh

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype,tol", [("fp32", 2e-5), ("bf16", 4e-3), ("f16", 6e-4)])
def test_factory_on_a_genuine_openai_layout_checkpoint(dtype, tol, tmp_path, monkeypatch):
    """reproducibility/embedders/factory.py:21-25: ``clip.load(arch)`` + ``load_state_dict(torch.load(backbone))``.
    The checkpoint is ``state_dict()`` of oracle/openai_clip_ref.py (torch.nn.MultiheadAttention: packed
    ``attn.in_proj_weight``, ``visual.proj [width, proj]`` used as ``x @ proj``) -- NOT the converter's own inverse --
    and the expected embeddings are that module's own forward."""
    from oracle import openai_clip_ref as R
    from plip_amd import weights as W
    from plip_amd.config import get_config
    from plip_amd.reproducibility import EmbedderFactory
    cfg = get_config("tiny")
    ref = R.build_random(cfg, seed=11)
    ck = tmp_path / "openai_layout.pt"
    torch.save(ref.state_dict(), ck)                                    # what `torch.save(model.state_dict())` writes
    monkeypatch.setenv("PC_CLIP_ARCH", "tiny")
    emb = EmbedderFactory().factory(argparse.Namespace(model_name="plip", backbone=str(ck), dtype=dtype, max_batch=8))
    try:
        rs = np.random.RandomState(12)
        px = rs.standard_normal((6, 3, cfg.image_size, cfg.image_size)).astype(np.float32)
        ids, _ = W.synthetic_ids(cfg, 6, seed=13, pad="zero")            # clip.tokenize: 0-padded, EOT = highest id
        with torch.no_grad():
            wi = ref.encode_image(torch.from_numpy(px))
            wt = ref.encode_text(torch.from_numpy(ids))
            wi, wt = (wi / wi.norm(dim=1, keepdim=True)).numpy(), (wt / wt.norm(dim=1, keepdim=True)).numpy()
            lpi, _ = ref(torch.from_numpy(px), torch.from_numpy(ids))
        gi, gt = emb.embed_images(px, batch_size=4), emb.embed_text(ids, batch_size=4)
        assert np.abs(gi - wi).max() < tol and np.abs(gt - wt).max() < tol, (np.abs(gi - wi).max(), np.abs(gt - wt).max())
        # OpenAI call surface: model(images, tokens) -> (logits_per_image, logits_per_text)
        got_lpi, got_lpt = emb.model(torch.from_numpy(px), torch.from_numpy(ids))
        scale = float(ref.logit_scale.exp())
        assert np.abs(got_lpi.cpu().numpy() - lpi.numpy()).max() / scale < tol
        assert torch.equal(got_lpi, got_lpt.T.contiguous())
    finally:
        emb.model.engine.close()


def test_plip_constructor_from_local_dir_with_tokenizer_and_strings(tmp_path):
    """plip.py:14-29,55-71,89-103 end to end: ``PLIP(<local HF dir>)`` loads config.json + model.safetensors AND the
    tokenizer (vocab.json / merges.txt, here the synthetic CLIP-format vocabulary), ``encode_text`` takes STRINGS, pads
    to the context length, and ``zero_shot_classification`` returns label strings."""
    from oracle import clip_oracle as O
    from plip_amd import weights as W
    from plip_amd.config import get_config
    from plip_amd.plip import PLIP
    from plip_amd.preprocess import load_tokenizer, preprocess_images
    cfg = get_config("tiny")                                            # vocab 512, bos 510, eos 511, context 16
    sd = W.synthetic_state_dict(cfg, 4)
    d = Hh.write_hf_model_dir(tmp_path / "plip_local", cfg, sd, with_tokenizer=True)
    plip = PLIP(d, dtype="fp32", max_batch=8)
    try:
        assert plip.tokenizer is not None and plip.model.config == cfg
        caps = ["an image of the tumor cell", "the cell", "tumor", "an image of an image of an image of the tumor cell the"]
        got = plip.encode_text(caps, batch_size=3)
        ids, mask = load_tokenizer(d)(caps, cfg.context_length)
        assert ids.shape == (4, 16) and (ids[:, 0] == 510).all() and (ids[3, -1] == 511)      # truncated row keeps EOS
        want = O.text_tower(ids, sd, cfg, mask)
        assert got.shape == (4, cfg.projection_dim) and np.abs(got - want).max() < 2e-4
        rs = np.random.RandomState(5)
        tiles = [rs.randint(0, 256, (cfg.image_size, cfg.image_size, 3), dtype=np.uint8) for _ in range(5)]
        pred = plip.zero_shot_classification(tiles, caps[:3])
        img = O.vision_tower(preprocess_images(tiles, cfg.image_size), sd, cfg)
        sim = O.plip_cosine_similarity(img, want[:3])
        assert pred == [caps[i] for i in sim.argmax(1)]
        # retrieval with top_k beyond the corpus size: every image, best first (the reference's argsort()[:, -k:])
        plip.index_images(tiles, batch_size=4)
        nn = plip.retrieval(caps[:2], top_k=10)
        assert nn.shape == (2, 5)
        np.testing.assert_array_equal(nn[:, 0], O.plip_cosine_similarity(want[:2], img).argmax(1))
    finally:
        plip.model.engine.close()
    # a .pt checkpoint has no tokenizer next to it: the engine still builds, strings are refused, ids work
    ck = tmp_path / "state.pt"
    torch.save({k: torch.from_numpy(np.asarray(v)) for k, v in W.to_openai_state_dict(sd, cfg).items()}, ck)
    from plip_amd.model import PlipModel
    p2 = PLIP(model=PlipModel.from_pretrained(str(ck), dtype="fp32", max_batch=4))
    try:
        assert p2.tokenizer is None
        with pytest.raises(RuntimeError, match="tokenizer"):
            p2.encode_text(["tumor"], batch_size=1)
        assert p2.encode_text(ids, batch_size=2).shape == (4, cfg.projection_dim)
    finally:
        p2.model.engine.close()
    with pytest.raises(FileNotFoundError):
        PLIP(d, tokenizer_dir=str(tmp_path / "nowhere"))               # fails BEFORE an engine is created


def test_non_square_images_follow_the_hf_crop_rule(engines):
    """PLIP.encode_images on 227 x 224 / 224 x 231 images == HF CLIPImageProcessor pixels through the tower, on both
    the GPU resize path (equal sizes) and the host Pillow path (mixed sizes)."""
    pytest.importorskip("transformers")
    from transformers import CLIPImageProcessor
    from oracle import clip_oracle as O
    from PIL import Image
    from plip_amd.config import get_config
    from plip_amd import weights as W
    from plip_amd.model import PlipModel
    from plip_amd.plip import PLIP
    cfg = get_config("tiny").replace(image_size=224, patch_size=32)
    sd = W.synthetic_state_dict(cfg, 9)
    model = PlipModel(cfg, sd, dtype="fp32", max_batch=8)
    try:
        plip = PLIP(model=model)
        proc = CLIPImageProcessor()
        rs = np.random.RandomState(1)
        same = [Image.fromarray(rs.randint(0, 256, (224, 227, 3), dtype=np.uint8)) for _ in range(3)]    # h=224, w=227
        mixed = same[:1] + [Image.fromarray(rs.randint(0, 256, (231, 224, 3), dtype=np.uint8))]
        for imgs in (same, mixed):
            want = O.vision_tower(proc(images=imgs, return_tensors="np")["pixel_values"], sd, cfg)
            got = plip.encode_images(imgs, batch_size=4)
            assert np.abs(got - want).max() < 2e-4
    finally:
        model.engine.close()
