"""CPU-side checks: the C-ABI library builds/loads and exports every symbol the header
declares, checkpoint ingestion (HF and OpenAI-clip formats), preprocessing, config maths.
No kernel is launched here (no GPU in this container)."""
import os
import re

import numpy as np
import pytest

from plip_amd import weights as W
from plip_amd.config import PRESETS, get_config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from plip_amd import _lib
    from plip_amd.build import build
    build(verbose=False)
    lib = _lib.load()
    # the product header (the drop-in boundary) and the test header (kernel-level entries, A/B hooks): each lists exactly
    # what the binding binds from it, and the library exports all of it
    for hdr, table in (("plipmi.h", _lib.SYMBOLS), ("plipmi_test.h", _lib.TEST_SYMBOLS)):
        header = open(os.path.join(ROOT, "include", hdr)).read()
        declared = set(re.findall(r"\b(plipmi_[a-z0-9_]+)\s*\(", header))
        assert declared == set(table), (hdr, declared ^ set(table))
        for name in declared:
            assert hasattr(lib, name), name
    assert not set(_lib.SYMBOLS) & set(_lib.TEST_SYMBOLS)
    # nothing test-only in the product header: no kernel-level GEMM / attention entry, no process-wide switch
    product = open(os.path.join(ROOT, "include", "plipmi.h")).read()
    for hook in ("plipmi_gemm_nt", "plipmi_attention", "plipmi_test_force_gemm_tile", "plipmi_test_reset_hooks", "plipmi_recode_planes", "plipmi_debug_hidden"):
        assert not re.search(r"\b%s\s*\(" % hook, product), hook
    assert lib.plipmi_version() == 412
    names = []
    i = 0
    while lib.plipmi_gemm_variant_name(i):
        names.append(lib.plipmi_gemm_variant_name(i).decode())
        i += 1
    assert len(names) >= 2 and lib.plipmi_gemm_variant_name(-1) is None


def test_product_modules_do_not_call_the_test_header():
    """include/plipmi_test.h is for tests/ and tools/: the only module of the package that binds its kernel-level entries is
    plip_amd/kernel_entries.py (imported by tests and tools, never by the product path); Engine.hidden -- the parity tests'
    window on the hidden states -- is the one test hook a product class carries."""
    from plip_amd import _lib
    pkg = os.path.join(ROOT, "plip_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if not f.endswith(".py") or f in ("kernel_entries.py", "_lib.py"):
                continue
            src = open(os.path.join(dirpath, f)).read()
            assert "kernel_entries" not in src, f
            for name in _lib.TEST_SYMBOLS:
                if name == "plipmi_debug_hidden" and f == "engine.py":
                    continue
                assert name not in src, (f, name)
    for f in ("bench.py", "__graft_entry__.py"):
        assert "kernel_entries" not in open(os.path.join(ROOT, f)).read(), f


def test_library_reads_no_environment_variables():
    """Every switch of the library is an argument (plipmi_config.flags, per-handle setters, the test hooks): the shared
    object does not even import getenv."""
    import subprocess
    from plip_amd.build import LIB as LIB_PATH, build
    build(verbose=False)
    syms = subprocess.run(["nm", "-D", "--undefined-only", LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in syms
    for f in os.listdir(os.path.join(ROOT, "plip_amd", "csrc")):
        if f.endswith((".hip", ".h")):
            assert "getenv" not in open(os.path.join(ROOT, "plip_amd", "csrc", f)).read(), f


def test_struct_layouts_match_header():
    """ctypes mirrors of the C structs: field order/sizes as in include/plipmi.h."""
    import ctypes as C

    from plip_amd import _lib
    assert C.sizeof(_lib.Config) == 21 * 4
    assert _lib.Config._fields_[0][0] == "struct_size"      # plipmi_create reads that many bytes of the caller's struct
    assert C.sizeof(_lib.LayerWeights) == 16 * 8
    assert C.sizeof(_lib.Weights) == 15 * 8
    assert C.sizeof(_lib.KernelStat) == 96 + 8 + 3 * 8
    header = open(os.path.join(ROOT, "include", "plipmi.h")).read()
    cfg_block = header[header.index("typedef struct plipmi_config {"):header.index("} plipmi_config;")]
    fields = re.findall(r"^\s*(?:int32_t|float)\s+(\w+);", cfg_block, flags=re.M)
    assert fields == [f[0] for f in _lib.Config._fields_]


def test_engine_creation_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from plip_amd.model import PlipModel
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        PlipModel.from_synthetic("tiny")


def test_flop_model_matches_survey():
    c = get_config("ViT-B/32")
    assert abs(c.image_flops() / 1e9 - 8.8176) < 1e-3
    assert abs(c.text_flops() / 1e9 - 5.9595) < 1e-3
    assert abs(c.pair_flops() / 1e9 - 14.7772) < 1e-3
    assert abs(get_config("ViT-L/14@336px").image_flops() / 1e9 - 381.92) < 1e-1
    for name, cfg in PRESETS.items():
        cfg.validate()


def test_openai_checkpoint_round_trip():
    cfg = get_config("tiny")
    sd = W.synthetic_state_dict(cfg, 3)
    oa = W.to_openai_state_dict(sd, cfg)
    assert W.is_openai_state_dict(oa) and not W.is_openai_state_dict(sd)
    rcfg = W.config_from_openai_state_dict(oa)
    for f in ("image_size", "patch_size", "v_width", "v_layers", "v_heads", "v_mlp", "vocab_size",
              "context_length", "t_width", "t_layers", "t_heads", "t_mlp", "projection_dim"):
        assert getattr(rcfg, f) == getattr(cfg, f), f
    assert rcfg.eos_token_id == 2          # OpenAI pools at argmax(ids)
    back, _ = W.normalize_state_dict(oa)
    assert set(back) == set(sd)
    for k in sd:
        np.testing.assert_array_equal(back[k], sd[k])


def test_state_dict_validation():
    cfg = get_config("tiny")
    sd = W.synthetic_state_dict(cfg, 0)
    W.check_state_dict(sd, cfg)
    bad = dict(sd)
    bad.pop("visual_projection.weight")
    with pytest.raises(KeyError):
        W.check_state_dict(bad, cfg)
    bad = dict(sd)
    bad["text_projection.weight"] = bad["text_projection.weight"].T
    with pytest.raises(ValueError):
        W.check_state_dict(bad, cfg.replace(projection_dim=32))


def test_hf_checkpoint_dir_round_trip(tmp_path):
    """load_checkpoint on an HF-layout directory (config.json + model.safetensors)."""
    pytest.importorskip("safetensors")
    import json

    from safetensors.numpy import save_file
    cfg = get_config("tiny")
    sd = W.synthetic_state_dict(cfg, 1)
    save_file({k: np.ascontiguousarray(np.asarray(v, dtype=np.float32)) for k, v in sd.items()},
              str(tmp_path / "model.safetensors"))
    hf_cfg = {"projection_dim": cfg.projection_dim, "logit_scale_init_value": 2.6592,
              "text_config": {"vocab_size": cfg.vocab_size, "hidden_size": cfg.t_width, "intermediate_size": cfg.t_mlp,
                              "num_hidden_layers": cfg.t_layers, "num_attention_heads": cfg.t_heads,
                              "max_position_embeddings": cfg.context_length, "eos_token_id": cfg.eos_token_id,
                              "bos_token_id": cfg.bos_token_id},
              "vision_config": {"hidden_size": cfg.v_width, "intermediate_size": cfg.v_mlp,
                                "num_hidden_layers": cfg.v_layers, "num_attention_heads": cfg.v_heads,
                                "image_size": cfg.image_size, "patch_size": cfg.patch_size, "layer_norm_eps": 1e-5}}
    (tmp_path / "config.json").write_text(json.dumps(hf_cfg))
    got, gcfg = W.load_checkpoint(str(tmp_path))
    assert gcfg == cfg
    for k in sd:
        np.testing.assert_array_equal(got[k], sd[k])


def test_synthetic_ids_are_tokenizer_shaped():
    cfg = get_config("ViT-B/32")
    ids, mask = W.synthetic_ids(cfg, 64, seed=5)
    assert ids.shape == (64, 77) and ids.dtype == np.int64
    assert (ids[:, 0] == 49406).all() and ids.max() == 49407 and ids.min() >= 1
    first_eos = (ids == 49407).argmax(1)
    np.testing.assert_array_equal(mask.sum(1), first_eos + 1)
    idz, _ = W.synthetic_ids(cfg, 8, seed=5, pad="zero")
    assert ((idz == 0).sum(1) > 0).all() and ((idz == 49407).sum(1) == 1).all()


def test_preprocess_reduces_to_affine_on_native_tiles():
    """transform.py:45-52 / CLIPImageProcessor on a tile that is already n_px x n_px."""
    from plip_amd.preprocess import CLIP_MEAN, CLIP_STD, preprocess_image
    rs = np.random.RandomState(0)
    tile = rs.randint(0, 256, size=(224, 224, 3), dtype=np.uint8)
    got = preprocess_image(tile, 224)
    want = ((tile.astype(np.float32) / 255.0 - np.float32(CLIP_MEAN)) / np.float32(CLIP_STD)).transpose(2, 0, 1)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-6)
    big = rs.randint(0, 256, size=(300, 260, 3), dtype=np.uint8)
    assert preprocess_image(big, 224).shape == (3, 224, 224)


def test_preprocess_matches_hf_image_processor():
    tr = pytest.importorskip("transformers")
    from PIL import Image

    from plip_amd.preprocess import preprocess_image
    rs = np.random.RandomState(1)
    tile = rs.randint(0, 256, size=(224, 224, 3), dtype=np.uint8)
    proc = tr.CLIPImageProcessor()
    want = proc(images=Image.fromarray(tile), return_tensors="np")["pixel_values"][0]
    np.testing.assert_allclose(preprocess_image(tile), want, rtol=0, atol=1e-6)


def test_host_pipeline_order_and_errors():
    """plip_amd.pipeline.run_batches: batches arrive in order whatever the decode timing; decode errors propagate."""
    import time
    import torch
    from plip_amd.pipeline import run_batches
    items = list(range(23))

    def prep(i):
        time.sleep(0.001 * (i % 3))
        return np.full((2, 2), i, np.float32)

    for workers in (0, 1, 4):
        outs = run_batches(items, 5, prep, lambda t: t.sum(dim=(1, 2)), device=None, num_workers=workers)
        assert [o.shape[0] for o in outs] == [5, 5, 5, 5, 3]
        assert torch.cat(outs).tolist() == [4.0 * i for i in items]
    assert run_batches([], 5, prep, lambda t: t) == []

    def bad(i):
        if i == 7:
            raise OSError("truncated image")
        return prep(i)

    with pytest.raises(OSError):
        run_batches(items, 5, bad, lambda t: t, device=None, num_workers=2)

    # lanes (Engine.lanes(): consecutive batches alternate between an engine and its clone): batch k is consumed with lane k % n's
    # engine as the last argument, the order of the outputs is the order of the batches -- with and without a batch tag
    seen = []
    outs = run_batches(items, 5, prep, lambda t, e: (seen.append(e), t.sum(dim=(1, 2)))[1], device=None, num_workers=2,
                       lanes=[("engine", None), ("clone", None)])
    assert seen == ["engine", "clone", "engine", "clone", "engine"] and torch.cat(outs).tolist() == [4.0 * i for i in items]
    seen = []
    outs = run_batches(items, 5, None, lambda tag, t, e: (seen.append((tag, e)), t.sum(dim=(1, 2)))[1], device=None, num_workers=0,
                       prepare_batch=lambda chunk, pool: ("tiles", np.stack([prep(i) for i in chunk])), lanes=[("engine", None), ("clone", None)])
    assert seen == [("tiles", "engine"), ("tiles", "clone")] * 2 + [("tiles", "engine")] and torch.cat(outs).tolist() == [4.0 * i for i in items]


@pytest.mark.parametrize("h,w,n", [(300, 500, 224), (512, 512, 224), (256, 256, 224), (224, 300, 224), (1000, 700, 224),
                                   (100, 130, 64), (64, 200, 64), (233, 224, 224), (225, 225, 224), (96, 96, 224)])
def test_resample_tables_reproduce_pillow_bit_for_bit(h, w, n):
    """The fixed-point tables handed to plipmi_resize_crop_u8 + the two integer passes (numpy emulation of the
    kernels) == Image.resize(BICUBIC) + centre crop, for down- and up-scaling, one or both axes."""
    from PIL import Image
    from plip_amd.preprocess import preprocess_image, resize_crop_plan, resize_crop_reference
    img = np.random.RandomState(h * 7 + w).randint(0, 256, (h, w, 3), dtype=np.uint8)
    plan = resize_crop_plan(w, h, n)
    got = resize_crop_reference(img, plan)
    im = Image.fromarray(img).resize((plan["nw"], plan["nh"]), resample=Image.BICUBIC)
    want = np.asarray(im.crop((plan["left"], plan["top"], plan["left"] + n, plan["top"] + n)))
    np.testing.assert_array_equal(got, want)
    # and it is the geometry of the host preprocessing path (which is checked against HF's CLIPImageProcessor above)
    px = preprocess_image(img, n)
    from plip_amd.preprocess import CLIP_MEAN, CLIP_STD
    ref = ((got.astype(np.float32) / np.float32(255.0) - np.asarray(CLIP_MEAN, np.float32)) / np.asarray(CLIP_STD, np.float32))
    np.testing.assert_array_equal(px, ref.transpose(2, 0, 1))


def test_resize_plan_rejects_images_that_cannot_be_cropped():
    from plip_amd.preprocess import resize_crop_plan
    plan = resize_crop_plan(640, 480, 224)
    assert (plan["nw"], plan["nh"], plan["left"], plan["top"]) == (298, 224, 37, 0) and plan["yb"].shape == (224, 2)
    assert resize_crop_plan(224, 224, 224)["xb"] is None and resize_crop_plan(224, 224, 224)["yb"] is None


def test_crop_rules_match_their_front_ends():
    """The two preprocessing front ends of the reference disagree by one pixel on odd excesses:
    HF CLIPImageProcessor (top-level PLIP, plip.py:27,35) crops at (extent - n) // 2, torchvision CenterCrop (OpenAI
    _transform, reproducibility/embedders/transform.py:47) at int(round((extent - n) / 2.0))."""
    from PIL import Image
    from plip_amd.preprocess import crop_offset, preprocess_image, resize_crop_plan, resize_crop_reference
    assert [crop_offset(e, 224, "hf") for e in (224, 225, 227, 231, 298)] == [0, 0, 1, 3, 37]
    assert [crop_offset(e, 224, "torchvision") for e in (224, 225, 227, 231, 298)] == [0, 0, 2, 4, 37]
    with pytest.raises(ValueError):
        crop_offset(230, 224, "middle")
    transformers = pytest.importorskip("transformers")
    proc = transformers.CLIPImageProcessor()
    rs = np.random.RandomState(3)
    for (w, h) in [(224, 224), (227, 224), (231, 224), (224, 231), (454, 448), (300, 225), (500, 375)]:
        im = Image.fromarray(rs.randint(0, 256, (h, w, 3), dtype=np.uint8))
        want = proc(images=im, return_tensors="np")["pixel_values"][0]
        np.testing.assert_allclose(preprocess_image(im, 224, crop="hf"), want, rtol=0, atol=1e-6, err_msg=str((w, h)))
        # the GPU resize path's integer plan, emulated: same pixels as Pillow under the same rule
        plan = resize_crop_plan(w, h, 224, crop="hf")
        u8 = resize_crop_reference(np.asarray(im), plan)
        x = (u8.astype(np.float32) / np.float32(255.0) - np.asarray([0.48145466, 0.4578275, 0.40821073], np.float32)) / \
            np.asarray([0.26862954, 0.26130258, 0.27577711], np.float32)
        np.testing.assert_allclose(x.transpose(2, 0, 1), want, rtol=0, atol=1e-6)
    odd = Image.fromarray(rs.randint(0, 256, (224, 227, 3), dtype=np.uint8))
    assert np.abs(preprocess_image(odd, 224, crop="hf") - preprocess_image(odd, 224, crop="torchvision")).max() > 0.1


def test_tokenizer_contract_on_synthetic_vocab(tmp_path):
    """plip.py:56-60: ``self.preprocess(text=..., max_length=77, padding="max_length", truncation=True)``.  No CLIP
    vocabulary is on disk, so a ~100-entry vocab.json / merges.txt in CLIP's format stands in (tests/helpers.py);
    ``load_tokenizer`` must give BOS first, EOS right after the last token, EOS-padding to the context length,
    truncation that keeps the EOS, and the matching attention mask -- what ``plipmi_encode_text`` pools on."""
    pytest.importorskip("transformers")
    from plip_amd.preprocess import load_tokenizer
    from tests.helpers import write_tokenizer_fixture
    write_tokenizer_fixture(str(tmp_path), 510, 511)
    tok = load_tokenizer(str(tmp_path))
    ids, mask = tok(["An image of the TUMOR cell", "a", "the " * 40], 16)
    assert ids.shape == (3, 16) and ids.dtype == np.int64 and mask.dtype == np.int64
    assert (ids[:, 0] == 510).all()
    first_eos = (ids == 511).argmax(1)
    np.testing.assert_array_equal(first_eos, [7, 2, 15])           # 6 word tokens, 1 token, truncated to 14 + BOS + EOS
    np.testing.assert_array_equal(mask.sum(1), first_eos + 1)
    for r in range(3):
        assert (ids[r, first_eos[r]:] == 511).all()                # pad token == EOS
    a, _ = tok(["an image of the tumor cell"], 16)
    np.testing.assert_array_equal(a[0], ids[0])                    # lower-cased
    ids77, _ = tok(["tumor"], 77)
    assert ids77.shape == (1, 77)


def test_load_checkpoint_refuses_code_carrying_pickles(tmp_path, monkeypatch):
    """``.pt`` files load with weights_only=True; a pickle that needs code execution is refused unless trusted."""
    import torch
    cfg = get_config("tiny")
    sd = W.synthetic_state_dict(cfg, 1)
    plain = tmp_path / "plain.pt"
    torch.save({k: torch.from_numpy(np.asarray(v)) for k, v in W.to_openai_state_dict(sd, cfg).items()}, plain)
    got, gcfg = W.load_checkpoint(str(plain))
    np.testing.assert_array_equal(got["visual_projection.weight"], sd["visual_projection.weight"])

    class Holder:                       # a whole pickled object, as `torch.save(model)` produces
        def __init__(self, d):
            self.d = d

        def state_dict(self):
            return self.d
    import tests.test_host as me        # picklable by reference
    me.Holder = Holder
    Holder.__module__, Holder.__qualname__ = "tests.test_host", "Holder"
    whole = tmp_path / "whole.pt"
    torch.save(Holder({k: torch.from_numpy(np.asarray(v)) for k, v in W.to_openai_state_dict(sd, cfg).items()}), whole)
    monkeypatch.delenv("PLIPMI_TRUST_PICKLE", raising=False)
    with pytest.raises(RuntimeError, match="trust_pickle"):
        W.load_checkpoint(str(whole))
    got2, _ = W.load_checkpoint(str(whole), trust_pickle=True)
    np.testing.assert_array_equal(got2["visual_projection.weight"], sd["visual_projection.weight"])
    # a DAMAGED file is reported as damaged -- no nudge towards arbitrary-code unpickling
    cut = tmp_path / "cut.pt"
    cut.write_bytes(plain.read_bytes()[: plain.stat().st_size // 2])
    with pytest.raises(RuntimeError) as ei:
        W.load_checkpoint(str(cut))
    assert "trust_pickle" not in str(ei.value) and "truncated or corrupt" in str(ei.value), str(ei.value)
    # ... whichever way torch words it: a ZIP checkpoint cut in its payload (the central directory is gone) and one cut to a stub
    for name, keep in (("tail.pt", plain.stat().st_size - 64), ("stub.pt", 100)):
        f = tmp_path / name
        f.write_bytes(plain.read_bytes()[:keep])
        with pytest.raises(RuntimeError) as ei:
            W.load_checkpoint(str(f))
        assert "trust_pickle" not in str(ei.value) and "truncated or corrupt" in str(ei.value), (name, str(ei.value))


def test_split_plane_host_mirror_follows_the_format():
    """The 16-bit engines keep their residual stream as the operand-type plane hi + an 8-bit remainder plane lo (csrc/common.h
    split_f32, round 6; rounds 2-5 carried a 16-bit remainder = the exact fp32 value): hi is the correctly rounded operand, the pair
    reproduces x to 2^-16 (bf16) / 2^-19 (f16) relative (twice that in the clamp corner), lo sits in the blocked layout the GEMM epilogues address in 16-byte pieces."""
    import torch
    from plip_amd.kernel_entries import join_planes, lo_plane_bytes, lo_plane_index, split_planes
    g = torch.Generator().manual_seed(5)
    M, N = 250, 800
    x = (torch.randn(M * N, generator=g) * torch.exp(torch.randn(M * N, generator=g) * 6)).clamp(-3.0e38, 3.0e38)
    x[:6] = torch.tensor([0.0, -0.0, 1e-38, -3.0e38, 1.00390625, -1.00390625])      # the last two are exact bf16 ties
    x = x.reshape(M, N)
    hi, lo = split_planes(x)
    assert hi.dtype == torch.bfloat16 and lo.dtype == torch.uint8 and lo.numel() == lo_plane_bytes(M, N) == 256 * 800
    assert ((hi.float() - x).abs() <= (x.bfloat16().float() - x).abs()).all()        # hi = the nearest bf16
    assert hi[0, 4].item() == 1.0078125 and hi[0, 5].item() == -1.0078125            # ties go away from zero
    xr = join_planes(hi, lo)
    big = x.abs() > 1e-37
    rel = ((xr - x).abs() / x.abs())[big]                                              # 8 + 8 significand bits: 2^-16, and 2^-15 in the
    assert (rel <= 2.0 ** -15).all() and (rel > 2.0 ** -16).float().mean() < 4e-3    # corner where +128 is clamped to +127 (r >= 32640)
    hi2, lo2 = split_planes(xr)                                                       # a stored value splits into itself
    assert torch.equal(hi2.view(torch.int16), hi.view(torch.int16)) and torch.equal(lo2, lo)
    # layout: a permutation of [0, M * N) into the padded plane; rows r and r + 8 of a band, 8 consecutive columns = 16 contiguous bytes
    idx = lo_plane_index(M, N)
    assert idx.unique().numel() == M * N and int(idx.max()) < lo_plane_bytes(M, N)
    assert idx[35, 16:24].tolist() == list(range(int(idx[35, 16]), int(idx[35, 16]) + 8))
    assert int(idx[43, 16]) == int(idx[35, 16]) + 8 and int(idx[35, 16]) % 16 == 0 and int(idx[35, 24]) == int(idx[35, 16]) + 128
    # f16 engine: hi = nearest f16 (ties to even, saturating), lo = the remainder in units of 2^(E(hi) - 18), 11 + 8 bits
    y = x.clamp(-65504.0, 65504.0)
    y = torch.where(y.abs() < 2.0 ** -14, torch.zeros_like(y), y)
    y[0, :4] = torch.tensor([2.0 ** -14, 65504.0, 1.00048828125, -2047.5])           # smallest normal, largest, a tie, a tie
    hi, lo = split_planes(y, torch.float16)
    assert hi.dtype == torch.float16 and torch.equal(hi, y.half())
    yr = join_planes(hi, lo)
    assert ((yr - y).abs() <= y.abs() * 2.0 ** -18).all() and ((yr - y).abs() > y.abs() * 2.0 ** -19).float().mean() < 4e-3
    z = torch.tensor([[1e5, -3e38, 1e-9, 0.0, 0.0, 0.0, 0.0, 0.0]])                   # beyond the range: saturates / tiny: 2^-32 steps
    hi, lo = split_planes(z, torch.float16)
    assert hi[0, 0].item() == 65504.0 and hi[0, 1].item() == -65504.0
    assert abs(join_planes(hi, lo)[0, 2].item() - 1e-9) < 2.0 ** -32


def test_u8_normalisation_by_one_fma_is_exact_after_rounding():
    """csrc/gemm.h ADDR 3 (im2col on load from uint8 tiles) normalises a byte of channel c as fl(b * A_c + B_c); the unfold kernel --
    and the reference's transform, reproducibility/embedders/transform.py:45-52 -- compute (b / 255 - mean_c) * (1 / std_c) in three fp32
    roundings.  The two agree after rounding to the operand type for EVERY byte value and channel, bf16 and f16: checked here on all
    2 x 3 x 256 cases with the kernel's constants (bit patterns) -- so the fused patch GEMM's A operand is the unfold pass's, bit for bit."""
    import numpy as np
    src = open(os.path.join(ROOT, "plip_amd", "csrc", "gemm.h")).read()
    m = re.search(r"const unsigned A\[3\] = \{(0x[0-9a-f]+)u, (0x[0-9a-f]+)u, (0x[0-9a-f]+)u\}, Bc\[3\] = \{(0x[0-9a-f]+)u, (0x[0-9a-f]+)u, (0x[0-9a-f]+)u\}", src)
    assert m, "u8_norm's constants not found in gemm.h"
    bits = np.array([int(x, 16) for x in m.groups()], dtype=np.uint32)
    A, Bc = bits[:3].view(np.float32), bits[3:].view(np.float32)
    mean = np.array([0.48145466, 0.4578275, 0.40821073], dtype=np.float32)
    istd = (np.float32(1.0) / np.array([0.26862954, 0.26130258, 0.27577711], dtype=np.float32)).astype(np.float32)
    assert np.array_equal(A, (istd.astype(np.float64) / 255.0).astype(np.float32)) and np.array_equal(Bc, (-mean.astype(np.float64) * istd.astype(np.float64)).astype(np.float32))
    b = np.arange(256, dtype=np.float32)

    def bf16(x):
        u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
        return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)

    for c in range(3):
        three = (((b / np.float32(255.0)).astype(np.float32) - mean[c]).astype(np.float32) * istd[c]).astype(np.float32)
        one = (b.astype(np.float64) * np.float64(A[c]) + np.float64(Bc[c])).astype(np.float32)    # fma: exact product and sum in fp64, ONE rounding
        assert np.array_equal(bf16(one), bf16(three)) and np.array_equal(one.astype(np.float16).view(np.uint16), three.astype(np.float16).view(np.uint16)), c


def test_named_test_hooks_reject_values_outside_their_range():
    """include/plipmi_test.h (round 6): the A/B hooks are separate, named, range-checked setters -- a stray value changes
    nothing and says so (the old multiplexed plipmi_set_gemm_variant let 5000 fall through into an unrelated switch).  No GPU needed."""
    from plip_amd import _lib
    lib = _lib.load()
    n = 0
    while lib.plipmi_gemm_variant_name(n):
        n += 1
    try:
        for fn, good, bad in ((lib.plipmi_test_force_gemm_tile, (-2, -1, 0, n - 1), (-3, n, 1000, 2001)),
                              (lib.plipmi_test_fused_qkv_attention, (0, 1, 2), (-1, 3, 3002)),
                              (lib.plipmi_test_patch_gather, (0, 1), (-1, 2, 4001))):
            for v in good:
                assert fn(v) == 0, (fn.__name__, v)
            for v in bad:
                assert fn(v) == 1 and str(v) in _lib.last_error(), (fn.__name__, v, _lib.last_error())
        assert lib.plipmi_test_remap_gemm_tile(2, 3) == 0 and lib.plipmi_test_remap_gemm_tile(2, -1) == 0
        for a, b in ((-1, 0), (n, 0), (0, n), (0, -2)):
            assert lib.plipmi_test_remap_gemm_tile(a, b) == 1
    finally:
        lib.plipmi_test_reset_hooks()
    assert lib.plipmi_get_pass_batch(None) == 0
    import ctypes as C
    out, ratio = C.c_void_p(1), C.c_float(0)
    assert lib.plipmi_clone(None, C.byref(out)) != 0 and "null" in _lib.last_error()          # no source handle: an error, no crash
    assert lib.plipmi_streams_overlap(None, None, None, C.byref(ratio)) != 0
