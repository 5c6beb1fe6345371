"""GPU-side diagnostics (run on the MI355X box through gpurun; never part of the product path).

    python tools/gpu_diag.py gemm | tiny | vitb32 | gemmbench | e2e

Unlike the pytest suite nothing here stops at the first mismatch: every section prints a
table of errors / timings so one gpurun call localises a wrong kernel.
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle import clip_oracle as O  # noqa: E402
from oracle.make_golden import case_inputs  # noqa: E402
from plip_amd import weights as W  # noqa: E402
from plip_amd.config import get_config  # noqa: E402
from plip_amd.kernel_entries import gemm_nt, gemm_variants  # noqa: E402
from plip_amd.model import PlipModel  # noqa: E402

dev = torch.device("cuda:0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sec_gemm():
    print("device", torch.cuda.get_device_name(0), "variants", gemm_variants())
    g = torch.Generator().manual_seed(0)
    for dtype in (torch.float32, torch.bfloat16):
        for (M, N, K) in ((300, 512, 768), (77, 256, 64)):
            a = torch.randn(M, K, generator=g).to(dev).to(dtype)
            w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev).to(dtype)
            bias = torch.randn(N, generator=g).to(dev)
            c0 = torch.randn(M, N, generator=g).to(dev)
            base = a.double() @ w.double().T
            for variant in [-2] + list(range(len(gemm_variants()))):
                for epi in (0, 1, 2, 3):
                    ref = base + (bias.double() if epi < 3 else 0)
                    if epi == 1:
                        ref = ref * torch.sigmoid(1.702 * ref)
                    if epi == 2:
                        ref = ref + c0.double()
                    if epi == 3:
                        ref = 0.5 * ref
                    try:
                        y = gemm_nt(a, w, bias, epilogue=epi, variant=variant, alpha=0.5, out=c0.clone() if epi == 2 else None)
                        torch.cuda.synchronize()
                        err = (y.double() - ref).abs()
                        bad = (err > 0.05).sum().item()
                        print(f"{str(dtype):16s} {M}x{N}x{K} variant {variant:2d} epi {epi}: max err {err.max().item():.3e} "
                              f"bad {bad}/{err.numel()}" + ("" if bad == 0 else f"  first bad idx {torch.nonzero(err > 0.05)[0].tolist()}"))
                    except Exception as e:
                        print(f"{dtype} {M}x{N}x{K} variant {variant} epi {epi}: EXC {e}")


def _report(model, cfg, sd, px, ids, mask, name):
    eng = model.engine
    _, vh = O.vision_tower(px, sd, cfg, return_hidden=True)
    _, th = O.text_tower(ids, sd, cfg, mask, return_hidden=True)
    tpx, tids = torch.from_numpy(px), torch.from_numpy(ids)
    vl = sorted(set([0, 1, 2, cfg.v_layers // 2, cfg.v_layers]))
    for l in [x for x in vl if x <= cfg.v_layers]:
        h = eng.hidden("vision", l, tpx).cpu().numpy()
        print(f"{name} vision hidden[{l}] max err {np.abs(h - vh[l]).max():.3e}  (|h| max {np.abs(vh[l]).max():.2f})")
    m = mask[:, :, None].astype(np.float32)
    for l in [x for x in vl if x <= cfg.t_layers]:
        h = eng.hidden("text", l, tids).cpu().numpy()
        print(f"{name} text   hidden[{l}] max err {np.abs((h - th[l]) * m).max():.3e}  (|h| max {np.abs(th[l]).max():.2f})")
    ref = O.clip_forward(px, ids, sd, cfg, mask)
    out = model(input_ids=tids, pixel_values=tpx, attention_mask=torch.from_numpy(mask))
    img = model.get_image_features(pixel_values=tpx).cpu().numpy()
    txt = model.get_text_features(input_ids=tids, attention_mask=torch.from_numpy(mask)).cpu().numpy()
    scale = np.exp(np.float64(sd["logit_scale"]))
    print(f"{name} image_features err {np.abs(img - ref['image_features']).max():.3e}  text_features err "
          f"{np.abs(txt - ref['text_features']).max():.3e}")
    print(f"{name} image_embeds err {np.abs(out.image_embeds.cpu().numpy() - ref['image_embeds']).max():.3e}  text_embeds err "
          f"{np.abs(out.text_embeds.cpu().numpy() - ref['text_embeds']).max():.3e}  cosine-logits err "
          f"{np.abs(out.logits_per_image.cpu().numpy() - ref['logits_per_image']).max() / scale:.3e}")


def sec_tiny():
    for dtype in ("f32", "bf16"):
        cfg, sd, px, ids, mask = case_inputs("tiny_b6")
        model = PlipModel(cfg, sd, dtype=dtype, max_batch=8)
        _report(model, cfg, sd, px, ids, mask, f"tiny/{dtype}")


def sec_vitb32():
    for dtype in ("f32", "bf16"):
        cfg, sd, px, ids, mask = case_inputs("vitb32_b4")
        model = PlipModel(cfg, sd, dtype=dtype, max_batch=8)
        _report(model, cfg, sd, px, ids, mask, f"vitb32/{dtype}")
        # a second weight seed / bigger batch for the bf16 cosine bar
        cfg2 = cfg
        sd2 = W.synthetic_state_dict(cfg2, 5)
        px2 = W.synthetic_pixels(cfg2, 16, 6)
        ids2, mask2 = W.synthetic_ids(cfg2, 16, 7)
        m2 = PlipModel(cfg2, sd2, dtype=dtype, max_batch=16)
        ref = O.clip_forward(px2, ids2, sd2, cfg2, mask2)
        out = m2(input_ids=torch.from_numpy(ids2), pixel_values=torch.from_numpy(px2), attention_mask=torch.from_numpy(mask2))
        scale = np.exp(np.float64(sd2["logit_scale"]))
        print(f"vitb32/{dtype} seed5 b16: cosine-logits err {np.abs(out.logits_per_image.cpu().numpy() - ref['logits_per_image']).max() / scale:.3e} "
              f"argmax agree {(out.logits_per_image.cpu().numpy().argmax(1) == ref['logits_per_image'].argmax(1)).mean():.3f}")


def sec_attn():
    from plip_amd.kernel_entries import attention
    g = torch.Generator().manual_seed(0)
    for (B, S, H, causal) in ((3, 50, 12, False), (3, 77, 8, True), (2, 33, 2, True), (1, 128, 2, False)):
        qkv = torch.randn(B * S, 3 * H * 64, generator=g)
        qkv[:, : H * 64] *= 0.125 * 3
        x = qkv.double().reshape(B, S, 3, H, 64)
        q, k, v = (x[:, :, i].permute(0, 2, 1, 3) for i in range(3))
        lens = torch.randint(1, S + 1, (B,), generator=g)
        mask = (torch.arange(S)[None, :] < lens[:, None]).long()
        for use_mask in (False, True):
            for mode, dt, impl in (("valu_f32", torch.float32, 0), ("valu_bf16", torch.bfloat16, 0), ("mfma_bf16", torch.bfloat16, 1)):
                qd = qkv.to(dev).to(dt)
                xr = qd.double().cpu().reshape(B, S, 3, H, 64)
                q, k, v = (xr[:, :, i].permute(0, 2, 1, 3) for i in range(3))
                sc = q @ k.transpose(-1, -2)
                if causal:
                    sc = sc.masked_fill(~torch.tril(torch.ones(S, S, dtype=torch.bool)), float("-inf"))
                if use_mask:
                    sc = sc.masked_fill(~mask.bool()[:, None, None, :], float("-inf"))
                ref = (torch.softmax(sc, -1) @ v).permute(0, 2, 1, 3).reshape(B * S, H * 64)
                try:
                    out = attention(qd, B, S, H, causal, mask.to(dev) if use_mask else None, impl=impl)
                    torch.cuda.synchronize()
                    err = (out.double().cpu() - ref).abs()
                    print(f"attn {mode:10s} B{B} S{S} H{H} causal={causal} mask={use_mask}: max err {err.max().item():.3e} "
                          f"nan {int(torch.isnan(out).sum())} bad(>0.05) {int((err > 0.05).sum())}/{err.numel()}")
                except Exception as e:
                    print(f"attn {mode} B{B} S{S}: EXC {e}")
    # timing at production shapes
    for (B, S, H, causal) in ((256, 50, 12, False), (256, 77, 8, True), (256, 197, 12, False), (128, 257, 16, False), (64, 577, 16, False)):
        qd = torch.randn(B * S, 3 * H * 64, generator=g).to(dev).to(torch.bfloat16)
        for impl in (0, 1):
            ms = _time(lambda: attention(qd, B, S, H, causal, None, impl=impl), iters=20)
            print(f"attn timing B{B} S{S} H{H} impl {impl}: {ms * 1e3:.1f} us  ({4.0 * B * H * S * S * 64 / ms / 1e9:.1f} TFLOP/s dense-count, "
                  f"{B * S * 4 * H * 64 * 2 / ms / 1e6:.0f} GB/s)")


def _time(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def sec_gemmbench():
    """Every tile variant on the eight production GEMM shapes of the bs=256 step (random data)."""
    shapes = [("v.qkv", 12800, 2304, 768, 0), ("v.out", 12800, 768, 768, 2), ("v.fc1", 12800, 3072, 768, 1),
              ("v.fc2", 12800, 768, 3072, 2), ("v.patch", 12544, 768, 3072, 3),
              ("t.qkv", 19712, 1536, 512, 0), ("t.out", 19712, 512, 512, 2), ("t.fc1", 19712, 2048, 512, 1),
              ("t.fc2", 19712, 512, 2048, 2)]
    g = torch.Generator().manual_seed(0)
    for dtype in (torch.bfloat16, torch.float32):
        print(f"== {dtype}: TFLOP/s per variant {gemm_variants()}")
        for name, M, N, K, epi in shapes:
            a = torch.randn(M, K, generator=g).to(dev).to(dtype)
            w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev).to(dtype)
            bias = torch.randn(N, generator=g).to(dev)
            out = torch.zeros(M, N, device=dev, dtype=dtype if epi in (0, 1) else torch.float32)
            row = []
            for v in range(len(gemm_variants())):
                try:
                    ms = _time(lambda: gemm_nt(a, w, bias, epilogue=epi, variant=v, out=out), iters=20 if dtype == torch.bfloat16 else 5)
                    row.append(f"{2.0 * M * N * K / ms / 1e9:7.1f}")
                except Exception as e:
                    row.append("   n/a ")
            print(f"{name:8s} {M:6d}x{N:5d}x{K:5d} epi{epi}: " + " ".join(row))


def sec_lnbench():
    """LayerNorm-folded epilogues vs the plain ones on the bs=256 production shapes (product tiles), plus the LayerNorm
    kernel they replace: what the fold costs inside the GEMMs and what it saves outside."""
    from plip_amd.kernel_entries import gemm_nt_ln
    shapes = [("v.qkv", 12800, 2304, 768, 0), ("v.fc1", 12800, 3072, 768, 1), ("v.out", 12800, 768, 768, 2),
              ("v.fc2", 12800, 768, 3072, 2), ("t.qkv", 19712, 1536, 512, 0), ("t.fc1", 19712, 2048, 512, 1),
              ("t.out", 19712, 512, 512, 2), ("t.fc2", 19712, 512, 2048, 2)]
    g = torch.Generator().manual_seed(0)
    for name, M, N, K, epi in shapes:
        a = torch.randn(M, K, generator=g).to(dev).bfloat16()
        w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev).bfloat16()
        bias = torch.randn(N, generator=g).to(dev)
        row = []
        for v in (3, 4, 2, 5, 1):
            try:
                if epi in (0, 1):
                    out = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
                    x = torch.randn(M, K, generator=g).to(dev)              # plausible rows: partials {sum, centred M2} per 64 columns
                    xs = x.reshape(M, K // 64, 64)
                    st = torch.stack((xs.sum(-1), ((xs - xs.mean(-1, keepdim=True)) ** 2).sum(-1)), dim=-1).contiguous()
                    t0 = _time(lambda: gemm_nt(a, w, bias, epilogue=epi, variant=v, out=out), iters=20)
                    t1 = _time(lambda: gemm_nt_ln(epi, a, w, bias, st, variant=v, out=out), iters=20)
                else:
                    out = torch.zeros(M, N, device=dev, dtype=torch.float32)
                    t0 = _time(lambda: gemm_nt(a, w, bias, epilogue=2, variant=v, out=out), iters=20)
                    t1 = _time(lambda: gemm_nt_ln(2, a, w, bias, variant=v, out=out), iters=20)
                    hi = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
                    lo = torch.zeros(M, N, device=dev, dtype=torch.int16)
                    t2 = _time(lambda: gemm_nt_ln(3, a, w, bias, variant=v, out=(hi, lo)), iters=20)
                    row.append(f"v{v}: {t0 * 1e3:6.1f} -> {t1 * 1e3:6.1f} -> split {t2 * 1e3:6.1f} us")
                    continue
                row.append(f"v{v}: {t0 * 1e3:6.1f} -> {t1 * 1e3:6.1f} us")
            except Exception as e:
                row.append(f"v{v}: n/a")
        print(f"{name:6s} {M}x{N}x{K} epi{epi} plain -> folded: " + "   ".join(row))


def sec_latency():
    """Small-batch latency of one zero_shot_classification-sized step (the reference runs both towers at batch 8,
    plip.py:90-91): big-tile GEMMs vs the latency path (split-K small-M GEMMs, plipmi_set_latency_batch), eager launches vs
    hipGraph replay; wall time per call with a host sync (what a caller sees) and the host's enqueue cost."""
    import time
    cfg = get_config("ViT-B/32")
    sd = W.synthetic_state_dict(cfg, 0)
    model = PlipModel(cfg, sd, dtype="bf16", max_batch=32)
    eng = model.engine
    for B in (1, 8, 16, 32):
        px = torch.from_numpy(W.synthetic_pixels(cfg, B, 1)).to(dev)
        ids = torch.from_numpy(W.synthetic_ids(cfg, B, 2)[0]).to(dev)
        ref = None
        for path, lb in (("big tiles", 0), ("split-K", 32)):
            eng.set_latency_batch(lb)
            for mode, gb in (("eager", 0), ("graph", 32)):
                eng.set_graph_batch(gb)
                for _ in range(4):
                    eng.encode_image(px, True); eng.encode_text(ids, None, True)
                torch.cuda.synchronize()
                res = {}
                for what, fn in (("image", lambda: eng.encode_image(px, True)), ("text", lambda: eng.encode_text(ids, None, True)),
                                 ("pair_2streams", lambda: eng.encode_pair(px, ids, None, True, True))):
                    for _ in range(3):
                        fn()
                    torch.cuda.synchronize()
                    n = 30
                    t0 = time.perf_counter()
                    for _ in range(n):
                        fn()
                        torch.cuda.synchronize()
                    res[what] = (time.perf_counter() - t0) / n * 1e3
                    t0 = time.perf_counter()
                    for _ in range(n):
                        fn()
                    host = (time.perf_counter() - t0) / n * 1e3        # enqueue cost only (no sync inside)
                    torch.cuda.synchronize()
                    res[what + "_enqueue"] = host
                out = torch.cat(eng.encode_pair(px, ids, None, True, True))
                if ref is None:
                    ref = out
                print(f"B={B:3d} {path:9s} {mode:5s}: " + "  ".join(f"{k} {v:.3f} ms" for k, v in res.items()) +
                      f"   | max |embedding diff| vs big tiles {float((out - ref).abs().max()):.1e}")
    model.engine.close()


def sec_latprof():
    """For rocprofv3 --kernel-trace --stats: the B=8 towers on the latency path (argv[2] = 1) or the big tiles (0), graph replay,
    20 calls each -- which kernels a small call spends its time in."""
    cfg = get_config("ViT-B/32")
    model = PlipModel(cfg, W.synthetic_state_dict(cfg, 0), dtype="bf16", max_batch=32)
    eng = model.engine
    B = 8
    px = torch.from_numpy(W.synthetic_pixels(cfg, B, 1)).to(dev)
    ids = torch.from_numpy(W.synthetic_ids(cfg, B, 2)[0]).to(dev)
    eng.set_latency_batch(32 if (len(sys.argv) > 2 and sys.argv[2] == "1") else 0)
    eng.set_graph_batch(32)
    for _ in range(4):
        eng.encode_image(px, True); eng.encode_text(ids, None, True)
    torch.cuda.synchronize()
    for _ in range(20):
        eng.encode_image(px, True)
        torch.cuda.synchronize()
    for _ in range(20):
        eng.encode_text(ids, None, True)
        torch.cuda.synchronize()
    model.engine.close()


def sec_zeroshot():
    """PLIP.zero_shot_classification (plip.py:89-103: both encoders at the hard-coded batch_size=8) on 256 native tiles + 10
    labels: one engine call per caller batch (round 3's behaviour, PLIP.coalesce = False) vs the caller's batches handed to the
    towers max_batch rows at a time (the default) -- same labels, wall time per call incl. host work and the copy back."""
    import time
    from plip_amd.plip import PLIP
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_api import fake_tokenizer
    cfg = get_config("ViT-B/32")
    model = PlipModel(cfg, W.synthetic_state_dict(cfg, 0), dtype="bf16", max_batch=256)
    plip = PLIP(model=model, tokenizer=fake_tokenizer(cfg))
    rs = np.random.RandomState(0)
    tiles = [rs.randint(0, 256, size=(cfg.image_size, cfg.image_size, 3), dtype=np.uint8) for _ in range(256)]
    labels = [f"an h&e image of tissue class {i} " + "x " * i for i in range(10)]
    res = {}
    for co in (False, True, False, True):
        plip.coalesce = co
        plip.zero_shot_classification(tiles, labels)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            pred = plip.zero_shot_classification(tiles, labels)
        res.setdefault(co, []).append(((time.perf_counter() - t0) / 5 * 1e3, pred))
    a, b = min(r[0] for r in res[False]), min(r[0] for r in res[True])
    same = res[False][0][1] == res[True][0][1]
    print(f"zero_shot_classification(256 tiles, 10 labels): one engine call per batch of 8: {a:7.2f} ms   coalesced to max_batch: {b:7.2f} ms   "
          f"({a / b:.1f}x)   identical labels: {same}")
    model.engine.close()


def sec_libgemm():
    """Calibration only (never used by the product): what the vendor GEMM library (hipBLASLt/rocBLAS behind
    torch.nn.functional.linear / torch.mm) reaches on the production shapes -- an external yardstick for gemm_nt.  Like for
    like: the plain bias epilogue with a 16-bit output on both sides, same random operands, interleaved in one process, best
    of three rounds of 20 launches.  gemm_nt: the tile the engine's cost model picks, and the best tile of the table."""
    import torch.nn.functional as F
    shapes = [("v.qkv", 12800, 2304, 768), ("v.out", 12800, 768, 768), ("v.fc1", 12800, 3072, 768),
              ("v.fc2", 12800, 768, 3072), ("t.qkv", 19712, 1536, 512), ("t.out", 19712, 512, 512),
              ("t.fc1", 19712, 2048, 512), ("t.fc2", 19712, 512, 2048), ("big", 8192, 8192, 8192)]
    names = gemm_variants()
    LV = [int(x) for x in sys.argv[2:] if x.isdigit()] or [2, 3, 4, 5, 6]
    for dt in (torch.bfloat16, torch.float16):
        g = torch.Generator().manual_seed(0)
        print(f"== {dt}")
        for name, M, N, K in shapes:
            a = torch.randn(M, K, generator=g).to(dev).to(dt)
            w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev).to(dt)
            bias = torch.randn(N, generator=g).to(dev)
            bias_h = bias.to(dt)
            out = torch.zeros(M, N, device=dev, dtype=dt)
            best = {}
            for rep in range(3):
                for key, fn in [("F.linear", lambda: F.linear(a, w, bias_h)), ("torch.mm", lambda: torch.mm(a, w.t(), out=out)),
                                ("own", lambda: gemm_nt(a, w, bias, epilogue=0, out=out))] + \
                               [(v, (lambda v=v: gemm_nt(a, w, bias, epilogue=0, variant=v, out=out))) for v in LV]:
                    ms = _time(fn, iters=20, warm=3)
                    best[key] = min(best.get(key, 1e9), ms)
            f = 2.0 * M * N * K / 1e9
            bv = min(LV, key=lambda v: best[v])
            print(f"{name:6s} {M:6d}x{N:5d}x{K:5d}: F.linear+bias {f / best['F.linear']:7.1f}  torch.mm {f / best['torch.mm']:7.1f}  "
                  f"gemm_nt(bias) engine's tile {f / best['own']:7.1f}  best tile {f / best[bv]:7.1f} ({names[bv]})  TF/s   "
                  f"[{best['F.linear'] * 1e3:.1f} / {best['own'] * 1e3:.1f} us]   per tile: " + " ".join(f"v{v} {f / best[v]:6.1f}" for v in LV))


def sec_towerswap():
    """BASELINE configs[4] tower swap: ViT-B/16, ViT-L/14 and ViT-L/14@336 (synthetic weights, bf16)."""
    for arch, B in (("ViT-B/16", 128), ("ViT-L/14", 64), ("ViT-L/14@336px", 32)):
        cfg = get_config(arch)
        sd = W.synthetic_state_dict(cfg, 0)
        model = PlipModel(cfg, sd, dtype="bf16", max_batch=B)
        px = torch.from_numpy(W.synthetic_pixels(cfg, B, 1)).to(dev)
        ids_np, mask_np = W.synthetic_ids(cfg, B, 2)
        ids = torch.from_numpy(ids_np).to(dev)
        ms_i = _time(lambda: model.engine.encode_image(px), iters=5, warm=2)
        ms_t = _time(lambda: model.engine.encode_text(ids, None), iters=5, warm=2)
        fi, ft = cfg.image_flops() * B, cfg.text_flops() * B
        print(f"{arch:16s} B={B}: image {ms_i:8.2f} ms ({B / ms_i * 1e3:8.0f} img/s, {fi / ms_i / 1e9:6.1f} TF/s)   "
              f"text {ms_t:7.2f} ms ({B / ms_t * 1e3:8.0f} cap/s, {ft / ms_t / 1e9:6.1f} TF/s)")
        rows = []
        with model.engine.profile(rows):
            model.engine.encode_image(px)
            torch.cuda.synchronize()
        tot = sum(r["total_ms"] for r in rows)
        for r in sorted(rows, key=lambda r: -r["total_ms"])[:6]:
            print(f"      {r['name']:52s} {r['total_ms']:8.3f} ms {100 * r['total_ms'] / tot:5.1f}%")
        model.engine.close()
        del model
        torch.cuda.empty_cache()


def sec_ldpad():
    """Does padding the leading dimension (rows no longer a multiple of 2 KB apart) change the fill rate?"""
    from plip_amd.kernel_entries import gemm_nt_ld
    g = torch.Generator().manual_seed(0)
    shapes = [("v.qkv", 12800, 2304, 768, 0), ("v.fc1", 12800, 3072, 768, 1), ("v.fc2", 12800, 768, 3072, 2),
              ("t.fc1", 19712, 2048, 512, 1), ("t.fc2", 19712, 512, 2048, 2), ("v.out", 12800, 768, 768, 2)]
    for name, M, N, K, epi in shapes:
        row = []
        for pad in (0, 8, 32, 64, 72, 128):
            a = torch.randn(M, K + pad, generator=g).to(dev).to(torch.bfloat16)
            w = (torch.randn(N, K + pad, generator=g) / K ** 0.5).to(dev).to(torch.bfloat16)
            bias = torch.randn(N, generator=g).to(dev)
            out = torch.zeros(M, N, device=dev, dtype=torch.bfloat16 if epi in (0, 1) else torch.float32)
            res = []
            for v in (16, 8):
                if N % 256 and v == 16:
                    res.append("  n/a")
                    continue
                ms = _time(lambda: gemm_nt_ld(a, w, K, bias, epilogue=epi, variant=v, out=out), iters=20)
                res.append(f"{2.0 * M * N * K / ms / 1e9:6.1f}")
            row.append(f"pad{pad}: " + "/".join(res))
        print(f"{name:6s} {M}x{N}x{K} (variants 16/8): " + "   ".join(row))


def sec_gemmone():
    """One variant / one shape in a loop -- the target of rocprofv3 --pmc runs.
    usage: gpu_diag.py gemmone <variant> <M> <N> <K> <epi> [bf16|f32] [iters]"""
    v, M, N, K, epi = (int(x) for x in sys.argv[2:7])
    dtype = torch.float32 if (len(sys.argv) > 7 and sys.argv[7] == "f32") else torch.bfloat16
    iters = int(sys.argv[8]) if len(sys.argv) > 8 else 20
    g = torch.Generator().manual_seed(0)
    a = torch.randn(M, K, generator=g).to(dev).to(dtype)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev).to(dtype)
    bias = torch.randn(N, generator=g).to(dev)
    out = torch.zeros(M, N, device=dev, dtype=dtype if epi in (0, 1) else torch.float32)
    ms = _time(lambda: gemm_nt(a, w, bias, epilogue=epi, variant=v, out=out), iters=iters, warm=3)
    print(f"gemmone variant {v} {M}x{N}x{K} epi{epi} {dtype}: {ms * 1e3:.1f} us  {2.0 * M * N * K / ms / 1e9:.1f} TFLOP/s")


def sec_gemmtrace():
    """In-kernel timeline of one GEMM launch: where a workgroup's lifetime goes.
    usage: gpu_diag.py gemmtrace <variant> <M> <N> <K> <epi>"""
    v, M, N, K, epi = (int(x) for x in sys.argv[2:7]) if len(sys.argv) > 6 else (5, 12800, 3072, 768, 1)
    g = torch.Generator().manual_seed(0)
    dtype = torch.bfloat16
    a = torch.randn(M, K, generator=g).to(dev).to(dtype)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev).to(dtype)
    bias = torch.randn(N, generator=g).to(dev)
    out = torch.zeros(M, N, device=dev, dtype=dtype if epi in (0, 1) else torch.float32)
    names = gemm_variants()
    bm, bn = (int(x) for x in names[v].split("_")[0].split("x"))
    nblk = ((M + bm - 1) // bm) * (N // bn)
    for _ in range(3):
        gemm_nt(a, w, bias, epilogue=epi, variant=v, out=out)
    tr = torch.zeros(nblk * 8, dtype=torch.int64, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    gemm_nt(a, w, bias, epilogue=epi, variant=v, out=out, trace=tr)
    e1.record()
    torch.cuda.synchronize()
    t = tr.cpu().numpy().reshape(nblk, 8).astype(np.int64)
    ms = e0.elapsed_time(e1)
    xcc = (t[:, 5] >> 32) & 0xF
    # cycle stamps (s_memtime: shader cycles, one counter per XCD -> only differences inside a workgroup mean anything) and
    # wall stamps (s_memrealtime: 100 MHz, one counter for the device -> start offsets and lifetimes in real time)
    pro, loop, epi_t, life = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 3] - t[:, 0]
    real0 = (t[:, 7] & 0xFFFFFFFF).astype(np.int64)
    real_life_us = ((t[:, 7] >> 32) & 0xFFFFFFFF).astype(np.float64) / 100.0
    start_us = ((real0 - real0.min()) & 0xFFFFFFFF).astype(np.float64) / 100.0
    end_us = start_us + real_life_us
    clk = life / np.maximum(real_life_us, 1e-9) / 1e3            # GHz, per workgroup
    print(f"variant {names[v]} {M}x{N}x{K} epi{epi}: {nblk} workgroups, kernel {ms * 1e3:.1f} us by events "
          f"({2.0 * M * N * K / ms / 1e9:.0f} TFLOP/s); first start -> last end {end_us.max():.1f} us; "
          f"shader clock while the workgroups ran: median {np.median(clk):.2f} GHz (min {clk.min():.2f}, max {clk.max():.2f})")
    qc = lambda x: (f"min {x.min():8.0f}  p50 {np.median(x):8.0f}  p90 {np.percentile(x, 90):8.0f}  max {x.max():8.0f} cycles"
                    f"  = p50 {np.median(x) / np.median(clk) / 1e3:6.2f} us at that clock")
    qu = lambda x: f"min {x.min():7.2f}  p50 {np.median(x):7.2f}  p90 {np.percentile(x, 90):7.2f}  max {x.max():7.2f} us"
    kt = max(1, int(t[0, 6] & 0xFFFFFFFF) - 1)
    lds_alloc = (t[:, 6] >> 32) & 0xFFFFFFFF
    print("  start offset :", qu(start_us))
    print("  prologue     :", qc(pro))
    print("  main loop    :", qc(loop), f"  ({np.median(loop) / kt:.0f} cycles per k-tile, {kt} tiles)")
    print("  epilogue     :", qc(epi_t))
    print("  lifetime     :", qc(life), "  wall:", qu(real_life_us))
    print("  end          :", qu(end_us))
    print("  workgroups per XCC:", np.bincount(xcc, minlength=8).tolist())
    # two workgroups per CU (variant 7): which LDS allocation a workgroup got (HW_REG LDS_ALLOC, LDS_BASE = bits 7:0), and when
    # the workgroups of each slot reached their epilogue / ended -- staggered slots mean one slot's stores run under the other's K loop
    bases = sorted(set((lds_alloc & 0xFF).tolist()))
    print(f"  LDS_ALLOC register values: {sorted(set(hex(int(x)) for x in lds_alloc.tolist()))[:6]}")
    if len(bases) > 1:
        loop_end_us = start_us + (t[:, 2] - t[:, 0]) / np.maximum(clk, 1e-9) / 1e3
        for b in bases:
            sel = (lds_alloc & 0xFF) == b
            print(f"    LDS_BASE {b:3d}: {int(sel.sum()):4d} workgroups  start p50 {np.median(start_us[sel]):6.2f}  K loop done p50 {np.median(loop_end_us[sel]):6.2f}"
                  f"  end p50 {np.median(end_us[sel]):6.2f} us   loop {np.median(loop[sel]) / kt:6.0f} cyc/tile  epilogue p50 {np.median(epi_t[sel]) / np.median(clk) / 1e3:5.2f} us")


def sec_qkvattn():
    """The text tower's fused q/k/v + attention kernel on the bs=256 production shape: us per launch against the two kernels it
    replaces (warm, and with the A operand cycling through buffer sets larger than the Infinity Cache), and its in-kernel
    timeline.  usage: gpu_diag.py qkvattn [B S H]"""
    from plip_amd.kernel_entries import attention, gemm_nt_ln, qkv_attention
    B, S, H = (int(x) for x in sys.argv[2:5]) if len(sys.argv) > 4 else (256, 77, 8)
    D = H * 64
    g = torch.Generator().manual_seed(0)
    hdt = torch.bfloat16
    nb = min(8, max(2, int(600e6 // (B * S * D * 2 * 5)) + 1))
    xs = [torch.randn(B * S, D, generator=g) for _ in range(nb)]
    a = [x.to(dev).to(hdt) for x in xs]
    st = []
    for x in xs:
        r = x.to(dev).reshape(B * S, H, 64)
        st.append(torch.stack((r.sum(-1), ((r - r.mean(-1, keepdim=True)) ** 2).sum(-1)), dim=-1).contiguous())
    w = torch.randn(3 * D, D, generator=g) / D ** 0.5
    w = (w - w.mean(-1, keepdim=True)).to(dev).to(hdt)
    c2 = (torch.randn(3 * D, generator=g) * 0.2).to(dev)
    mask = (torch.arange(S)[None, :] < torch.randint(8, S + 1, (B,), generator=g)[:, None]).long().to(dev)
    qkvs = [torch.empty(B * S, 3 * D, device=dev, dtype=hdt) for _ in range(nb)]
    state = {"i": 0}

    def two(cold):
        i = state["i"] % nb if cold else 0
        state["i"] += 1
        gemm_nt_ln(0, a[i], w, c2, st[i], variant=-1, out=qkvs[i])
        return attention(qkvs[i], B, S, H, True, mask, impl=1)

    def one(cold, mode=1):
        i = state["i"] % nb if cold else 0
        state["i"] += 1
        return qkv_attention(a[i], w, c2, st[i], B, S, H, mode, mask)

    assert torch.equal(one(False), two(False))
    for cold in (False, True):
        t2 = min(_time(lambda: two(cold), iters=3 * nb if cold else 30, warm=nb) for _ in range(2)) * 1e3
        t1 = min(_time(lambda: one(cold), iters=3 * nb if cold else 30, warm=nb) for _ in range(2)) * 1e3
        print(f"B={B} S={S} H={H} {'cold' if cold else 'warm'} ({nb} buffer sets): q/k/v GEMM + attention_mfma {t2:6.1f} us   fused qkv_attention {t1:6.1f} us")
    ngrp = (B + 3) // 4
    nblk = ngrp * H
    tr = torch.zeros(nblk * 8, dtype=torch.int64, device=dev)
    qkv_attention(a[0], w, c2, st[0], B, S, H, True, mask, trace=tr)
    torch.cuda.synchronize()
    t = tr.cpu().numpy().reshape(nblk, 8).astype(np.int64)
    real0 = (t[:, 7] & 0xFFFFFFFF).astype(np.int64)
    life_us = ((t[:, 7] >> 32) & 0xFFFFFFFF).astype(np.float64) / 100.0
    start_us = ((real0 - real0.min()) & 0xFFFFFFFF).astype(np.float64) / 100.0
    clk = (t[:, 4] - t[:, 0]) / np.maximum(life_us, 1e-9) / 1e3
    q = lambda x: f"p50 {np.median(x):8.0f} cycles = {np.median(x) / np.median(clk) / 1e3:6.2f} us (min {x.min() / np.median(clk) / 1e3:5.2f}, p90 {np.percentile(x, 90) / np.median(clk) / 1e3:5.2f})"
    print(f"  {nblk} workgroups; first start -> last end {np.max(start_us + life_us):.1f} us; clock median {np.median(clk):.2f} GHz; start offset p50 {np.median(start_us):.2f} p90 {np.percentile(start_us, 90):.2f} us")
    print("  prologue (2 tiles requested, rstd rows, first tile landed):", q(t[:, 1] - t[:, 0]))
    print(f"  K loop ({D // 64} tiles):", q(t[:, 2] - t[:, 1]), f" {np.median(t[:, 2] - t[:, 1]) / (D // 64):.0f} cycles per K tile (MFMA issue 1920)")
    print("  q/k/v -> LDS images :", q(t[:, 3] - t[:, 2]))
    print("  attention + stores  :", q(t[:, 4] - t[:, 3]))
    print("  lifetime            :", q(t[:, 4] - t[:, 0]))


def _step_inputs(B=256, arch="ViT-B/32"):
    cfg = get_config(arch)
    sd = W.synthetic_state_dict(cfg, 0)
    px = torch.from_numpy(W.synthetic_pixels(cfg, B, 1000)).to(dev)
    ids_np, mask_np = W.synthetic_ids(cfg, B, 2000)
    return cfg, sd, px, torch.from_numpy(ids_np).to(dev), torch.from_numpy(mask_np).to(dev)


def sec_policy():
    """In-process interleaved A/B of the step under forced tiles / write-through stores, one and two streams, per engine
    dtype.  arguments: list of 'dtype:variant:wt' (variant -1 = the cost model's own choice)."""
    from plip_amd import _lib
    from plip_amd.dist import sharded_pair_logits
    lib = _lib.load()
    B = 256
    cfg, sd, px, ids, mask = _step_inputs(B)
    arms = sys.argv[2:] or ["bf16:-1", "bf16:4", "bf16:6", "f16:-1"]
    models = {}
    res = {(a, ov): [] for a in arms for ov in (False, True)}
    for rep in range(4):
        for arm in arms:
            dt, var = arm.split(":")
            if dt not in models:
                models[dt] = PlipModel(cfg, sd, dtype=dt, max_batch=B)
            model = models[dt]
            lib.plipmi_test_force_gemm_tile(int(var))
            for ov in (False, True):
                ms = _time(lambda: sharded_pair_logits(model, px, ids, mask, overlap=ov), iters=10, warm=2)
                if rep:                      # rep 0 = warm-up of clocks / caches
                    res[(arm, ov)].append(ms)
    lib.plipmi_test_reset_hooks()
    for arm in arms:
        a, b = res[(arm, False)], res[(arm, True)]
        print(f"dtype:variant {arm:12s} one stream {np.median(a):6.3f} ms (min {min(a):6.3f})   "
              f"two streams {np.median(b):6.3f} ms (min {min(b):6.3f})  -> {B / np.median(b) * 1e3:7.0f} pairs/s")


def sec_tiles():
    """The product epilogues (LayerNorm-folded consumers, split-plane producers) on the bs=256 production shapes: us per
    launch for every 256-wide tile variant, with and without write-through stores, bf16 (f16 on request) -- interleaved
    in one process.  The vendor library (plain bias epilogue) on the same shapes as the yardstick."""
    import torch.nn.functional as F
    from plip_amd import _lib
    from plip_amd.kernel_entries import gemm_nt_ln, split_planes
    lib = _lib.load()
    hdt = torch.float16 if (len(sys.argv) > 2 and sys.argv[2] == "f16") else torch.bfloat16
    shapes = [("v.qkv", 12800, 2304, 768, 0), ("v.fc1", 12800, 3072, 768, 1), ("v.out", 12800, 768, 768, 3),
              ("v.fc2", 12800, 768, 3072, 3), ("t.qkv", 19712, 1536, 512, 0), ("t.fc1", 19712, 2048, 512, 1),
              ("t.out", 19712, 512, 512, 3), ("t.fc2", 19712, 512, 2048, 3)]
    variants = [int(x) for x in sys.argv[2:] if x.lstrip("-").isdigit()] or [2, 3, 4, 5, 6]
    names = gemm_variants()
    print("variants:", {v: names[v] for v in variants}, "dtype", hdt)
    g = torch.Generator().manual_seed(0)
    for name, M, N, K, mode in shapes:
        a = torch.randn(M, K, generator=g).to(dev).to(hdt)
        w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev).to(hdt)
        bias = torch.randn(N, generator=g).to(dev)
        fl = 2.0 * M * N * K / 1e6
        if mode == 3:
            hi, lo = split_planes(torch.randn(M, N, generator=g).to(dev), hdt)
            run = lambda v: gemm_nt_ln(3, a, w, bias, variant=v, out=(hi, lo))
        else:
            out = torch.zeros(M, N, device=dev, dtype=hdt)
            xs = torch.randn(M, K, generator=g).to(dev).reshape(M, K // 64, 64)
            st = torch.stack((xs.sum(-1), ((xs - xs.mean(-1, keepdim=True)) ** 2).sum(-1)), dim=-1).contiguous()
            run = lambda v: gemm_nt_ln(mode, a, w, bias, st, variant=v, out=out)
        best = {}
        for rep in range(2):
            for v in variants:
                us = _time(lambda: run(v), iters=20, warm=3) * 1e3
                best[v] = min(best.get(v, 1e9), us)
        lib_us = _time(lambda: F.linear(a, w, bias.to(hdt)), iters=20, warm=3) * 1e3
        row = "  ".join(f"v{v}: {best[v]:6.1f}" for v in variants)
        bv = min(best, key=best.get)
        print(f"{name:6s} {M}x{N}x{K} {('ln_bias', 'ln_qgelu', '', 'resid_split')[mode]:11s} us  {row}   "
              f"| best v{bv} {fl / best[bv]:7.1f} TF/s | vendor F.linear+bias {lib_us:6.1f} us {fl / lib_us:7.1f} TF/s")


def sec_stepab():
    """In-process interleaved A/B of the bs=256 bf16 step under remapped tile choices (test hook plipmi_test_remap_gemm_tile(a, b):
    the cost model's choice a runs as tile b).  arguments: arms like  base  2>3  6>5,3>2  f0  g0"""
    from plip_amd import _lib
    from plip_amd.dist import sharded_pair_logits
    lib = _lib.load()
    B = int(os.environ.get("STEPAB_BATCH", "256"))            # STEPAB_ARCH / STEPAB_BATCH: the same A/B on another tower pair
    cfg, sd, px, ids, mask = _step_inputs(B, os.environ.get("STEPAB_ARCH", "ViT-B/32"))
    model = PlipModel(cfg, sd, dtype=os.environ.get("STEPAB_DTYPE", "bf16"), max_batch=B)
    arms = sys.argv[2:] or ["base"]
    names = gemm_variants()
    ref = None
    res = {a: {False: [], True: []} for a in arms}
    for rep in range(3):
        for arm in arms:
            lib.plipmi_test_reset_hooks()
            if arm != "base":
                for pair in arm.split(","):
                    if pair.startswith("g"):                  # g0 / g1: unfold pass + plain patch GEMM / im2col on load (default)
                        lib.plipmi_test_patch_gather(int(pair[1:]))
                        continue
                    if pair.startswith("f"):                  # f0 / f1: text q/k/v + attention as two kernels / fused (default)
                        lib.plipmi_test_fused_qkv_attention(int(pair[1:]))
                        continue
                    a, b = (int(x) for x in pair.split(">"))
                    lib.plipmi_test_remap_gemm_tile(a, b)
            for ov in (False, True):
                ms = _time(lambda: sharded_pair_logits(model, px, ids, mask, overlap=ov), iters=20, warm=3)
                res[arm][ov].append(ms)
            if rep == 0:
                out = sharded_pair_logits(model, px, ids, mask, overlap=False)[0].float().cpu()
                if ref is None:
                    ref = out
                print(f"arm {arm:24s} logits max |diff| vs first arm {float((out - ref).abs().max()):.3e}")
    lib.plipmi_test_reset_hooks()
    for arm in arms:
        one, two = res[arm][False], res[arm][True]
        print(f"{arm:24s} one stream {min(one):6.3f} ms (median {sorted(one)[1]:6.3f})   two streams {min(two):6.3f} ms (median {sorted(two)[1]:6.3f})"
              f"   {B / min(two) * 1e3:8.0f} pairs/s")
    print("tiles:", {i: n for i, n in enumerate(names)})
    model.engine.close()


def sec_cumask():
    """Experiment: the two towers on CU-masked HIP streams (hipExtStreamCreateWithCUMask) instead of two plain streams -- each
    tower owns a share of the CUs, so one tower's K loops run beside the other's memory-bound phases by construction.
    arguments: masks as  name:spec  with spec = 'lo<N>' (first N mask bits), 'xcd<K>' (bits i with i % 8 < K), 'even'"""
    import ctypes as C
    from plip_amd.dist import sharded_pair_logits
    hip = C.CDLL("libamdhip64.so")
    B = 256
    cfg, sd, px, ids, mask = _step_inputs(B)
    model = PlipModel(cfg, sd, dtype="bf16", max_batch=B)
    eng = model.engine
    ncu = torch.cuda.get_device_properties(0).multi_processor_count

    def bits(spec, invert=False):
        if spec.startswith("lo"):
            n = int(spec[2:]); on = [i < n for i in range(ncu)]
        elif spec.startswith("xcd"):
            k = int(spec[3:]); on = [(i % 8) < k for i in range(ncu)]
        elif spec == "even":
            on = [(i % 2) == 0 for i in range(ncu)]
        elif spec.startswith("se"):      # blocks of 8 consecutive bits, first K of every 16
            k = int(spec[2:]); on = [((i // 8) % 2) < 1 if k == 1 else True for i in range(ncu)]
        else:
            raise ValueError(spec)
        if invert:
            on = [not b for b in on]
        words = (C.c_uint32 * ((ncu + 31) // 32))()
        for i, b in enumerate(on):
            if b:
                words[i // 32] |= (1 << (i % 32))
        return words, sum(on)

    def masked_stream(spec, invert=False):
        words, n = bits(spec, invert)
        st = C.c_void_p()
        rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), C.c_uint32(len(words)), words)
        assert rc == 0, rc
        return torch.cuda.ExternalStream(st.value, device=dev), n

    def step_masked(vis, txt_s):
        main = torch.cuda.current_stream(dev)
        vis.wait_stream(main); txt_s.wait_stream(main)
        with torch.cuda.stream(txt_s):
            t = eng.encode_text(ids, mask, True)
        with torch.cuda.stream(vis):
            i = eng.encode_image(px, True)
        main.wait_stream(vis); main.wait_stream(txt_s)
        return eng.logits(i, t, scale=eng.logit_scale_exp, want_text=False)[0]

    ref = sharded_pair_logits(model, px, ids, mask, overlap=True)[0]
    print(f"{ncu} CUs; plain two streams {_time(lambda: sharded_pair_logits(model, px, ids, mask, overlap=True), iters=20, warm=3):.3f} ms, "
          f"one stream {_time(lambda: sharded_pair_logits(model, px, ids, mask, overlap=False), iters=20, warm=3):.3f} ms")
    for spec in sys.argv[2:] or ["lo128", "lo160", "lo152", "xcd4", "xcd5", "even"]:
        vis, nv = masked_stream(spec)
        txt_s, nt = masked_stream(spec, invert=True)
        out = step_masked(vis, txt_s)
        torch.cuda.synchronize()
        same = bool(torch.equal(out, ref))
        ms = min(_time(lambda: step_masked(vis, txt_s), iters=20, warm=3) for _ in range(2))
        # each tower alone on its share
        with torch.cuda.stream(vis):
            tv = _time(lambda: eng.encode_image(px, True), iters=10, warm=2)
        with torch.cuda.stream(txt_s):
            tt = _time(lambda: eng.encode_text(ids, mask, True), iters=10, warm=2)
        print(f"mask {spec:8s}: vision on {nv:3d} CUs, text on {nt:3d}: step {ms:.3f} ms ({B / ms * 1e3:.0f} pairs/s) bit-identical {same};  "
              f"alone on its share: vision {tv:.3f} ms, text {tt:.3f} ms")
    print(f"plain two streams again {_time(lambda: sharded_pair_logits(model, px, ids, mask, overlap=True), iters=20, warm=3):.3f} ms")
    model.engine.close()


def sec_mixed():
    """plipmi_config.text_f16_layers: cost and parity of a bf16 engine whose first N text blocks run on f16 operands -- the
    step interleaved across engines in one process, parity against the HF goldens of both bs=256 checkpoints."""
    from plip_amd.dist import sharded_pair_logits
    from oracle.make_golden import case_inputs
    B = 256
    cfg, sd, px, ids, mask = _step_inputs(B)
    arms = [int(a) for a in sys.argv[2:]] or [0, 2, 4, 6, 12]
    models = {n: PlipModel(cfg, sd, dtype="bf16", max_batch=B, text_f16_layers=n) for n in arms}
    res = {n: {False: [], True: []} for n in arms}
    for rep in range(3):
        for n in arms:
            for ov in (False, True):
                res[n][ov].append(_time(lambda: sharded_pair_logits(models[n], px, ids, mask, overlap=ov), iters=20, warm=3))
    gold = np.load(os.path.join(ROOT, "tests", "golden", "vitb32_b256.npz"))
    scale = float(np.exp(np.float64(sd["logit_scale"])))
    base = min(res[arms[0]][True])
    for n in arms:
        out = models[n](input_ids=ids, pixel_values=px, attention_mask=mask)
        cos = np.abs(out.logits_per_image.cpu().numpy() - gold["logits_per_image"]).max() / scale
        et = np.abs(out.text_embeds.cpu().numpy() - gold["text_embeds"]).max()
        ei = np.abs(out.image_embeds.cpu().numpy() - gold["image_embeds"]).max()
        one, two = res[n][False], res[n][True]
        print(f"first {n:2d} text blocks on f16: one stream {min(one):6.3f} ms  two streams {min(two):6.3f} ms (median {sorted(two)[1]:6.3f}) "
              f"{B / min(two) * 1e3:8.0f} pairs/s ({100 * (base / min(two) - 1):+5.1f} %)  |  cosine err {cos:.2e}  text_embeds {et:.2e}  image_embeds {ei:.2e}")
        models[n].engine.close()
    try:      # the heavy-tailed checkpoint
        cfg2, sd2, px2, ids2, mask2 = case_inputs("vitb32_b256_heavy")
        g2 = np.load(os.path.join(ROOT, "tests", "golden", "vitb32_b256_heavy.npz"))
        s2 = float(np.exp(np.float64(sd2["logit_scale"])))
        for n in arms:
            m = PlipModel(cfg2, sd2, dtype="bf16", max_batch=B, text_f16_layers=n)
            out = m(input_ids=torch.from_numpy(ids2), pixel_values=torch.from_numpy(px2), attention_mask=torch.from_numpy(mask2))
            print(f"heavy-tailed checkpoint, first {n:2d} on f16: cosine err {np.abs(out.logits_per_image.cpu().numpy() - g2['logits_per_image']).max() / s2:.2e}  "
                  f"text_embeds {np.abs(out.text_embeds.cpu().numpy() - g2['text_embeds']).max():.2e}  image_embeds {np.abs(out.image_embeds.cpu().numpy() - g2['image_embeds']).max():.2e}")
            m.engine.close()
    except Exception as e:
        print("heavy-tailed fixture:", repr(e))


def sec_cold():
    """The production GEMMs with operands that are NOT resident in the 256 MiB Infinity Cache: every launch takes its A
    operand and its residual planes / output from the next of a ring of buffers (> 600 MB in total), as the engine's
    kernels find them -- written by the previous kernel, read once.  Same launches with ONE buffer set (warm) beside it."""
    from plip_amd.kernel_entries import gemm_nt_ln, split_planes
    shapes = [("v.qkv", 12800, 2304, 768, 0), ("v.fc1", 12800, 3072, 768, 1), ("v.out", 12800, 768, 768, 3),
              ("v.fc2", 12800, 768, 3072, 3), ("t.fc2", 19712, 512, 2048, 3)]
    variants = [int(x) for x in sys.argv[2:]] or [2, 3, 4, 5, 6]
    g = torch.Generator().manual_seed(0)
    for name, M, N, K, mode in shapes:
        per = M * K * 2 + M * N * 4
        nb = max(2, int(700e6 // per) + 1)
        A = [torch.randn(M, K, generator=g).to(dev).bfloat16() for _ in range(nb)]
        w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev).bfloat16()
        bias = torch.randn(N, generator=g).to(dev)
        if mode == 3:
            planes = [split_planes(torch.randn(M, N, generator=g).to(dev)) for _ in range(nb)]
            def run(v, i):
                gemm_nt_ln(3, A[i], w, bias, variant=v, out=planes[i])
        else:
            outs = [torch.zeros(M, N, device=dev, dtype=torch.bfloat16) for _ in range(nb)]
            xs = torch.randn(M, K, generator=g).to(dev).reshape(M, K // 64, 64)
            st = torch.stack((xs.sum(-1), ((xs - xs.mean(-1, keepdim=True)) ** 2).sum(-1)), dim=-1).contiguous()
            def run(v, i):
                gemm_nt_ln(mode, A[i], w, bias, st, variant=v, out=outs[i])
        row = []
        for v in variants:
            res = {}
            for cold in (0, 1):
                state = {"i": 0}
                def once():
                    run(v, state["i"] % nb if cold else 0)
                    state["i"] += 1
                res[cold] = min(_time(once, iters=3 * nb if cold else 30, warm=nb) for _ in range(2)) * 1e3
            row.append(f"v{v}: warm {res[0]:6.1f} cold {res[1]:6.1f}")
        print(f"{name:6s} {M}x{N}x{K} ({nb} buffer sets): " + "   ".join(row) + " us")
        del A


def sec_power():
    """Is the step limited by the chip's power budget?  The same step (same kernels, same launches, same bytes) on inputs
    that make every sample identical -- all-zero pixels, one repeated caption: the operands of every GEMM then repeat row
    after row and toggle far fewer bits -- against the random batch.  (MI355X_MICROARCH.md, DVFS: the 8-phase GEMM runs
    15-21 % faster on zero-filled operands.)"""
    from plip_amd.dist import sharded_pair_logits
    B = 256
    cfg, sd, px, ids, mask = _step_inputs(B)
    model = PlipModel(cfg, sd, dtype="bf16", max_batch=B)
    px0 = torch.zeros_like(px)
    ids0 = ids[:1].repeat(B, 1).contiguous()
    mask0 = mask[:1].repeat(B, 1).contiguous()
    for rep in range(3):
        for name, a, b, c in (("random batch", px, ids, mask), ("identical samples", px0, ids0, mask0)):
            for ov in (True, False):
                ms = _time(lambda: sharded_pair_logits(model, a, b, c, overlap=ov), iters=20, warm=3)
                if rep:
                    print(f"{name:18s} {'two streams' if ov else 'one stream '}: {ms:6.3f} ms/step")


def sec_sustain():
    """Burst vs sustained: the same GEMM timed over 20 launches after an idle gap, and over ~3000 back-to-back launches
    (~0.2 s of continuous MFMA load) -- does the chip hold its burst clock?"""
    from plip_amd.kernel_entries import gemm_nt_ln, split_planes
    g = torch.Generator().manual_seed(0)
    for name, M, N, K, mode, variants in (("v.fc2", 12800, 768, 3072, 3, (4, 5)), ("v.fc1", 12800, 3072, 768, 1, (3, 5))):
        a = torch.randn(M, K, generator=g).to(dev).bfloat16()
        w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev).bfloat16()
        bias = torch.randn(N, generator=g).to(dev)
        if mode == 3:
            hi, lo = split_planes(torch.randn(M, N, generator=g).to(dev))
            run = lambda v: gemm_nt_ln(3, a, w, bias, variant=v, out=(hi, lo))
        else:
            out = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
            xs = torch.randn(M, K, generator=g).to(dev).reshape(M, K // 64, 64)
            st = torch.stack((xs.sum(-1), ((xs - xs.mean(-1, keepdim=True)) ** 2).sum(-1)), dim=-1).contiguous()
            run = lambda v: gemm_nt_ln(mode, a, w, bias, st, variant=v, out=out)
        for v in variants:
            torch.cuda.synchronize(); time.sleep(0.5)
            burst = _time(lambda: run(v), iters=20, warm=3) * 1e3
            torch.cuda.synchronize(); time.sleep(0.5)
            rows = []
            for chunk in range(6):                    # 6 x 500 launches, timed chunk by chunk without idling in between
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(500):
                    run(v)
                e1.record()
                rows.append((e0, e1))
            torch.cuda.synchronize()
            sus = [e0.elapsed_time(e1) / 500 * 1e3 for e0, e1 in rows]
            print(f"{name} variant {v}: burst of 20 launches {burst:6.1f} us; sustained chunks of 500 launches: " + " ".join(f"{x:6.1f}" for x in sus) + " us")


def sec_parity():
    """Where the cosine error of the 16-bit engines comes from at the benchmark size: vitb32_b256 against its HF golden,
    per engine form (dtype x LayerNorm fold x pooled last block)."""
    gold = np.load(os.path.join(ROOT, "tests", "golden", "vitb32_b256.npz"))
    cfg, sd, px, ids, mask = _step_inputs(256)
    assert np.array_equal(gold["ids"], ids.cpu().numpy())
    scale = float(np.exp(np.float64(sd["logit_scale"])))
    for dt in ("f32", "bf16", "f16"):
        for kw in ({}, {"ln_fold": False}, {"pooled_last_block": False}, {"text_f16": True}):
            if (dt == "f32" and kw) or ("text_f16" in kw and dt != "bf16"):
                continue
            model = PlipModel(cfg, sd, dtype=dt, max_batch=256, **kw)
            out = model(input_ids=ids, pixel_values=px, attention_mask=mask)
            lpi = out.logits_per_image.cpu().numpy() / scale
            want = gold["logits_per_image"] / scale
            e_img = np.abs(out.image_embeds.cpu().numpy() - gold["image_embeds"])
            e_txt = np.abs(out.text_embeds.cpu().numpy() - gold["text_embeds"])
            print(f"{dt:5s} {str(kw):30s} cosine max {np.abs(lpi - want).max():.3e} rms {np.sqrt(((lpi - want) ** 2).mean()):.3e} | "
                  f"image_embeds max {e_img.max():.3e} rms {np.sqrt((e_img ** 2).mean()):.3e} | text_embeds max {e_txt.max():.3e} "
                  f"rms {np.sqrt((e_txt ** 2).mean()):.3e} | argmax agreement {(lpi.argmax(1) == want.argmax(1)).mean():.4f}")
            model.engine.close()


def sec_overlap():
    """One-stream vs two-stream step time at bs=256 (bf16)."""
    from plip_amd.dist import sharded_pair_logits
    cfg = get_config("ViT-B/32")
    sd = W.synthetic_state_dict(cfg, 0)
    B = 256
    px = torch.from_numpy(W.synthetic_pixels(cfg, B, 1)).to(dev)
    ids_np, mask_np = W.synthetic_ids(cfg, B, 2)
    ids, mask = torch.from_numpy(ids_np).to(dev), torch.from_numpy(mask_np).to(dev)
    model = PlipModel(cfg, sd, dtype="bf16", max_batch=B)
    for rep in range(2):
        for ov in (False, True):
            ms = _time(lambda: sharded_pair_logits(model, px, ids, mask, overlap=ov), iters=20, warm=3)
            print(f"overlap={ov}: {ms:.3f} ms/step  {B / ms * 1e3:.0f} pairs/s")
    a = sharded_pair_logits(model, px, ids, mask, overlap=False)[0]
    b = sharded_pair_logits(model, px, ids, mask, overlap=True)[0]
    torch.cuda.synchronize()
    print("bitwise equal one-stream vs two-stream:", bool(torch.equal(a, b)))


def sec_e2e():
    cfg = get_config("ViT-B/32")
    sd = W.synthetic_state_dict(cfg, 0)
    B = 256
    px = torch.from_numpy(W.synthetic_pixels(cfg, B, 1)).to(dev)
    ids_np, mask_np = W.synthetic_ids(cfg, B, 2)
    ids, mask = torch.from_numpy(ids_np).to(dev), torch.from_numpy(mask_np).to(dev)
    for dtype in ("bf16", "f32"):
        model = PlipModel(cfg, sd, dtype=dtype, max_batch=B)
        eng = model.engine
        ti = _time(lambda: eng.encode_image(px, True), iters=10, warm=2)
        tt = _time(lambda: eng.encode_text(ids, mask, True), iters=10, warm=2)
        print(f"{dtype}: image tower {ti:.3f} ms ({B / ti * 1e3:.0f} img/s, {B * cfg.image_flops() / ti / 1e9:.1f} TFLOP/s)  "
              f"text tower {tt:.3f} ms ({B / tt * 1e3:.0f} cap/s, {B * cfg.text_flops() / tt / 1e9:.1f} TFLOP/s)  "
              f"pairs/s {B / (ti + tt) * 1e3:.0f}")
        rows = []
        with eng.profile(rows):
            eng.encode_image(px, True)
            eng.encode_text(ids, mask, True)
        rows.sort(key=lambda r: -r["total_ms"])
        tot = sum(r["total_ms"] for r in rows)
        for r in rows:
            tf = f"{r['flops'] / r['total_ms'] / 1e9:8.1f} TF/s" if r["flops"] else f"{r['bytes'] / r['total_ms'] / 1e6:8.1f} GB/s"
            print(f"   {r['name']:48s} calls {r['calls']:4d}  {r['total_ms']:8.3f} ms  {100 * r['total_ms'] / tot:5.1f}%  {tf}")
        model.engine.close()


if __name__ == "__main__":
    t0 = time.time()
    {"gemm": sec_gemm, "attn": sec_attn, "tiny": sec_tiny, "vitb32": sec_vitb32, "gemmbench": sec_gemmbench, "lnbench": sec_lnbench, "latency": sec_latency, "libgemm": sec_libgemm, "tiles": sec_tiles, "parity": sec_parity, "sustain": sec_sustain, "power": sec_power, "cold": sec_cold, "towerswap": sec_towerswap, "e2e": sec_e2e, "gemmone": sec_gemmone, "policy": sec_policy, "ldpad": sec_ldpad, "gemmtrace": sec_gemmtrace,
     "overlap": sec_overlap, "qkvattn": sec_qkvattn, "stepab": sec_stepab, "cumask": sec_cumask, "mixed": sec_mixed, "zeroshot": sec_zeroshot, "latprof": sec_latprof}[sys.argv[1]]()
    print(f"[{sys.argv[1]} done in {time.time() - t0:.1f} s]")
