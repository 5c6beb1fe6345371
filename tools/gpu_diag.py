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
from plip_amd.engine import gemm_nt, gemm_variants  # noqa: E402
from plip_amd.model import PlipModel  # noqa: E402

dev = torch.device("cuda:0")


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
    from plip_amd.engine import attention
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
    from plip_amd.engine import gemm_nt_ln
    shapes = [("v.qkv", 12800, 2304, 768, 0), ("v.fc1", 12800, 3072, 768, 1), ("v.out", 12800, 768, 768, 2),
              ("v.fc2", 12800, 768, 3072, 2), ("t.qkv", 19712, 1536, 512, 0), ("t.fc1", 19712, 2048, 512, 1),
              ("t.out", 19712, 512, 512, 2), ("t.fc2", 19712, 512, 2048, 2)]
    g = torch.Generator().manual_seed(0)
    for name, M, N, K, epi in shapes:
        a = torch.randn(M, K, generator=g).to(dev).bfloat16()
        w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev).bfloat16()
        bias = torch.randn(N, generator=g).to(dev)
        row = []
        for v in (36, 37, 42, 41):
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
    plip.py:90-91): eager launches vs hipGraph replay, wall time per call with a host sync (what a caller sees)."""
    import time
    cfg = get_config("ViT-B/32")
    sd = W.synthetic_state_dict(cfg, 0)
    model = PlipModel(cfg, sd, dtype="bf16", max_batch=32)
    eng = model.engine
    for B in (1, 8, 32):
        px = torch.from_numpy(W.synthetic_pixels(cfg, B, 1)).to(dev)
        ids = torch.from_numpy(W.synthetic_ids(cfg, B, 2)[0]).to(dev)
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
            print(f"B={B:3d} {mode:5s}: " + "  ".join(f"{k} {v:.3f} ms" for k, v in res.items()))
    model.engine.close()


def sec_prio():
    """Two-stream step with the vision tower on a high-priority stream vs the default arrangement (in-process, interleaved)."""
    from plip_amd.dist import sharded_pair_logits
    cfg = get_config("ViT-B/32")
    sd = W.synthetic_state_dict(cfg, 0)
    B = 256
    px = torch.from_numpy(W.synthetic_pixels(cfg, B, 1)).to(dev)
    ids_np, mask_np = W.synthetic_ids(cfg, B, 2)
    ids, mask = torch.from_numpy(ids_np).to(dev), torch.from_numpy(mask_np).to(dev)
    model = PlipModel(cfg, sd, dtype="bf16", max_batch=B)
    res = {False: [], True: []}
    for rep in range(5):
        for prio in (False, True):
            model.engine.pair_vision_priority = prio
            ms = _time(lambda: sharded_pair_logits(model, px, ids, mask, overlap=True), iters=10, warm=2)
            if rep:
                res[prio].append(ms)
    for prio in (False, True):
        print(f"vision tower on a high-priority stream = {prio}: two streams {np.median(res[prio]):6.3f} ms (min {min(res[prio]):6.3f})")
    model.engine.close()


def sec_fourstream():
    """Experiment: the bs=256 step as FOUR independent chains (each tower's batch cut in two halves, on four HIP streams, two
    handles sharing nothing) against the shipped two-stream arrangement -- does finer interleaving of kernels from
    independent chains hide more of the GEMMs' prologue / epilogue phases?"""
    cfg = get_config("ViT-B/32")
    sd = W.synthetic_state_dict(cfg, 0)
    B = 256
    px = torch.from_numpy(W.synthetic_pixels(cfg, B, 1)).to(dev)
    ids_np, mask_np = W.synthetic_ids(cfg, B, 2)
    ids, mask = torch.from_numpy(ids_np).to(dev), torch.from_numpy(mask_np).to(dev)
    mA = PlipModel(cfg, sd, dtype="bf16", max_batch=B)
    mB = PlipModel(cfg, sd, dtype="bf16", max_batch=B)
    s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    h = B // 2

    def two():
        return mA.engine.encode_pair(px, ids, mask, True, overlap=True)

    def four():
        main = torch.cuda.current_stream(dev)
        s1.wait_stream(main); s2.wait_stream(main)
        with torch.cuda.stream(s1):
            a = mA.engine.encode_pair(px[:h], ids[:h], mask[:h], True, overlap=True)
        with torch.cuda.stream(s2):
            b = mB.engine.encode_pair(px[h:], ids[h:], mask[h:], True, overlap=True)
        main.wait_stream(s1); main.wait_stream(s2)
        return a, b

    ia, ta = two()
    (i1, t1), (i2, t2) = four()
    torch.cuda.synchronize()
    print("four-chain vs two-stream embeddings max diff", (torch.cat([i1, i2]) - ia).abs().max().item(), (torch.cat([t1, t2]) - ta).abs().max().item())
    res = {"two": [], "four": []}
    for rep in range(5):
        for name, fn in (("two", two), ("four", four)):
            ms = _time(fn, iters=10, warm=2)
            if rep:
                res[name].append(ms)
    for name in res:
        print(f"{name:5s} streams/chains: both towers {np.median(res[name]):6.3f} ms (min {min(res[name]):6.3f})")
    for pol in (0, 3):
        mA.engine.pair_policy = pol; mB.engine.pair_policy = pol
        print(f"  four chains, tile policy {pol}: {_time(four, iters=10, warm=2):6.3f} ms")
    mA.engine.close(); mB.engine.close()


def sec_libgemm():
    """Calibration only (never used by the product): what the vendor GEMM library (hipBLASLt/rocBLAS behind
    torch.nn.functional.linear) reaches on the production shapes -- an external yardstick for gemm_nt."""
    import torch.nn.functional as F
    shapes = [("v.qkv", 12800, 2304, 768), ("v.out", 12800, 768, 768), ("v.fc1", 12800, 3072, 768),
              ("v.fc2", 12800, 768, 3072), ("t.qkv", 19712, 1536, 512), ("t.out", 19712, 512, 512),
              ("t.fc1", 19712, 2048, 512), ("t.fc2", 19712, 512, 2048), ("big", 8192, 8192, 8192)]
    g = torch.Generator().manual_seed(0)
    for name, M, N, K in shapes:
        a = torch.randn(M, K, generator=g).to(dev).to(torch.bfloat16)
        w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev).to(torch.bfloat16)
        bias = torch.randn(N, generator=g).to(dev).to(torch.bfloat16)
        out = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
        ms_lib = _time(lambda: F.linear(a, w, bias), iters=20)
        ms_mm = _time(lambda: torch.mm(a, w.t(), out=out), iters=20)
        if K % 64 == 0 and N % 256 == 0:
            ms_own = _time(lambda: gemm_nt(a, w, bias.float(), epilogue=0, out=out), iters=20)
        else:
            ms_own = float("nan")
        f = 2.0 * M * N * K / 1e9
        print(f"{name:6s} {M:6d}x{N:5d}x{K:5d}: F.linear+bias {f / ms_lib:7.1f} TF/s   torch.mm {f / ms_mm:7.1f} TF/s   gemm_nt(bias) {f / ms_own:7.1f} TF/s")


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


def sec_fp8():
    """EXPERIMENTAL fp8 (e4m3fn) GEMM test hook: exactness against fp64 of the same fp8 operands, then TFLOP/s on
    the production QKV / fc1 shapes beside the bf16 kernels."""
    g = torch.Generator().manual_seed(0)
    for (M, N, K, epi) in ((300, 256, 128, 0), (515, 512, 768, 0), (1000, 768, 3072, 1)):
        a = (torch.randn(M, K, generator=g)).to(dev).to(torch.float8_e4m3fn)
        w = (torch.randn(N, K, generator=g) / K ** 0.5 * 4).to(dev).to(torch.float8_e4m3fn)
        bias = torch.randn(N, generator=g).to(dev)
        ref = a.float().double() @ w.float().double().T + bias.double()
        if epi == 1:
            ref = ref * torch.sigmoid(1.702 * ref)
        for v in range(4):
            y = gemm_nt(a, w, bias, epilogue=epi, variant=v)
            torch.cuda.synchronize()
            err = (y.double() - ref).abs().max().item()
            print(f"fp8 check {M}x{N}x{K} epi{epi} variant {v}: max err {err:.3e} (|ref| max {ref.abs().max().item():.2f}, bf16 output rounding ~{ref.abs().max().item() * 2 ** -9:.1e})")
    shapes = [("v.qkv", 12800, 2304, 768, 0), ("v.fc1", 12800, 3072, 768, 1), ("v.fc2*", 12800, 768, 3072, 0),
              ("t.qkv", 19712, 1536, 512, 0), ("t.fc1", 19712, 2048, 512, 1), ("big", 8192, 8192, 8192, 0)]
    for name, M, N, K, epi in shapes:
        a8 = torch.randn(M, K, generator=g).to(dev).to(torch.float8_e4m3fn)
        w8 = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev).to(torch.float8_e4m3fn)
        a16, w16 = a8.to(torch.bfloat16), w8.to(torch.bfloat16)
        bias = torch.randn(N, generator=g).to(dev)
        out = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
        row = []
        for v in range(4):
            ms = _time(lambda: gemm_nt(a8, w8, bias, epilogue=epi, variant=v, out=out), iters=20)
            row.append(f"fp8 v{v} {2.0 * M * N * K / ms / 1e9:7.1f}")
        ms = _time(lambda: gemm_nt(a16, w16, bias, epilogue=epi, out=out), iters=20)
        print(f"{name:7s} {M}x{N}x{K} epi{epi}: " + "  ".join(row) + f"   bf16 auto {2.0 * M * N * K / ms / 1e9:7.1f} TFLOP/s")


def sec_fp8w():
    """EXPERIMENTAL fp8-weights engine (QKV + fc1 in fp8): error against the CPU oracle and tower throughput."""
    from oracle import clip_oracle as O
    for arch, B in (("tiny-w256", 6), ("ViT-B/32", 4)):
        cfg = get_config(arch)
        sd = W.synthetic_state_dict(cfg, 0)
        px = W.synthetic_pixels(cfg, B, 1)
        ids, mask = W.synthetic_ids(cfg, B, 2)
        ref = O.clip_forward(px, ids, sd, cfg, mask)
        scale = float(np.exp(np.float64(sd["logit_scale"])))
        for dtype in ("bf16", "fp8"):
            model = PlipModel(cfg, sd, dtype=dtype, max_batch=8)
            out = model(input_ids=torch.from_numpy(ids), pixel_values=torch.from_numpy(px), attention_mask=torch.from_numpy(mask))
            cos = np.abs(out.logits_per_image.cpu().numpy() - ref["logits_per_image"]).max() / scale
            emb = max(np.abs(out.image_embeds.cpu().numpy() - ref["image_embeds"]).max(),
                      np.abs(out.text_embeds.cpu().numpy() - ref["text_embeds"]).max())
            print(f"{arch:10s} {dtype:5s}: cosine-logit max-abs-err {cos:.3e}   embedding component max-abs-err {emb:.3e}")
            model.engine.close()
    for arch, B in (("ViT-B/32", 256), ("ViT-L/14@336px", 32)):
        cfg = get_config(arch)
        sd = W.synthetic_state_dict(cfg, 0)
        px = torch.from_numpy(W.synthetic_pixels(cfg, B, 1)).to(dev)
        ids_np, mask_np = W.synthetic_ids(cfg, B, 2)
        ids = torch.from_numpy(ids_np).to(dev)
        for dtype in ("bf16", "fp8"):
            model = PlipModel(cfg, sd, dtype=dtype, max_batch=B)
            ms_i = _time(lambda: model.engine.encode_image(px), iters=5, warm=2)
            ms_t = _time(lambda: model.engine.encode_text(ids, None), iters=5, warm=2)
            print(f"{arch:16s} B={B} {dtype:5s}: image {ms_i:7.2f} ms ({B / ms_i * 1e3:8.0f} img/s, {cfg.image_flops() * B / ms_i / 1e9:6.1f} TF/s)   "
                  f"text {ms_t:6.2f} ms ({B / ms_t * 1e3:8.0f} cap/s)")
            if dtype == "fp8":
                rows = []
                with model.engine.profile(rows):
                    model.engine.encode_image(px)
                    torch.cuda.synchronize()
                tot = sum(r["total_ms"] for r in rows)
                for r in sorted(rows, key=lambda r: -r["total_ms"])[:6]:
                    tf = f"{r['flops'] / r['total_ms'] / 1e9:7.1f} TF/s" if r["flops"] else ""
                    print(f"      {r['name']:58s} {r['total_ms']:8.3f} ms {100 * r['total_ms'] / tot:5.1f}%  {tf}")
            model.engine.close()
            del model
            torch.cuda.empty_cache()


def sec_ldpad():
    """Does padding the leading dimension (rows no longer a multiple of 2 KB apart) change the fill rate?"""
    from plip_amd.engine import gemm_nt_ld
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
    kt = max(1, int(t[0, 6]) - 1)
    print("  start offset :", qu(start_us))
    print("  prologue     :", qc(pro))
    print("  main loop    :", qc(loop), f"  ({np.median(loop) / kt:.0f} cycles per k-tile, {kt} tiles)")
    print("  epilogue     :", qc(epi_t))
    print("  lifetime     :", qc(life), "  wall:", qu(real_life_us))
    print("  end          :", qu(end_us))
    print("  workgroups per XCC:", np.bincount(xcc, minlength=8).tolist())
    return
    m0 = xcc == 0
    first_end = end[m0].min()
    print(f"  XCC0: {m0.sum()} workgroups, {(start[m0] < first_end).sum()} started before its first one ended")
    idx = np.nonzero(m0)[0]
    order = idx[np.argsort(start[idx])]
    for i in list(order[:4]) + list(order[len(order) // 2: len(order) // 2 + 3]) + list(order[-3:]):
        print(f"    wg {i:4d} tile {t[i, 4]:4d} hw_id {t[i, 5] & 0xffffffff:08x}: start {start[i] / tick_us:7.2f} "
              f"pro {pro[i] / tick_us:6.2f} loop {loop[i] / tick_us:7.2f} epi {epi_t[i] / tick_us:6.2f} end {end[i] / tick_us:7.2f} us")


def sec_policy():
    """In-process interleaved A/B of GEMM tile policies (forced variant vs the cost model), one and two streams."""
    from plip_amd import _lib
    from plip_amd.dist import sharded_pair_logits
    lib = _lib.load()
    cfg = get_config("ViT-B/32")
    sd = W.synthetic_state_dict(cfg, 0)
    B = 256
    px = torch.from_numpy(W.synthetic_pixels(cfg, B, 1)).to(dev)
    ids_np, mask_np = W.synthetic_ids(cfg, B, 2)
    ids, mask = torch.from_numpy(ids_np).to(dev), torch.from_numpy(mask_np).to(dev)
    model = PlipModel(cfg, sd, dtype="bf16", max_batch=B)
    pols = [-10, -1, -12, -13]       # one stream: always the cost model; two streams: pair policy 0 (cost model) / 1 / 2 / 3
    res = {(p, ov): [] for p in pols for ov in (False, True)}
    for rep in range(4):
        for pol in pols:
            lib.plipmi_set_gemm_variant(pol if pol >= 0 else -1)
            model.engine.pair_policy = {-10: 0, -12: 2, -13: 3}.get(pol, 1)
            for ov in (False, True):
                ms = _time(lambda: sharded_pair_logits(model, px, ids, mask, overlap=ov), iters=10, warm=2)
                if rep:                      # rep 0 = warm-up of clocks / caches
                    res[(pol, ov)].append(ms)
    lib.plipmi_set_gemm_variant(-1)
    names = gemm_variants()
    for pol in pols:
        a, b = res[(pol, False)], res[(pol, True)]
        label = {-10: "cost model/pair policy 0", -1: "cost model/pair policy 1", -12: "cost model/pair policy 2",
                 -13: "cost model/pair policy 3"}.get(pol) or names[pol]
        print(f"policy {label:34s} one stream {np.median(a):6.3f} ms (min {min(a):6.3f})   "
              f"two streams {np.median(b):6.3f} ms (min {min(b):6.3f})  -> {B / np.median(b) * 1e3:7.0f} pairs/s")


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
    # experiment: more kernel-level blending -- two engines, each on half the batch
    m2 = PlipModel(cfg, sd, dtype="bf16", max_batch=B)
    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
    h = B // 2

    def four_streams():
        main = torch.cuda.current_stream()
        sA.wait_stream(main); sB.wait_stream(main)
        with torch.cuda.stream(sA):
            model.engine.encode_pair(px[:h], ids[:h], mask[:h])
        with torch.cuda.stream(sB):
            m2.engine.encode_pair(px[h:], ids[h:], mask[h:])
        main.wait_stream(sA); main.wait_stream(sB)

    def two_streams_swapped():
        main = torch.cuda.current_stream()
        sA.wait_stream(main); sB.wait_stream(main)
        model.engine._set_policy(1)
        with torch.cuda.stream(sA):
            model.engine.encode_image(px[:h]); model.engine.encode_text(ids[:h], mask[:h])
        with torch.cuda.stream(sB):
            m2.engine.encode_text(ids[h:], mask[h:]); m2.engine.encode_image(px[h:])
        model.engine._set_policy(0)
        main.wait_stream(sA); main.wait_stream(sB)

    for rep in range(2):
        for name, fn in (("four streams (2 engines x half batch)", four_streams), ("two streams, half batches, swapped tower order", two_streams_swapped)):
            ms = _time(fn, iters=20, warm=3)
            print(f"{name}: {ms:.3f} ms/step  {B / ms * 1e3:.0f} pairs/s")
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
    {"gemm": sec_gemm, "attn": sec_attn, "tiny": sec_tiny, "vitb32": sec_vitb32, "gemmbench": sec_gemmbench, "lnbench": sec_lnbench, "latency": sec_latency, "prio": sec_prio, "fourstream": sec_fourstream, "libgemm": sec_libgemm, "fp8": sec_fp8, "fp8w": sec_fp8w, "towerswap": sec_towerswap, "e2e": sec_e2e, "gemmone": sec_gemmone, "policy": sec_policy, "ldpad": sec_ldpad, "gemmtrace": sec_gemmtrace,
     "overlap": sec_overlap}[sys.argv[1]]()
    print(f"[{sys.argv[1]} done in {time.time() - t0:.1f} s]")
