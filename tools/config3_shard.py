#!/usr/bin/env python
"""BASELINE.json configs[3] -- "1M-pair synthetic corpus batch-sharded across 8 x MI355X, zero-shot top-1 over 10 class
prompts" -- ONE rank's share at its real size on ONE GPU: 125 000 images = 489 batches of 256 through
``plipmi_encode_image_u8`` + the arg-max head (reproducibility/evaluation/zero_shot/zero_shot.py:12-13), class prompts
replicated.  Three numbers (the scaling risk of this config is the host feed, not the collective):

  resident   tiles already in HBM (a pool of distinct uint8 batches, cycled)           -> the GPU-side rate
  h2d        tiles start in pinned host memory and cross PCIe through the double-buffer pipeline (plip_amd/pipeline.py),
             copy stream overlapped with the towers                                    -> the H2D-inclusive rate
  agreement  scores, top-1 and class ordering of the engine vs HF CLIPModel on the 512-tile sample of
             tests/golden/config3_zero_shot.npz (oracle/make_config3_fixture.py): structured synthetic tiles and ten class
             prompts CHOSEN so that the classes are populated (nine of ten hold >= 4 % of the sample) -- on pure-noise tiles
             and arbitrary prompts every image lands in one class and "top-1 agreement" says nothing

    python tools/config3_shard.py [--images 125000] [--pool 8] [--dtype bf16|f16] > gpurun_out/config3_shard.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from plip_amd import weights as W  # noqa: E402
from plip_amd.config import get_config  # noqa: E402
from plip_amd.model import PlipModel  # noqa: E402
from plip_amd.pipeline import run_batches  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=125000)
    ap.add_argument("--classes", type=int, default=10)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--pool", type=int, default=8, help="distinct synthetic batches (cycled)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16"])
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = get_config("ViT-B/32")
    sd = W.synthetic_state_dict(cfg, 0)
    model = PlipModel(cfg, sd, device=dev, dtype=args.dtype, max_batch=args.batch)
    eng = model.engine
    gold = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "config3_zero_shot.npz"))
    tile_seed, prompt_seed, weight_seed, n_sample, n_pool = (int(v) for v in gold["seeds"])
    assert weight_seed == 0 and args.classes == gold["prompts"].shape[0]
    prompts = gold["prompts"]                                            # stand-ins for "An H&E image patch of <class>."
    class_emb = eng.encode_text(torch.from_numpy(prompts), None, normalize=True)
    B, n_px = args.batch, cfg.image_size
    tiles = W.synthetic_tiles(cfg, args.pool * B, tile_seed)             # rank 0's seed; first 512 = the fixture's sample
    host_pool = [torch.from_numpy(tiles[k * B:(k + 1) * B]).pin_memory() for k in range(args.pool)]
    dev_pool = [t.to(dev) for t in host_pool]
    nb = (args.images + B - 1) // B
    sizes = [min(B, args.images - k * B) for k in range(nb)]

    def classify(tiles, e=eng):
        img = e.encode_image_u8(tiles, normalize=True)
        return e.logits(img, class_emb, scale=1.0, want_text=False, want_argmax=True)[2]

    def scores(tiles):
        img = eng.encode_image_u8(tiles, normalize=True)
        return eng.logits(img, class_emb, scale=1.0, want_text=False)[0]

    for k in range(3):
        classify(dev_pool[k % args.pool])
    torch.cuda.synchronize()
    # ---- resident: consecutive batches on two lanes (the engine + a clone on a second stream, Engine.lane_loop) = the product's loop;
    #      `one_lane` = every batch on the one engine and stream (rounds 1-5) ---------------------------------------------------
    def resident(two_lanes):
        eng.use_lanes = two_lanes
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with eng.lane_loop() as run:
            preds = [run(lambda e, k=k: classify(dev_pool[k % args.pool][:sizes[k]], e)) for k in range(nb)]
        torch.cuda.synchronize()
        return time.perf_counter() - t0, torch.cat(preds).cpu().numpy()

    # ---- H2D-inclusive: pinned host batches -> copy stream -> towers (double buffered) ----------------------------------
    items = list(range(nb))

    def h2d(two_lanes):
        eng.use_lanes = two_lanes
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outs = run_batches(items, 1, None, lambda tag, t, e=eng: classify(t, e), device=dev, num_workers=1,
                           prepare_batch=lambda chunk, pool: ("tiles", host_pool[chunk[0] % args.pool][:sizes[chunk[0]]]),
                           lanes=eng.lanes() if two_lanes else None)
        torch.cuda.synchronize()
        return time.perf_counter() - t0, torch.cat(outs).cpu().numpy()

    resident(True)                                   # the clone's workspace and the second stream exist before anything is timed
    dt_res1, pred_res1 = resident(False)
    dt_res, pred_res = resident(True)
    dt_h2d1, pred_h2d1 = h2d(False)
    dt_h2d, pred_h2d = h2d(True)
    eng.use_lanes = True
    assert np.array_equal(pred_res, pred_h2d) and np.array_equal(pred_res, pred_res1) and np.array_equal(pred_res, pred_h2d1), \
        "resident / H2D / one-lane / two-lane paths disagree"
    # ---- agreement with the reference's head on the fixture's sample (HF scores stored in the fixture) ---------------------------
    ns = min(n_sample, args.pool * B, args.images)
    sim = gold["scores"][:ns]
    ref_pred = gold["argmax"][:ns]
    tol = 1e-3 if args.dtype == "bf16" else 2.5e-4
    gaps_all = np.diff(np.sort(sim, axis=1), axis=1)
    clear = gaps_all[:, -1] > 2 * tol                                     # winner ahead by 2x the engine's cosine tolerance
    got = pred_res[:ns]
    sim_gpu = torch.cat([scores(dev_pool[k][:min(B, ns - k * B)]) for k in range((ns + B - 1) // B)]).cpu().numpy()
    order_ok = (np.argsort(-sim_gpu, axis=1, kind="stable") == np.argsort(-sim, axis=1, kind="stable")).all(axis=1)
    gaps = gaps_all.min(axis=1)
    ref_kind = "HF transformers CLIPModel (CPU fp32), tests/golden/config3_zero_shot.npz"
    res = {
        "config": f"BASELINE.json configs[3], one rank's shard on one MI355X: ViT-B/32 {args.dtype}, structured synthetic uint8 224x224 tiles, "
                  f"{args.images} images = {nb} batches of {B}, {args.classes} class prompts replicated, arg-max head",
        "device": eng.device_name,
        "resident": {"images_per_s": round(args.images / dt_res, 1), "seconds": round(dt_res, 3),
                     "one_lane_images_per_s": round(args.images / dt_res1, 1),
                     "note": "tiles resident in HBM (pool of distinct batches, cycled); consecutive batches alternate between the engine and "
                             "a clone on a second stream (Engine.lane_loop) -- one_lane: every batch on one engine and stream, same predictions"},
        "h2d_inclusive": {"images_per_s": round(args.images / dt_h2d, 1), "seconds": round(dt_h2d, 3),
                          "one_lane_images_per_s": round(args.images / dt_h2d1, 1),
                          "bytes_per_image": 3 * n_px * n_px,
                          "pcie_GBps": round(args.images * 3 * n_px * n_px / dt_h2d / 1e9, 2),
                          "note": "tiles in pinned host memory -> copy stream -> towers, double buffered (plip_amd/pipeline.py)"},
        "projected_8gpu_images_per_s": round(8 * args.images / dt_h2d, 1),
        "top1_agreement": {"sample": int(ns), "all_rows": float((got == ref_pred).mean()),
                           "clear_rows": float((got[clear] == ref_pred[clear]).mean()) if clear.any() else None,
                           "n_clear_rows": int(clear.sum()), "scores_max_abs_err": float(np.abs(sim_gpu - sim).max()),
                           "full_class_ordering_agreement": float(order_ok.mean()),
                           "full_class_ordering_agreement_where_all_gaps_exceed_2tol": float(order_ok[gaps > 2 * tol].mean()) if (gaps > 2 * tol).any() else None,
                           "rows_with_all_gaps_above_2tol": int((gaps > 2 * tol).sum()), "tolerance": tol, "vs": ref_kind},
        "class_histogram_of_the_sample": {"engine": np.bincount(got, minlength=args.classes).tolist(),
                                          "hf": np.bincount(ref_pred, minlength=args.classes).tolist()},
        "class_histogram_of_the_shard": np.bincount(pred_res, minlength=args.classes).tolist(),
    }
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
