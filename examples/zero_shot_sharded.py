#!/usr/bin/env python
"""BASELINE.json config 4 in miniature: a synthetic corpus batch-sharded across the GPUs of one node,
class prompts replicated, zero-shot top-1 per image, predictions gathered with RCCL.

    python examples/zero_shot_sharded.py --images 4096 --classes 10                      # one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 examples/zero_shot_sharded.py --images 1000000 --classes 10   # 8 x MI355X

Mirrors reproducibility/scripts/zero_shot_evaluation.py:36-71 + evaluation/zero_shot/zero_shot.py:12-13
(normalised image @ text.T, argmax) with the embedding forward on the MI355X engine.
"""
import argparse
import contextlib
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from plip_amd import weights as W  # noqa: E402
from plip_amd.config import get_config  # noqa: E402
from plip_amd.dist import all_gather_rows, shard_bounds  # noqa: E402
from plip_amd.model import PlipModel  # noqa: E402


def main(argv=None, model_factory=None):
    """``model_factory`` + ``--backend gloo``: the tests' host-only rehearsal of the multi-rank control flow with a stub
    engine (tests/test_bench_ranks.py); the example itself runs on MI355X GPUs over RCCL."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=4096)
    ap.add_argument("--classes", type=int, default=10)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--checkpoint", default=None, help="local HF dir or OpenAI .pt; default: synthetic weights")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help=argparse.SUPPRESS)
    args = ap.parse_args(argv)
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    rehearsal = args.backend == "gloo"
    if rehearsal != (model_factory is not None):
        raise SystemExit("--backend gloo and a stub model factory go together (tests only)")
    if rehearsal:
        dev = torch.device("cpu")
    else:
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    sync = (lambda: torch.cuda.synchronize()) if dev.type == "cuda" else (lambda: None)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    cfg = get_config("ViT-B/32")
    if model_factory is not None:
        model = model_factory(cfg, None, device=dev, dtype="bf16", max_batch=args.batch)
    elif args.checkpoint:
        model = PlipModel.from_pretrained(args.checkpoint, device=dev, max_batch=args.batch)
    else:
        model = PlipModel(cfg, W.synthetic_state_dict(cfg, 0), device=dev, max_batch=args.batch)
    eng = model.engine
    prompts, _ = W.synthetic_ids(cfg, args.classes, seed=7)           # stand-in for "An H&E image patch of [class]."
    class_emb = eng.encode_text(torch.from_numpy(prompts), None, normalize=True)   # [C,512], replicated on every rank
    lo, hi = shard_bounds(args.images, rank, world)
    preds = []
    sync()
    t0 = time.perf_counter()
    g = torch.Generator(device="cpu").manual_seed(1000 + rank)
    def classify(e, tiles):
        img = e.encode_image_u8(tiles, normalize=True)                 # u8 tiles -> fused normalise + towers
        return e.logits(img, class_emb, scale=1.0, want_text=False, want_argmax=True)[2]

    # consecutive batches alternate between the engine and a clone (same weights) on a second stream: one batch's launch boundaries
    # and pooled tail run under the next batch's GEMMs (Engine.lane_loop; +8 % on the resident shard)
    loop = getattr(eng, "lane_loop", None)
    with (loop() if loop is not None else contextlib.nullcontext(lambda fn: fn(eng))) as run:
        for s in range(lo, hi, args.batch):
            n = min(args.batch, hi - s)
            tiles = torch.randint(0, 256, (n, cfg.image_size, cfg.image_size, 3), dtype=torch.uint8, generator=g)
            preds.append(run(lambda e, tiles=tiles: classify(e, tiles)))
    local_pred = torch.cat(preds) if preds else torch.empty(0, dtype=torch.int32, device=dev)
    all_pred = all_gather_rows(local_pred)                              # only 4 bytes per image cross xGMI
    sync()
    dt = time.perf_counter() - t0
    if rank == 0:
        hist = torch.bincount(all_pred.long(), minlength=args.classes).tolist()
        print(f"{args.images} images on {world} GPU(s): {args.images / dt:.0f} img/s incl. host tile synthesis; class histogram {hist}")
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
