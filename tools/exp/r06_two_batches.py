#!/usr/bin/env python
"""Round 6 experiment: consecutive batches of ONE tower on two HIP streams (two handles: a handle owns one workspace per tower).

The pair step hides a tower's launch boundaries, prologues / epilogues and tail behind the OTHER tower's kernels (4.65 -> 4.30 ms).  A corpus
pass (configs[3]: the image tower + the arg-max head over 489 batches) has no other tower -- but it has the NEXT batch.  Arms, interleaved:
  one     every batch on the default stream, one handle (what tools/config3_shard.py measures as `resident`)
  two     even batches on the default stream / handle A, odd batches on a second stream (Engine.pair_stream) / handle B, no cross-stream waits
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from plip_amd import weights as W  # noqa: E402
from plip_amd.config import get_config  # noqa: E402
from plip_amd.model import PlipModel  # noqa: E402

dev = torch.device("cuda", 0)
cfg = get_config("ViT-B/32")
sd = W.synthetic_state_dict(cfg, 0)
B, NB = 256, int(sys.argv[1]) if len(sys.argv) > 1 else 120
A = PlipModel(cfg, sd, device=dev, dtype="bf16", max_batch=B)
Bm = PlipModel(cfg, sd, device=dev, dtype="bf16", max_batch=B)
ids, _ = W.synthetic_ids(cfg, 10, seed=7)
cls = A.engine.encode_text(torch.from_numpy(ids), None, normalize=True)
tiles = W.synthetic_tiles(cfg, 4 * B, 11)
pool = [torch.from_numpy(tiles[k * B:(k + 1) * B]).to(dev) for k in range(4)]
main = torch.cuda.current_stream(dev)
side = A.engine.pair_stream(main)


def classify(eng, t):
    img = eng.encode_image_u8(t, normalize=True)
    return eng.logits(img, cls, scale=1.0, want_text=False, want_argmax=True)[2]


def one():
    return [classify(A.engine, pool[k % 4]) for k in range(NB)]


def two():
    out = [None] * NB
    side.wait_stream(main)
    for k in range(NB):
        if k & 1:
            with torch.cuda.stream(side):
                out[k] = classify(Bm.engine, pool[k % 4])
        else:
            out[k] = classify(A.engine, pool[k % 4])
    main.wait_stream(side)
    return out


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    return time.perf_counter() - t0, out


ref = torch.cat(timed(one)[1])
got = torch.cat(timed(two)[1])
print("same predictions:", bool(torch.equal(ref, got)))
res = {"one": [], "two": []}
for rnd in range(4):
    for name, fn in (("one", one), ("two", two)):
        res[name].append(timed(fn)[0])
for name, ts in res.items():
    t = sorted(ts)[len(ts) // 2]
    print(f"{name}: {NB * B / t / 1e3:7.1f} k img/s   ({t / NB * 1e3:.3f} ms per batch; rounds {[round(x / NB * 1e3, 3) for x in ts]})")
