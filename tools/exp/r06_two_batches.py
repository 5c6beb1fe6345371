#!/usr/bin/env python
"""Round 6 experiment: consecutive batches of ONE tower on two HIP streams (two handles: a handle owns one workspace per tower).

The pair step hides a tower's launch boundaries, prologues / epilogues and tail behind the OTHER tower's kernels (4.65 -> 4.30 ms).  A corpus
pass (configs[3]: the image tower + the arg-max head over 489 batches) has no other tower -- but it has the NEXT batch.  Arms, interleaved:
  n lanes   batch k on stream k % n / handle k % n (plipmi_clone of the first: same weights, a workspace each), no cross-stream waits;
            1 lane = what tools/config3_shard.py measured as `resident` through round 5
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
ids, _ = W.synthetic_ids(cfg, 10, seed=7)
cls = A.engine.encode_text(torch.from_numpy(ids), None, normalize=True)
tiles = W.synthetic_tiles(cfg, 4 * B, 11)
pool = [torch.from_numpy(tiles[k * B:(k + 1) * B]).to(dev) for k in range(4)]
main = torch.cuda.current_stream(dev)
engines = [A.engine] + [A.engine.clone() for _ in range(3)]            # plipmi_clone: the same packed weights, a workspace each
streams = [main, A.engine.pair_stream(main)]
for _ in range(16):                                                    # two more streams that overlap with main and with each other
    if len(streams) == 4:
        break
    c = torch.cuda.Stream(device=dev)
    if all(A.engine.streams_overlap(s, c) < 1.5 for s in streams):
        streams.append(c)
print("streams found:", len(streams))


def classify(eng, t):
    img = eng.encode_image_u8(t, normalize=True)
    return eng.logits(img, cls, scale=1.0, want_text=False, want_argmax=True)[2]


def lanes(n):
    def fn():
        out = [None] * NB
        for st in streams[1:n]:
            st.wait_stream(main)
        for k in range(NB):
            with torch.cuda.stream(streams[k % n]):
                out[k] = classify(engines[k % n], pool[k % 4])
        for st in streams[1:n]:
            main.wait_stream(st)
        return out
    return fn


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    return time.perf_counter() - t0, out


arms = {f"{n} lane(s)": lanes(n) for n in range(1, len(streams) + 1)}
ref = torch.cat(timed(arms["1 lane(s)"])[1])
for name, fn in arms.items():
    print(name, "same predictions:", bool(torch.equal(ref, torch.cat(timed(fn)[1]))))
res = {k: [] for k in arms}
for rnd in range(4):
    for name, fn in arms.items():
        res[name].append(timed(fn)[0])
for name, ts in res.items():
    t = sorted(ts)[len(ts) // 2]
    print(f"{name}: {NB * B / t / 1e3:7.1f} k img/s   ({t / NB * 1e3:.3f} ms per batch; rounds {[round(x / NB * 1e3, 3) for x in ts]})")
