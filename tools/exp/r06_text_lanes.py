#!/usr/bin/env python
"""Round 6: the TEXT tower over many batches of 256 captions (PLIP.encode_text's loop, plip.py:64-71) on one lane and on the product's two
lanes (Engine.lane_loop: the engine + a plipmi_clone of it on a second stream); padded and packed captions."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from plip_amd import weights as W  # noqa: E402
from plip_amd.config import get_config  # noqa: E402
from plip_amd.model import PlipModel  # noqa: E402

dev = torch.device("cuda", 0)
cfg = get_config("ViT-B/32")
B, NB = 256, 60
eng = PlipModel(cfg, W.synthetic_state_dict(cfg, 0), device=dev, dtype="bf16", max_batch=B).engine
pool = []
for k in range(4):
    i, m = W.synthetic_ids(cfg, B, seed=100 + k)
    pool.append((torch.from_numpy(i).to(dev), torch.from_numpy(m).to(dev)))


def walk(two):
    eng.use_lanes = two
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with eng.lane_loop() as run:
        outs = [run(lambda e, k=k: e.encode_text(pool[k % 4][0], pool[k % 4][1], True)) for k in range(NB)]
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / NB, outs


for packed in (False, True):
    eng.set_text_packing(packed)
    a, b = walk(False)[1], walk(True)[1]
    same = all(torch.equal(x, y) for x, y in zip(a, b))
    res = {False: [], True: []}
    for _ in range(3):
        for two in (False, True):
            res[two].append(walk(two)[0])
    t1, t2 = sorted(res[False])[1], sorted(res[True])[1]
    print(f"captions {'packed' if packed else 'padded'}: one lane {B / t1 / 1e3:6.1f} k captions/s ({t1 * 1e3:.3f} ms per batch)   two lanes {B / t2 / 1e3:6.1f} k captions/s "
          f"({t2 * 1e3:.3f} ms)   {100 * (t1 / t2 - 1):+.1f} %   same bits: {same}")
