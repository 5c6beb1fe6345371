"""round 6: default stream + first pool stream (the product) against two dedicated non-default streams, interleaved, five rounds"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from plip_amd import weights as W, engine as E
from plip_amd.config import get_config
from plip_amd.model import PlipModel
from plip_amd.dist import sharded_pair_logits
dev = torch.device("cuda", 0)
cfg = get_config("ViT-B/32")
sd = W.synthetic_state_dict(cfg, 0)
px = torch.from_numpy(W.synthetic_pixels(cfg, 256, seed=1000)).to(dev)
i, m = W.synthetic_ids(cfg, 256, seed=2000)
ids, mask = torch.from_numpy(i).to(dev), torch.from_numpy(m).to(dev)
model = PlipModel(cfg, sd, device=dev, dtype="bf16", max_batch=256)
first = E._SIDE_STREAMS[model.engine.device]
pool = [torch.cuda.Stream(device=dev) for _ in range(8)]
def timed(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
step = lambda: sharded_pair_logits(model, px, ids, mask, overlap=True, equal_shards=True)
cur = torch.cuda.current_stream(dev)
arms = {"default + first": (None, first), "pool[2] + pool[5]": (pool[2], pool[5]), "pool[3] + pool[6]": (pool[3], pool[6]), "pool[1] + pool[3]": (pool[1], pool[3]), "pool[0] + first": (pool[0], first)}
res = {k: [] for k in arms}
for rnd in range(5):
    for name, (main, side) in arms.items():
        E._PAIR_STREAMS[(model.engine.device, (main or cur).cuda_stream)] = side     # overrides Engine.pair_stream's measured choice
        if main is None:
            t = timed(step)
        else:
            main.wait_stream(cur)
            with torch.cuda.stream(main):
                t = timed(step)
            cur.wait_stream(main)
        res[name].append(t)
for name, v in res.items():
    print(f"{name:20s} median {sorted(v)[2]:.3f} ms   rounds {[round(t, 3) for t in v]}")
