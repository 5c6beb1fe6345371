"""round 6: which PAIR of HIP streams carries the two towers?  (streams share hardware queues; profiles/r06_batch_scaling.txt)"""
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
pool = [torch.cuda.Stream(device=dev) for _ in range(8)]
hi = [torch.cuda.Stream(device=dev, priority=-1) for _ in range(2)]
def timed(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
step = lambda: sharded_pair_logits(model, px, ids, mask, overlap=True, equal_shards=True)
names = {id(s): f"pool[{k}]" for k, s in enumerate(pool)}; names.update({id(s): f"hi[{k}]" for k, s in enumerate(hi)})
cur = torch.cuda.current_stream(dev)
res = []
for main in [None] + pool[:4] + hi[:1]:
    for side in pool + hi:
        if side is main: continue
        E._PAIR_STREAMS[(model.engine.device, (main or cur).cuda_stream)] = side     # overrides Engine.pair_stream's measured choice
        if main is None:
            t = timed(step)
        else:
            main.wait_stream(cur)
            with torch.cuda.stream(main):
                t = timed(step)
            cur.wait_stream(main)
        res.append((t, "default" if main is None else names[id(main)], names[id(side)]))
        print(f"vision on {res[-1][1]:8s} text on {res[-1][2]:8s}: {t:.3f} ms", flush=True)
res.sort()
print("best five:", [(round(t, 3), a, b) for t, a, b in res[:5]])
print("worst three:", [(round(t, 3), a, b) for t, a, b in res[-3:]])
E._PAIR_STREAMS.clear()
print("one stream:", round(timed(lambda: sharded_pair_logits(model, px, ids, mask, overlap=False, equal_shards=True)), 3))
