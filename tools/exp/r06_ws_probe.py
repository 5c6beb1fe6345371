"""round 6: does a bs = 256 step run slower on an engine whose workspace is sized for 512 / 1024?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from plip_amd import weights as W
from plip_amd.config import get_config
from plip_amd.model import PlipModel
from plip_amd.dist import sharded_pair_logits
dev = torch.device("cuda", 0)
cfg = get_config("ViT-B/32")
sd = W.synthetic_state_dict(cfg, 0)
g = torch.Generator(device=dev).manual_seed(5)
px = torch.randn((1024, 3, 224, 224), generator=g, device=dev)
i, m = W.synthetic_ids(cfg, 1024, seed=2000)
ids, mask = torch.from_numpy(i).to(dev), torch.from_numpy(m).to(dev)
def timed(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
engines = {mb: PlipModel(cfg, sd, device=dev, dtype="bf16", max_batch=mb, pass_batch=-1) for mb in (256, 512, 1024)}
for rep in range(2):
    for mb, mdl in engines.items():
        same = timed(lambda: sharded_pair_logits(mdl, px[:256], ids[:256], mask[:256], overlap=True, equal_shards=True))
        k = [0]
        def rot():
            a = (k[0] % 4) * 256; k[0] += 1
            return sharded_pair_logits(mdl, px[a:a + 256], ids[a:a + 256], mask[a:a + 256], overlap=True, equal_shards=True)
        rotating = timed(rot)
        print(f"engine max_batch {mb:5d}: bs 256 step, same inputs every step {same:.3f} ms   inputs rotating over 4 batches {rotating:.3f} ms", flush=True)

print("--- B = 512 on one engine (max_batch 512): ways to run it", flush=True)
for e in engines.values():
    e.engine.close()
auto = PlipModel(cfg, sd, device=dev, dtype="bf16", max_batch=512)            # pass_batch automatic (256)
one = PlipModel(cfg, sd, device=dev, dtype="bf16", max_batch=512, pass_batch=-1)
P, I, M = px[:512], ids[:512], mask[:512]
for rep in range(2):
    t_a = timed(lambda: sharded_pair_logits(auto, P, I, M, overlap=True, equal_shards=True))
    t_b = timed(lambda: [sharded_pair_logits(one, P[a:a + 256], I[a:a + 256], M[a:a + 256], overlap=True, equal_shards=True) for a in (0, 256)])
    t_c = timed(lambda: sharded_pair_logits(one, P, I, M, overlap=True, equal_shards=True))
    t_d = timed(lambda: auto.engine.encode_pair(P, I, M, normalize=True, overlap=True))
    t_e = timed(lambda: [one.engine.encode_pair(P[a:a + 256], I[a:a + 256], M[a:a + 256], normalize=True, overlap=True) for a in (0, 256)])
    print(f"passes inside encode_pair {t_a:.3f} ms | two bs-256 steps {t_b:.3f} ms | one pass {t_c:.3f} ms | encode_pair only: passes {t_d:.3f}, two calls {t_e:.3f} ms", flush=True)
