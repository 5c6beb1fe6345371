"""round 6: why does the H2D-inclusive step not overlap its copies?  The same double-buffered loop with different copy streams."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from plip_amd import weights as W
from plip_amd.config import get_config
from plip_amd.model import PlipModel
dev = torch.device("cuda", 0)
cfg = get_config("ViT-B/32")
sd = W.synthetic_state_dict(cfg, 0)
B = 256
px = torch.from_numpy(W.synthetic_pixels(cfg, B, seed=1000)).to(dev)
i, m = W.synthetic_ids(cfg, B, seed=2000)
ids, mask = torch.from_numpy(i).to(dev), torch.from_numpy(m).to(dev)
# raw copy rate, nothing else running
h = px.cpu().pin_memory(); d = torch.empty_like(px)
for _ in range(2): d.copy_(h, non_blocking=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): d.copy_(h, non_blocking=True)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print(f"raw pinned H2D of {h.numel() * 4 / 1e6:.0f} MB: {dt * 1e3:.2f} ms = {h.numel() * 4 / dt / 1e9:.1f} GB/s", flush=True)
pool = [torch.cuda.Stream(device=dev) for _ in range(6)]          # streams created BEFORE the engine's side stream
hp = torch.cuda.Stream(device=dev, priority=-1)
model = PlipModel(cfg, sd, device=dev, dtype="bf16", max_batch=B)
for ov in (True, False):
    dt, _ = bench.timed_steps(lambda: bench.sharded_pair_logits(model, px, ids, mask, overlap=ov, equal_shards=True) if hasattr(bench, "sharded_pair_logits") else None, 1, dev, 0) if False else (0, 0)
from plip_amd.dist import sharded_pair_logits
for ov in (True, False):
    dt, _ = bench.timed_steps(lambda: sharded_pair_logits(model, px, ids, mask, overlap=ov, equal_shards=True), 20, dev, 3)
    print(f"resident, overlap={ov}: {dt * 1e3:.3f} ms/step", flush=True)
    for name, st in [("fresh default", None), ("high priority", hp)] + [(f"pool[{k}]", pool[k]) for k in range(6)]:
        r = bench.h2d_inclusive(model, cfg, px, ids, mask, 20, 3, ov, copy=st)
        print(f"  copy stream {name:14s}: fp32 {r['fp32_pixels']['ms_per_step']:.3f} ms ({r['fp32_pixels']['pcie_GBps']} GB/s)   u8 {r['u8_tiles']['ms_per_step']:.3f} ms", flush=True)
