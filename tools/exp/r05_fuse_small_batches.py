"""Text tower alone (ViT-B/32, bf16, eager): the fused q/k/v + attention kernel against the two kernels over the batch size -- where
the product rule qkv_attention_pays() switches.  usage: python tools/exp/r05_fuse_small_batches.py"""
import sys, time, torch, numpy as np
sys.path.insert(0, '/root/repo')
from plip_amd import _lib, weights as W
from plip_amd.config import get_config
from plip_amd.model import PlipModel
lib = _lib.load()
cfg = get_config("ViT-B/32"); sd = W.synthetic_state_dict(cfg, 0)
model = PlipModel(cfg, sd, dtype="bf16", max_batch=256, graph_batch=0)
dev = model.device
def t(fn, it=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e3
for B in (1, 4, 8, 16, 32, 48, 64, 96, 128, 256):
    ids_np, mask_np = W.synthetic_ids(cfg, B, 2000)
    ids, mask = torch.from_numpy(ids_np).to(dev), torch.from_numpy(mask_np).to(dev)
    r = {}
    for rep in range(2):
        for mode in (3000, 3002, 3001):       # two kernels / fused wherever it applies / the product rule (fused when the batch fills the chip)
            lib.plipmi_test_fused_qkv_attention(mode - 3000)
            ms = t(lambda: model.engine.encode_text(ids, mask, normalize=True))
            r[mode] = min(r.get(mode, 1e9), ms)
    lib.plipmi_test_reset_hooks()
    print(f"text tower B={B:4d}: two kernels {r[3000]:7.3f} ms   fused {r[3002]:7.3f} ms ({(r[3002]/r[3000]-1)*100:+.1f} %)   product rule {r[3001]:7.3f} ms")
