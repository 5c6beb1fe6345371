"""round 6 experiment: the bs = 256 pair step as FOUR concurrent half-batch towers (two engines of max_batch 128, two streams each)
against the shipped two-stream step -- does finer interleaving of unlike phases (one kernel's epilogue beside another's K loop) pay?
Same bits by batch invariance."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
full = PlipModel(cfg, sd, device=dev, dtype="bf16", max_batch=B)
parts = int(sys.argv[1]) if len(sys.argv) > 1 else 2
halves = [PlipModel(cfg, sd, device=dev, dtype="bf16", max_batch=B // parts) for _ in range(parts)]
streams = [(torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)) for _ in range(parts)]
main = torch.cuda.current_stream(dev)

def step_full():
    img, txt = full.engine.encode_pair(px, ids, mask, normalize=True, overlap=True)
    return full.engine.logits(img, txt, scale=full.engine.logit_scale_exp, want_text=False)[0]

def step_split(order=0):
    n = B // parts
    imgs, txts = [None] * parts, [None] * parts
    for k in range(parts):
        for st in streams[k]:
            st.wait_stream(main)
    seq = [(k, t) for k in range(parts) for t in (0, 1)] if order == 0 else [(k, t) for t in (1, 0) for k in range(parts)]
    for k, t in seq:
        sl = slice(k * n, (k + 1) * n)
        with torch.cuda.stream(streams[k][t]):
            if t == 0:
                imgs[k] = halves[k].engine.encode_image(px[sl], True)
            else:
                txts[k] = halves[k].engine.encode_text(ids[sl], mask[sl], True)
    for k in range(parts):
        for st in streams[k]:
            main.wait_stream(st)
    img, txt = torch.cat(imgs), torch.cat(txts)
    return full.engine.logits(img, txt, scale=full.engine.logit_scale_exp, want_text=False)[0]

def timed(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): out = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, out

a, ref = timed(step_full)
print(f"two streams, one engine, bs 256: {a:.3f} ms", flush=True)
for rep in range(2):
    for order in (0, 1):
        b, out = timed(lambda: step_split(order))
        print(f"{2 * parts} streams, {parts} engines of bs {B // parts} (order {order}): {b:.3f} ms   logits identical: {bool(torch.equal(out, ref))}", flush=True)
    a, _ = timed(step_full)
    print(f"two streams, one engine, bs 256: {a:.3f} ms", flush=True)
