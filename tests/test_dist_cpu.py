"""The N>1 path (plip_amd/dist.py) on CPU: world_size-2 gloo processes stand in for two
MI355X ranks; the engine is replaced by a deterministic fake so only the sharding and
the single all-gather are exercised."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from plip_amd.dist import (all_gather_rows, shard_bounds, sharded_pair_logits, sharded_retrieval_topk,
                           sharded_zero_shot)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _FakeEngine:
    """encode = fixed random projection + L2 normalise; logits = scale * a @ b.T (all on CPU)."""
    logit_scale_exp = 7.0

    def __init__(self):
        g = torch.Generator().manual_seed(0)
        self.wi = torch.randn(12, 8, generator=g)
        self.wt = torch.randn(5, 8, generator=g)

    def encode_image(self, px, normalize=False):
        e = px.reshape(px.shape[0], -1)[:, :12] @ self.wi
        return e / e.norm(dim=-1, keepdim=True) if normalize else e

    def encode_text(self, ids, mask=None, normalize=False, eos_token_id=None):
        e = ids.float()[:, :5] @ self.wt
        return e / e.norm(dim=-1, keepdim=True) if normalize else e

    def similarity_topk(self, keys, space, k):
        return torch.argsort(-(keys @ space.T), dim=1, stable=True)[:, :k]

    def logits(self, a, b, scale=1.0, want_text=True, want_argmax=False):
        l = scale * a @ b.T
        return l, (l.T.contiguous() if want_text else None), (l.argmax(1).int() if want_argmax else None)


class _FakeModel:
    def __init__(self):
        self.engine = _FakeEngine()


def _worker(rank, world, port, n, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(42)
        px = torch.randn(n, 3, 2, 2, generator=g)
        ids = torch.randint(1, 100, (n, 7), generator=g)
        lo, hi = shard_bounds(n, rank, world)
        model = _FakeModel()
        rows, img_all, txt_all = sharded_pair_logits(model, px[lo:hi], ids[lo:hi])
        cls = model.engine.encode_text(ids[:3], normalize=True)
        pred = sharded_zero_shot(model, px[lo:hi], cls)
        ragged = all_gather_rows(torch.full((rank + 1, 2), float(rank)))
        img_l = model.engine.encode_image(px[lo:hi], normalize=True)
        txt_l = model.engine.encode_text(ids[lo:hi], normalize=True)
        best = sharded_retrieval_topk(model, txt_l, img_l, k=3)
        if n % world == 0:   # the fixed-batch fast path (no size exchange) must give the same matrix
            fast = sharded_pair_logits(model, px[lo:hi], ids[lo:hi], equal_shards=True)
            assert torch.equal(fast[0], rows) and torch.equal(fast[1], img_all) and torch.equal(fast[2], txt_all)
        q.put((rank, lo, hi, rows.numpy(), img_all.numpy(), txt_all.numpy(), pred.numpy(), ragged.numpy(), best.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [8, 7])
def test_two_rank_shard_and_gather_matches_single_process(n):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # single-process reference
    g = torch.Generator().manual_seed(42)
    px = torch.randn(n, 3, 2, 2, generator=g)
    ids = torch.randint(1, 100, (n, 7), generator=g)
    eng = _FakeEngine()
    img, txt = eng.encode_image(px, True), eng.encode_text(ids, normalize=True)
    full = (7.0 * img @ txt.T).numpy()
    cls = eng.encode_text(ids[:3], normalize=True)
    pred = (img @ cls.T).argmax(1).numpy()
    covered = []
    want_best = torch.argsort(-(txt @ img.T), dim=1, stable=True)[:, :3].numpy()
    for rank, lo, hi, rows, img_all, txt_all, p, ragged, best in results:
        np.testing.assert_array_equal(best, want_best)                          # every rank: the full [N, k] ranking
        np.testing.assert_allclose(rows, full[lo:hi], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(img_all, img.numpy(), rtol=1e-6, atol=1e-7)   # every rank holds the full matrix
        np.testing.assert_allclose(txt_all, txt.numpy(), rtol=1e-6, atol=1e-7)
        np.testing.assert_array_equal(p, pred)
        np.testing.assert_array_equal(ragged, np.array([[0, 0], [1, 1], [1, 1]], dtype=np.float32))
        covered += list(range(lo, hi))
    assert covered == list(range(n))


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 8, 256, 1000003):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_all_gather_rows_is_identity_without_process_group():
    x = torch.arange(6.0).reshape(3, 2)
    assert all_gather_rows(x) is x
