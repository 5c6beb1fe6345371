"""Host side of ``encode_images``: keep the GPU fed.

The reference's loop (plip.py:41-52) decodes a batch with PIL on the main thread, copies it to the device, runs the
tower and copies the features back -- every step waits for the previous one; ``CLIPEmbedder`` hides the decode behind
``DataLoader(num_workers=...)`` worker processes (reproducibility/embedders/plip.py:41-42).  Here:

* items of batch k+1 are decoded / resized by a thread pool (PIL releases the GIL in its C loops) while batch k is
  on the GPU -- ``num_workers`` keeps the reference's meaning;
* the prepared batch goes through one of two pinned staging buffers and an H2D copy on a dedicated copy stream, so
  the transfer of batch k+1 overlaps the towers of batch k (uint8 tiles: 38.5 MB per 256 images, ~0.6 ms on PCIe
  Gen5; fp32 pixels are 4x that);
* features stay on the GPU until the end (one D2H copy), instead of one synchronising ``.cpu()`` per batch;
* with ``lanes`` (``Engine.lanes()``: the engine + a ``plipmi_clone`` of it on a second stream) consecutive batches run side by side on
  the GPU -- one batch's launch boundaries, epilogues and pooled tail under the next batch's GEMMs -- and there are two staging slots
  per lane, so that the copy of batch k+2 does not wait for batch k (configs[3]'s shard H2D-inclusive: 96 -> 106 k img/s).
"""
from __future__ import annotations

from concurrent.futures import ThreadPoolExecutor
from typing import Callable, List, Optional, Sequence

import numpy as np
import torch


def _prepare_batch(items: Sequence, prepare_item: Callable, pool: Optional[ThreadPoolExecutor]) -> np.ndarray:
    arrs = list(pool.map(prepare_item, items)) if pool is not None and len(items) > 1 else [prepare_item(i) for i in items]
    return np.stack(arrs)


def run_batches(items: Sequence, batch_size: int, prepare_item: Optional[Callable], consume: Callable,
                device: Optional[torch.device] = None, num_workers: int = 1,
                prepare_batch: Optional[Callable] = None, lanes: Optional[Sequence] = None) -> List[torch.Tensor]:
    """``consume(batch_on_device)`` for consecutive batches of ``prepare_item(item)`` arrays, order preserved.

    With ``prepare_batch(items, pool) -> (tag, ndarray)`` the caller prepares whole batches (and may route them:
    raw tiles / to-be-resized images / host-preprocessed pixels) and ``consume(tag, batch_on_device)`` gets the tag.
    ``device=None`` (CPU-side tests) skips the pinned / copy-stream part and hands host tensors to ``consume``.
    ``lanes`` = ``Engine.lanes()``: batch k runs on lane k % n -- ``consume(..., engine)`` gets the lane's engine as its last
    argument and is called inside the lane's stream --, so consecutive batches overlap on the GPU; every lane is joined into the
    caller's stream before the outputs are returned."""
    n = len(items)
    if n == 0:
        return []
    bounds = [(s, min(s + batch_size, n)) for s in range(0, n, batch_size)]
    workers = max(0, int(num_workers))
    pool = ThreadPoolExecutor(max_workers=workers) if workers > 1 else None
    # one extra single-thread executor runs "prepare batch k+1" concurrently with the GPU work on batch k
    ahead = ThreadPoolExecutor(max_workers=1)
    outs: List[torch.Tensor] = []
    if prepare_batch is not None:
        prep = lambda chunk: prepare_batch(chunk, pool)
        eat = lambda tagged, t, *lane: consume(tagged[0], t, *lane)
        arr = lambda tagged: tagged[1]
    else:
        prep = lambda chunk: _prepare_batch(chunk, prepare_item, pool)
        eat = lambda host, t, *lane: consume(t, *lane)
        arr = lambda host: host
    try:
        fut = ahead.submit(prep, items[bounds[0][0]:bounds[0][1]])
        use_gpu = device is not None and torch.device(device).type == "cuda"
        if use_gpu:
            # high priority: a queue class of its own -- an ordinary stream may share its hardware queue with the stream the towers run
            # on, and its copies then run in order with the kernels instead of beside them (profiles/r06_h2d_copy_stream.txt)
            copy_stream = torch.cuda.Stream(device=device, priority=-1)
            # device-side slots: two for one lane (batch k+1 is copied while batch k runs); with lanes two batches run at once, so the
            # copy of batch k+2 needs a third slot to start before batch k has finished -- two per lane
            ns = 2 * len(lanes) if lanes else 2
            staging = [None] * ns
            copied = [torch.cuda.Event() for _ in range(ns)]
            consumed = [None] * ns
            pinned_src = [False] * ns
        for k, (lo, hi) in enumerate(bounds):
            prepared = fut.result()
            host = arr(prepared)
            if use_gpu and k >= 1 and pinned_src[(k - 1) % ns]:
                copied[(k - 1) % ns].synchronize()      # batch k-1 has left its producer-owned pinned buffer: it may be refilled
            if k + 1 < len(bounds):
                nlo, nhi = bounds[k + 1]
                fut = ahead.submit(prep, items[nlo:nhi])
            if not use_gpu:
                outs.append(eat(prepared, host if torch.is_tensor(host) else torch.from_numpy(host), *([lanes[k % len(lanes)][0]] if lanes else [])))
                continue
            slot = k % ns
            if consumed[slot] is not None:
                consumed[slot].synchronize()            # the tower that read this slot's device copy has finished
            if torch.is_tensor(host) and host.is_pinned():
                # the producer already wrote into page-locked memory: no staging copy.  The producer may refill that very
                # buffer while preparing batch k+2 (the usual double-buffer pattern), so the copy of batch k out of it must be
                # complete before `prepare` runs again on this slot: waited for below, before the next submit
                src = host
            else:
                ht = host if torch.is_tensor(host) else torch.from_numpy(host)
                if staging[slot] is None or staging[slot].shape != ht.shape or staging[slot].dtype != ht.dtype:
                    staging[slot] = torch.empty(ht.shape, dtype=ht.dtype).pin_memory()
                staging[slot].copy_(ht)
                src = staging[slot]
            pinned_src[slot] = src is host
            caller = torch.cuda.current_stream(device)
            lane = lanes[k % len(lanes)] if lanes else None
            main = lane[1] if lane else caller
            with torch.cuda.stream(copy_stream):
                dev = src.to(device, non_blocking=True)
                copied[slot].record(copy_stream)
            main.wait_event(copied[slot])
            dev.record_stream(main)
            if lane:
                with torch.cuda.stream(main):
                    out = eat(prepared, dev, lane[0])
                if torch.is_tensor(out):
                    out.record_stream(caller)
                outs.append(out)
            else:
                outs.append(eat(prepared, dev))
            consumed[slot] = torch.cuda.Event()
            consumed[slot].record(main)
        if use_gpu and lanes:
            caller = torch.cuda.current_stream(device)
            for _, st in lanes[1:]:
                caller.wait_stream(st)
    finally:
        ahead.shutdown(wait=True)
        if pool is not None:
            pool.shutdown(wait=True)
    return outs
