"""Multi-GPU layer: one process per GPU, batches sharded across ranks, ONE collective.

The towers are embarrassingly parallel over samples (weights replicated, <=605 MB),
so the only exchange step of the path is the all-gather of the L2-normalised
embeddings ``[N/W, P]`` into ``[N, P]`` on every rank for the similarity product (image and
text matrices stacked into one buffer, so a step contains exactly one collective)
(SURVEY.md section 8e).  ``torch.distributed`` backend "nccl" is RCCL over xGMI on the MI355X
node; "gloo" runs the same code on CPU for the tests.  The reference has no
distributed path at all (single process, plip.py:15).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of n samples for ``rank``; sizes differ by at most one."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_rows(local: torch.Tensor, group=None, equal_shards: bool = False, always_collective: bool = False) -> torch.Tensor:
    """Concatenate every rank's ``[n_r, ...]`` rows in rank order -> ``[sum n_r, ...]``.

    Equal shards go through one ``all_gather_into_tensor`` (a single RCCL collective on
    contiguous buffers); ragged shards are padded to the longest one first.  ``equal_shards=True``
    (the caller guarantees n_r is the same on every rank, e.g. a fixed per-GPU batch) skips the
    size exchange and its host synchronisation: the step then contains exactly ONE collective per matrix.
    ``always_collective=True`` issues the collective in a ONE-rank group too (where the result is the input): the way to run
    communicator creation, the collective's stream ordering against the tower streams and the gathered views on a single
    GPU (tests/test_gpu_dist.py, bench.py's ``rccl_one_rank`` field) -- a one-GPU box cannot run two ranks.
    """
    if not dist.is_available() or not dist.is_initialized():
        return local
    if dist.get_world_size(group) == 1 and not always_collective:
        return local
    world = dist.get_world_size(group)
    if equal_shards:
        local = local.contiguous()
        out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local, group=group)
        return out
    n_local = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    sizes = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local, group=group)
    sizes = [int(s.item()) for s in sizes]
    n_max = max(sizes)
    local = local.contiguous()
    if all(s == n_max for s in sizes):
        out = torch.empty((world * n_max,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local, group=group)
        return out
    padded = torch.zeros((n_max,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    out = torch.empty((world * n_max,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, padded, group=group)
    return torch.cat([out[r * n_max: r * n_max + sizes[r]] for r in range(world)], dim=0)


def sharded_pair_logits(model, pixels_local: torch.Tensor, ids_local: torch.Tensor,
                        attention_mask_local: Optional[torch.Tensor] = None, group=None, overlap: bool = True,
                        equal_shards: bool = False, always_collective: bool = False):
    """One data-parallel step of CLIPModel.forward: this rank embeds ITS images and captions,
    the normalised embeddings are all-gathered, and the rank computes its row block of
    ``logits_per_image`` ([n_local, N_text]) against every caption of the global batch.

    Returns ``(logits_rows, image_embeds_all, text_embeds_all)``.
    """
    eng = model.engine
    if hasattr(eng, "encode_pair"):
        img, txt = eng.encode_pair(pixels_local, ids_local, attention_mask_local, normalize=True, overlap=overlap)
    else:
        img = eng.encode_image(pixels_local, normalize=True)
        txt = eng.encode_text(ids_local, attention_mask_local, normalize=True)
    grouped = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if grouped else 1
    if equal_shards and (world > 1 or (always_collective and grouped)) and img.shape == txt.shape:
        # fixed per-rank batch: both embedding matrices travel in ONE all-gather ([2, n, P] per rank)
        both = all_gather_rows(torch.stack((img, txt)).unsqueeze(0), group, True, always_collective)     # [W, 2, n, P]
        img_all = both[:, 0].reshape(-1, img.shape[1])
        txt_all = both[:, 1].reshape(-1, txt.shape[1])
    else:
        txt_all = all_gather_rows(txt, group, equal_shards, always_collective)
        img_all = all_gather_rows(img, group, equal_shards, always_collective)
    lpi, _, _ = eng.logits(img, txt_all, scale=eng.logit_scale_exp, want_text=False)
    return lpi, img_all, txt_all


def sharded_zero_shot(model, pixels_local: torch.Tensor, class_text_embeds: torch.Tensor, group=None,
                      always_collective: bool = False):
    """Config-4 style zero-shot: class prompts ([C,P], tiny) are replicated, every rank
    classifies its image shard and only the int32 predictions are gathered."""
    eng = model.engine
    img = eng.encode_image(pixels_local, normalize=True)
    _, _, pred = eng.logits(img, class_text_embeds, scale=1.0, want_text=False, want_argmax=True)
    return all_gather_rows(pred, group, always_collective=always_collective)


def sharded_retrieval_topk(model, text_embeds_local: torch.Tensor, image_embeds_local: torch.Tensor, k: int = 50,
                           group=None, always_collective: bool = False):
    """Text-to-image retrieval over a corpus sharded by rank (reproducibility/evaluation/retrieval/retrieval.py:13-18
    at corpus scale): the image embeddings are all-gathered once, every rank ranks ITS captions against the whole
    corpus with the fused similarity + top-k head (no [N, N] matrix anywhere) and the [n_r, k] index blocks are
    gathered in rank order.  Returns int64 [N_text, k] global image indices."""
    eng = model.engine
    img_all = all_gather_rows(image_embeds_local, group, always_collective=always_collective)
    k = min(int(k), int(img_all.shape[0]))
    best = eng.similarity_topk(text_embeds_local, img_all, k)
    return all_gather_rows(best, group, always_collective=always_collective)
