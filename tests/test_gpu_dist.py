"""The RCCL side of plip_amd/dist.py on ONE MI355X.

A one-GPU box cannot run two ranks, but a ONE-rank ``nccl`` process group still exercises what the world-size-2 gloo
tests (tests/test_dist_cpu.py) cannot: communicator creation on ``device_id``, ``all_gather_into_tensor`` on device
buffers, the collective's stream ordering against the engine's two tower streams, and the stacked ``[W, 2, n, P]``
view the step gathers.  ``always_collective=True`` makes the helpers issue the collective at world size 1 (where the
gathered matrix must equal the local one bit for bit).
"""
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nccl_one_rank():
    import torch.distributed as dist
    if dist.is_initialized():           # another test left a group behind: do not stack a second default group on it
        pytest.skip("a default process group already exists in this process")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)   # RCCL on ROCm
    yield dist
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [False, True], ids=["one_stream", "two_streams"])
def test_pair_step_through_a_one_rank_rccl_all_gather(nccl_one_rank, engines, overlap):
    from plip_amd.dist import sharded_pair_logits
    dist = nccl_one_rank
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    model, cfg, sd, px, ids, mask = engines("tiny_b6", "bf16")
    dev = model.device
    px, ids, mask = (torch.from_numpy(np.asarray(a)).to(dev) for a in (px, ids, mask))
    ref_lpi, ref_img, ref_txt = sharded_pair_logits(model, px, ids, mask, overlap=overlap, equal_shards=True)
    for _ in range(3):                   # the collective is enqueued behind the towers every step: ordering must hold repeatedly
        lpi, img_all, txt_all = sharded_pair_logits(model, px, ids, mask, overlap=overlap, equal_shards=True,
                                                    always_collective=True)
    torch.cuda.synchronize(dev)
    # one stacked [W=1, 2, n, P] gather -> views of the image and text halves
    assert img_all.shape == ref_img.shape and txt_all.shape == ref_txt.shape
    assert torch.equal(img_all, ref_img) and torch.equal(txt_all, ref_txt) and torch.equal(lpi, ref_lpi)
    # the ragged form (size exchange + gather) on the same group
    lpi2, img2, txt2 = sharded_pair_logits(model, px[:5], ids[:5], mask[:5], overlap=overlap, always_collective=True)
    assert torch.equal(img2, ref_img[:5]) and torch.equal(lpi2, ref_lpi[:5, :5])


def test_zero_shot_and_retrieval_heads_through_one_rank_rccl(nccl_one_rank, engines):
    from plip_amd.dist import sharded_retrieval_topk, sharded_zero_shot
    model, cfg, sd, px, ids, mask = engines("tiny_b6", "bf16")
    dev = model.device
    px, ids, mask = (torch.from_numpy(np.asarray(a)).to(dev) for a in (px, ids, mask))
    eng = model.engine
    classes = eng.encode_text(ids[:4], mask[:4], normalize=True)
    pred = sharded_zero_shot(model, px, classes, always_collective=True)
    want = sharded_zero_shot(model, px, classes)
    assert pred.dtype == torch.int32 and pred.shape == (px.shape[0],) and torch.equal(pred, want)
    img = eng.encode_image(px, normalize=True)
    txt = eng.encode_text(ids, mask, normalize=True)
    got = sharded_retrieval_topk(model, txt, img, k=3, always_collective=True)
    assert torch.equal(got, sharded_retrieval_topk(model, txt, img, k=3))
    torch.cuda.synchronize(dev)
