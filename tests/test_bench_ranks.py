"""The multi-rank control flow of ``bench.py`` and ``examples/zero_shot_sharded.py`` executed before the driver does
(VERDICT r3 item 6): two gloo ranks on this CPU box run the very ``main()`` the 2/4/8-GPU runs use -- rendezvous from the
torchrun environment, warm-up, fenced windows of exactly --steps steps with the MAX over ranks, rank-0-only printing of ONE
JSON line, barrier + teardown -- with the engine replaced by a stub (``--backend gloo`` + ``main(model_factory=...)``: a
rehearsal, the line says so).  What is NOT covered here is RCCL itself and the towers; those run under ``-m gpu``."""
import contextlib
import io
import json
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _StubEngine:
    """The surface sharded_pair_logits / the example need, as cheap deterministic tensor algebra on the host."""
    device_name = "stub (gloo rehearsal)"
    logit_scale = 2.0
    logit_scale_exp = 7.389

    def __init__(self, device):
        self.device = torch.device(device)
        g = torch.Generator().manual_seed(0)
        self.wi, self.wt = torch.randn(24, 16, generator=g), torch.randn(8, 16, generator=g)

    def _n(self, e, normalize):
        return e / e.norm(dim=-1, keepdim=True) if normalize else e

    def encode_image(self, px, normalize=False):
        return self._n(px.reshape(px.shape[0], -1)[:, :24].float() @ self.wi, normalize)

    def encode_image_u8(self, tiles, normalize=False):
        return self.encode_image(tiles.float() / 255.0, normalize)

    def encode_text(self, ids, mask=None, normalize=False, eos_token_id=None):
        return self._n(torch.as_tensor(ids).float()[:, :8] / 1000.0 @ self.wt, normalize)

    def logits(self, a, b, scale=1.0, want_text=True, want_argmax=False):
        l = scale * a @ b.T
        return l, (l.T.contiguous() if want_text else None), (l.argmax(1).int() if want_argmax else None)


class _StubModel:
    def __init__(self, cfg, sd, device="cpu", dtype="bf16", max_batch=256, **_):
        self.config, self.engine = cfg, _StubEngine(device)


def _rank_main(rank, world, port, which, argv, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        if which == "bench":
            import bench
            bench.main(argv, model_factory=_StubModel)
        else:
            import zero_shot_sharded
            zero_shot_sharded.main(argv, model_factory=_StubModel)
    q.put((rank, buf.getvalue()))


def _run_two_ranks(which, argv):
    world, ctx = 2, mp.get_context("spawn")
    q, port = ctx.Queue(), _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, which, argv, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0                       # both ranks leave through destroy_process_group, none hangs or raises
    return out


def test_bench_main_on_two_gloo_ranks_prints_one_line_from_rank_zero():
    out = _run_two_ranks("bench", ["--gpus", "2", "--steps", "3", "--warmup", "1", "--backend", "gloo"])
    assert out[1].strip() == ""                      # only rank 0 prints
    lines = [l for l in out[0].splitlines() if l.strip()]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 3 and res["warmup"] == 1 and res["scaling"] == "weak"
    assert res["metric"] == "image+text pairs embedded/sec at 224px bs=256" and res["unit"] == "pairs/s"
    cfg = res["config"]
    assert cfg["global_batch"] == 512 and cfg["per_gpu_batch"] == 256 and cfg["parallelism"] == "dp2"
    assert cfg["collective"].startswith("RCCL all-gather") and cfg["collective_us"] > 0      # the collective timed on its own, max over ranks
    assert res["rccl_ranks"] == 2 and len(res["rank_devices"]) == 2 and res["rank_devices"][1].startswith("rank 1:")
    # value = the units ALL ranks processed / the max-over-ranks time of the first window of exactly --steps steps
    assert res["value"] == pytest.approx(512 * 3 / (res["ms_per_step"] * 3e-3), rel=2e-3)
    assert len(res["windows"]["ms_per_step"]) == 3 and res["windows"]["ms_per_step"][0] == pytest.approx(res["ms_per_step"], abs=2e-3)
    assert "REHEARSAL" in res["data"]                # a stub-engine line can never pass for a measurement
    assert res["roofline"] is None and res["cpu_baseline"] is None


def test_bench_refuses_the_rehearsal_backend_without_a_stub_and_a_world_size_mismatch(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    with pytest.raises(SystemExit, match="rehearsal"):
        bench.main(["--backend", "gloo"])
    with pytest.raises(SystemExit, match="cannot follow self-launched ranks"):
        bench.main(["--gpus", "2", "--backend", "gloo"], model_factory=_StubModel)
    monkeypatch.setenv("WORLD_SIZE", "4")
    with pytest.raises(SystemExit, match="!= WORLD_SIZE 4"):
        bench.main(["--gpus", "2"])
    monkeypatch.delenv("WORLD_SIZE")
    if not torch.cuda.is_available():
        with pytest.raises(SystemExit, match="no CPU path"):      # the product path never falls back to the host
            bench.main([])


def test_bench_launches_its_own_ranks_when_started_as_plain_python():
    """the driver's command form is `python3 bench.py --gpus N ...` with NO rank environment.  bench.py then
    re-executes itself under torch.distributed.run (one process per GPU), and the caller still reads exactly one JSON line."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--backend", "gloo",
                        "--stub-engine", "tests/test_bench_ranks.py:_StubModel"], env=env, cwd="/tmp", stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["rccl_ranks"] == 2 and res["config"]["global_batch"] == 512
    assert res["launcher"] == "self" and res["world_size_env"] == "2"
    assert len(res["rank_devices"]) == 2 and "REHEARSAL" in res["data"]


def test_bench_in_process_gives_the_caller_its_stdout_back(monkeypatch):
    """main() parks fd 1 on stderr while it runs; it must restore it and close its duplicate on every way out."""
    sys.path.insert(0, ROOT)
    import bench
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    before = os.fstat(1)
    n_fds = len(os.listdir("/proc/self/fd"))
    with pytest.raises(SystemExit):
        bench.main(["--backend", "gloo"])                     # refused inside the redirected region
    after = os.fstat(1)
    assert (before.st_dev, before.st_ino) == (after.st_dev, after.st_ino)
    assert len(os.listdir("/proc/self/fd")) == n_fds


def test_sharded_zero_shot_example_on_two_gloo_ranks():
    out = _run_two_ranks("example", ["--images", "70", "--classes", "4", "--batch", "16", "--backend", "gloo"])
    assert out[1].strip() == ""
    line = out[0].strip()
    assert line.startswith("70 images on 2 GPU(s)") and "class histogram" in line
    hist = json.loads(line[line.index("["):])
    assert len(hist) == 4 and sum(hist) == 70        # every image of both shards classified exactly once
