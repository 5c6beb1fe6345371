#!/usr/bin/env python
"""bench.py -- the PLIP embedding hot path on N MI355X GPUs of one node.

One "step" = one pass of the path over one synthetic batch per GPU:
    pixels [B,3,224,224] fp32 + token ids [B,77]  (already resident in HBM)
      -> image tower, text tower (libplipmi.so), L2-normalise
      -> all-gather of the embeddings across ranks (RCCL, only when N > 1)
      -> this rank's rows of logits_per_image against every caption of the global batch.
Workload at every N: BASELINE.json config "Full dual-encoder (image+text) bs=256 bf16"
per GPU (weak scaling: the global batch is 256*N pairs).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus 8 --steps 20 --warmup 3          # no rank environment: bench.py starts its own 8 ranks (self_launch)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 20 --warmup 3

Rank 0 prints ONE JSON line (see README / DESIGN.md section 5 for the fields).  `value` comes from the first timed window
of exactly --steps steps; two more windows of the same length follow and are reported beside it (`windows`).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"bf16": 2500.0, "f16": 2500.0, "f32": 157.3}   # dense MFMA peaks, MI355X_MICROARCH.md "Chip-level parameters"
HBM_PEAK_GBS = 8000.0                                          # HBM3E 8 TB/s, same table


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="pairs per GPU per step")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"])
    ap.add_argument("--arch", default="ViT-B/32")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="lower bound of CPU work for the baseline sample")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-extras", "--no-fp32-tower", dest="no_extras", action="store_true",
                    help="skip the side measurements (f16 engine, dense last block, caption packing, larger batches, "
                         "ViT-L/14@336 share, fp32 engine)")
    ap.add_argument("--overlap", type=int, default=1, help="1: text tower on a second HIP stream (default), 0: one stream")
    # "gloo": host-only REHEARSAL of the multi-rank control flow (rendezvous, fenced windows, max over ranks, rank-0-only
    # printing, teardown) for tests/test_bench_ranks.py, which supplies a stub engine through main(model_factory=...).
    # It measures nothing: the line it prints says so, and without a factory the flag is refused.
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help=argparse.SUPPRESS)
    # rehearsal only: "path/to/file.py:Name" of the stub engine factory, so that a SELF-LAUNCHED rehearsal (ranks in other processes)
    # can find the stub main(model_factory=...) would have been handed in-process
    ap.add_argument("--stub-engine", default=None, help=argparse.SUPPRESS)
    # set by self_launch() on the ranks it starts; "torchrun" when the ranks come from somebody else's launcher (the driver's form)
    ap.add_argument("--launcher", default=None, choices=["self"], help=argparse.SUPPRESS)
    return ap.parse_args(argv)


def self_launch(args, argv):
    """`python bench.py --gpus N` with N > 1 and no rank environment: start the N ranks ourselves, exactly the way the driver's
    multi-GPU command does (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
    bench.py ...`), relay their stderr, and print rank 0's ONE JSON line as ours.  Returns the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    argv = list(sys.argv[1:] if argv is None else argv)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv + ["--launcher", "self"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")           # dmabuf IPC: what RCCL between processes needs on this driver
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in proc.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    for l in proc.stdout.splitlines():
        if l not in lines and l.strip():
            print(l, file=sys.stderr)                            # anything else a rank wrote to fd 1 is not part of the contract
    if proc.returncode == 0 and len(lines) != 1:
        print(f"bench.py: expected one JSON line from rank 0, got {len(lines)}", file=sys.stderr)
        return 1
    if lines:
        print(lines[-1], flush=True)
    return proc.returncode


def load_stub_factory(spec):
    import importlib.util
    path, _, name = spec.rpartition(":")
    path = path if os.path.isabs(path) else os.path.join(ROOT, path)
    sp = importlib.util.spec_from_file_location("_bench_stub_engine", path)
    mod = importlib.util.module_from_spec(sp)
    sp.loader.exec_module(mod)
    return getattr(mod, name)


def load_pmc_traffic(kernel_name):
    """HBM/fabric bytes per launch of a kernel from the committed rocprofv3 --pmc summary (profiles/pmc_traffic.json,
    produced by tools/pmc_summary.py on the GPU box).  The file is stamped with the digest of the kernel sources it was
    collected on; if the sources have changed since (or there is no stamp), the counters describe another kernel and
    None is returned -- the bench line then says `"traffic": null` instead of quoting a stale number."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        from plip_amd.build import source_digest
        with open(path) as f:
            db = json.load(f)
        if db.get("_stamp", {}).get("csrc_sha16") != source_digest():
            return None
        return db.get(kernel_name)
    except (OSError, ValueError):
        return None


def live_pmc_traffic(kernel_name, timeout=180):
    """HBM/fabric bytes per launch of `kernel_name` measured ON THIS BOX, now: two `rocprofv3 --kernel-trace --pmc` passes (FETCH_SIZE, then
    WRITE_SIZE -- one counter per pass, kernel trace only, as MI355X_MICROARCH.md's HBM section prescribes; FETCH_SIZE doubled on gfx950,
    both in KB) over a 3-step one-stream run of this very script.  None when rocprofv3 is missing or a pass fails: the caller then falls
    back to the committed, source-stamped profiles/pmc_traffic.json."""
    import csv
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        from pmc_summary import pretty
    except ImportError:
        return None
    vals = {}
    try:
        with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
            for counter in ("FETCH_SIZE", "WRITE_SIZE"):
                out = os.path.join(tmp, counter)
                cmd = [exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "-o", "pmc", "--", sys.executable,
                       os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--overlap", "0", "--no-cpu-baseline",
                       "--no-profile", "--no-extras"]
                subprocess.run(cmd, cwd="/tmp", env={**os.environ, "TMPDIR": "/tmp"}, stdout=subprocess.DEVNULL,
                               stderr=subprocess.DEVNULL, timeout=timeout, check=True)
                path = next((os.path.join(dp, f) for dp, _, fs in os.walk(out) for f in fs if f.endswith("counter_collection.csv")), None)
                if path is None:
                    return None
                with open(path) as f:
                    v = [float(r["Counter_Value"]) for r in csv.DictReader(f)
                         if r["Counter_Name"] == counter and pretty(r["Kernel_Name"]) == kernel_name]
                if not v:
                    return None
                vals[counter] = (sum(v) / len(v), len(v))
    except (OSError, subprocess.SubprocessError, KeyError, ValueError):
        return None
    fetch, write = 2.0 * 1024.0 * vals["FETCH_SIZE"][0], 1024.0 * vals["WRITE_SIZE"][0]
    return {"bytes_per_launch": round(fetch + write), "fetch_bytes": round(fetch), "write_bytes": round(write),
            "launches": vals["FETCH_SIZE"][1],
            "note": "measured in THIS run on this box: rocprofv3 --kernel-trace --pmc FETCH_SIZE (x2 gfx950 correction) and WRITE_SIZE in "
                    "separate passes over `bench.py --steps 3 --overlap 0`, mean over the kernel's launches"}


def usable_cores() -> int:
    """Host cores this process may actually run on: affinity mask, capped by the cgroup CPU quota.
    (os.cpu_count() reports every core of the node; asking torch for 256 threads inside a container
    that is throttled to a few CPUs makes the CPU baseline 100x slower than it should be.)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, per = int(f.read()), int(g.read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return max(1, min(n, 64))      # torch CPU GEMMs stop scaling (and oversubscribe) far below a full 2-socket node


def cpu_baseline(cfg, sd, seconds):
    """The reference's own forward (HF CLIPModel, what plip.py:50,68 call) on the host cores of
    THIS box, same synthetic weights, bounded sample.  Falls back to the numpy oracle ("port")."""
    from plip_amd import weights as W
    cores = usable_cores()
    B = 32
    px = W.synthetic_pixels(cfg, B, seed=1)
    ids, mask = W.synthetic_ids(cfg, B, seed=2)
    try:
        from oracle import hf_reference as H
        torch.set_num_threads(cores)       # (torchrun exports OMP_NUM_THREADS=1: the thread count is set explicitly)
        model = H.build_model(cfg, sd, "sdpa")
        tp, ti, tm = torch.from_numpy(px), torch.from_numpy(ids), torch.from_numpy(mask)

        def run():
            with torch.no_grad():
                return model(input_ids=ti, pixel_values=tp, attention_mask=tm).logits_per_image     # late-bound: B=32, then B=256
        kind, what = "reference", "HF transformers CLIPModel.forward, torch CPU fp32 sdpa"
    except Exception as e:  # transformers missing -> time the oracle restatement instead
        from oracle import clip_oracle as O

        def run():
            return O.clip_forward(px, ids, sd, cfg, mask)["logits_per_image"]
        kind, what = "port", f"numpy oracle (HF unavailable: {type(e).__name__})"
    run()                                   # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        run()
        n += 1
        el = time.perf_counter() - t0
        if (el >= seconds and n >= 2) or el >= 3 * seconds:
            break
    cpu_name = ""
    try:
        with open("/proc/cpuinfo") as f:
            cpu_name = next((l.split(":", 1)[1].strip() for l in f if l.startswith("model name")), "")
    except OSError:
        pass
    res = {"value": round(n * B / el, 2), "unit": "pairs/s", "cores": cores, "kind": kind,
           "sample": f"{n} x {B} synthetic pairs ({el:.1f} s): {what}; {cpu_name}"}
    # BASELINE.md section 3 also asks for the reference at the bench's own batch size (bounded: ~10 s)
    try:
        B2 = 256
        px2 = W.synthetic_pixels(cfg, B2, seed=1000)
        ids2, mask2 = W.synthetic_ids(cfg, B2, seed=2000)
        if kind == "reference":
            tp, ti, tm = torch.from_numpy(px2), torch.from_numpy(ids2), torch.from_numpy(mask2)
        else:
            px, ids, mask = px2, ids2, mask2
        run()
        n2, t1 = 0, time.perf_counter()
        while True:
            run()
            n2 += 1
            el2 = time.perf_counter() - t1
            if el2 >= 0.6 * seconds or n2 >= 4:
                break
        res["b256"] = {"value": round(n2 * B2 / el2, 2), "unit": "pairs/s", "sample": f"{n2} x {B2} synthetic pairs ({el2:.1f} s)"}
    except Exception as e:  # pragma: no cover
        res["b256"] = {"error": repr(e)}
    return res


def executed_gflop_per_pair(cfg, dtype):
    """`dense_equivalent_tflops` prices a pair at SURVEY.md section 8d's dense figure (14.777 GFLOP for ViT-B/32: every token
    through every Linear).  The 16-bit engines run the LAST block's out_proj / fc1 / fc2 only on the row that is pooled
    afterwards (CLS / EOS) -- the other rows of that block cannot reach get_image_features / get_text_features -- so they
    execute slightly fewer FLOPs than that; `executed_tflops` and every kernel roofline use executed FLOPs."""
    full = cfg.pair_flops()
    pooled = dtype in ("bf16", "f16")
    saved = 0.0
    if pooled:
        for tokens, D, F in ((cfg.v_tokens, cfg.v_width, cfg.v_mlp), (cfg.context_length, cfg.t_width, cfg.t_mlp)):
            saved += (tokens - 1) * (2.0 * D * D + 4.0 * D * F)
    return {"dense_reference": round(full / 1e9, 4), "executed": round((full - saved) / 1e9, 4),
            "pooled_last_block": bool(pooled)}


def logits_error_vs_hf_golden(model, cfg, sd, px, ids, mask, B, arch, fixture="vitb32_b256"):
    """BASELINE.json metric, second half: "logits max-abs-err vs HF".  Rank 0's timed batch (weights seed 0, pixels seed
    1000, ids seed 2000, bs=256, ViT-B/32) is exactly the input of tests/golden/vitb32_b256.npz, whose logits come from
    HF CLIPModel itself (oracle/make_golden.py) -- so the error is over ALL 256 x 256 logits.  Other batch sizes /
    architectures have no fixture: they fall back to the CPU oracle on 8 pairs and say so."""
    scale = float(np.exp(np.float64(sd["logit_scale"])))
    path = os.path.join(ROOT, "tests", "golden", fixture + ".npz")
    got = model(input_ids=ids, pixel_values=px, attention_mask=mask)
    lpi = got.logits_per_image.cpu().numpy()
    if arch == "ViT-L/14@336px" and os.path.exists(path) and fixture != "vitb32_b256":
        # the first rows of the ViT-L/14@336 share ARE a committed HF fixture's inputs (oracle/make_golden.py vitl14_336_b8)
        g = np.load(path)
        n = g["ids"].shape[0]
        if n <= B and np.array_equal(g["ids"], ids[:n].cpu().numpy()):
            want = g["logits_per_image"]
            return {"cosine": float(np.abs(lpi[:n, :n] / scale - want / scale).max()),
                    "image_embeds": float(np.abs(got.image_embeds[:n].cpu().numpy() - g["image_embeds"]).max()),
                    "text_embeds": float(np.abs(got.text_embeds[:n].cpu().numpy() - g["text_embeds"]).max()),
                    "vs": f"HF transformers CLIPModel (CPU fp32) golden logits, tests/golden/{fixture}.npz (the batch's first {n} pairs; "
                          f"a row's embedding does not depend on the batch it travels in)",
                    "pairs": n, "logits_compared": n * n}
    if arch == "ViT-B/32" and B == 256 and os.path.exists(path):
        g = np.load(path)
        if np.array_equal(g["ids"], ids.cpu().numpy()):
            want = g["logits_per_image"]
            top2 = np.sort(want / scale, axis=1)[:, -2:]
            clear = (top2[:, 1] - top2[:, 0]) > 2e-3
            return {"cosine": float(np.abs(lpi / scale - want / scale).max()),
                    "cosine_rms": float(np.sqrt(((lpi / scale - want / scale) ** 2).mean())),
                    "scaled_logits": float(np.abs(lpi - want).max()),
                    "image_embeds": float(np.abs(got.image_embeds.cpu().numpy() - g["image_embeds"]).max()),
                    "text_embeds": float(np.abs(got.text_embeds.cpu().numpy() - g["text_embeds"]).max()),
                    "argmax_agreement": float((lpi.argmax(1) == want.argmax(1)).mean()),
                    "argmax_agreement_clear_rows": float((lpi.argmax(1)[clear] == want.argmax(1)[clear]).mean()) if clear.any() else None,
                    "clear_rows": int(clear.sum()),
                    "vs": f"HF transformers CLIPModel (CPU fp32) golden logits, tests/golden/{fixture}.npz",
                    "pairs": 256, "logits_compared": int(want.size)}
    from oracle import clip_oracle as O
    n = min(8, B)
    o = O.clip_forward(px[:n].cpu().numpy(), ids[:n].cpu().numpy(), sd, cfg, mask[:n].cpu().numpy())
    return {"cosine": float(np.abs(lpi[:n, :n] / scale - o["logits_per_image"] / scale).max()),
            "vs": "CPU oracle (numpy fp32 restatement of HF CLIPModel, pinned to HF golden vectors); no HF fixture for this config",
            "pairs": n, "logits_compared": n * n}


def h2d_inclusive(model, cfg, px, ids, mask, steps, warmup, overlap, copy=None):
    """pairs/s of the step when every batch starts in pinned host memory (double-buffered copy stream, overlapped with the towers)"""
    from plip_amd.dist import sharded_pair_logits
    dev = px.device
    B = px.shape[0]
    main = torch.cuda.current_stream(dev)
    out = {}
    # HIP streams share a handful of hardware queues: a copy stream that lands on the queue of one of the tower streams runs its
    # copies IN ORDER with that tower's kernels instead of beside them (profiles/r06_h2d_copy_stream.txt: 4.56 ... 7.5 ms per step
    # depending on which stream the pool hands out).  A high-priority stream lives in a queue class of its own; two ordinary ones are
    # tried beside it and the one that overlaps best on a 4-step trial is kept.
    candidates = [copy] if copy is not None else [torch.cuda.Stream(device=dev, priority=-1), torch.cuda.Stream(device=dev),
                                                   torch.cuda.Stream(device=dev)]
    tiles = torch.from_numpy(np.random.RandomState(7).randint(0, 256, size=(B, cfg.image_size, cfg.image_size, 3), dtype=np.uint8))
    for label, host_img in (("fp32_pixels", px.cpu()), ("u8_tiles", tiles)):
        host = [(host_img.clone().pin_memory(), ids.cpu().pin_memory(), mask.cpu().pin_memory()) for _ in range(2)]
        devb = [tuple(torch.empty_like(t, device=dev) for t in host[0]) for _ in range(2)]
        landed = [torch.cuda.Event() for _ in range(2)]
        consumed = [None, None]

        def fetch(k):
            slot = k & 1
            copy = fetch.stream
            with torch.cuda.stream(copy):
                if consumed[slot] is not None:
                    copy.wait_event(consumed[slot])           # the towers that read this slot's previous batch are done
                for d, h in zip(devb[slot], host[slot]):
                    d.copy_(h, non_blocking=True)
                landed[slot].record(copy)

        def run(n):
            fetch(0)
            for k in range(n):
                if k + 1 < n:
                    fetch(k + 1)                              # batch k+1 crosses PCIe while batch k is in the towers
                slot = k & 1
                main.wait_event(landed[slot])
                sharded_pair_logits(model, *devb[slot], overlap=overlap, equal_shards=True)
                consumed[slot] = torch.cuda.Event()
                consumed[slot].record(main)

        def timed(n):
            torch.cuda.synchronize(dev)
            consumed[0] = consumed[1] = None
            t0 = time.perf_counter()
            run(n)
            torch.cuda.synchronize(dev)
            return (time.perf_counter() - t0) / n

        trial = []
        for st in candidates:
            fetch.stream = st
            run(2)
            trial.append(timed(4))
        fetch.stream = candidates[int(np.argmin(trial))]
        run(max(2, warmup))
        wins = [timed(steps) for _ in range(3)]             # three windows of --steps steps; the median is reported
        dt = float(np.median(wins))
        nbytes = sum(t.numel() * t.element_size() for t in host[0])
        out[label] = {"pairs_per_s": round(B / dt, 1), "ms_per_step": round(dt * 1e3, 3), "host_bytes_per_step": nbytes,
                      "pcie_GBps": round(nbytes / dt / 1e9, 2),
                      "windows_ms": [round(t * 1e3, 3) for t in wins], "copy_stream_trial_ms": [round(t * 1e3, 3) for t in trial]}
        del host, devb
    out["note"] = ("inputs start in pinned host memory every step; a copy stream fills the second of two device buffers while the towers "
                   "run on the first (the pattern of plip_amd/pipeline.py); `value` above is the HBM-resident rate")
    return out


def timed_steps(step, n, dev, warmup=0):
    sync = (lambda: torch.cuda.synchronize(dev)) if dev.type == "cuda" else (lambda: None)
    for _ in range(warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(n):
        out = step()
    sync()
    return (time.perf_counter() - t0) / n, out


def main(argv=None, model_factory=None):
    """``model_factory(cfg, sd, device=, dtype=, max_batch=)``: tests only (with --backend gloo); the product path builds
    ``plip_amd.model.PlipModel`` and fails without an MI355X."""
    args = parse(argv)
    world_size_env = os.environ.get("WORLD_SIZE")
    if world_size_env is None and args.gpus > 1:
        # the driver starts benches as plain `python3 bench.py --gpus N ...`: one process per GPU is then OUR job
        if model_factory is not None:
            raise SystemExit("an in-process model factory cannot follow self-launched ranks: pass --stub-engine file.py:Name")
        rc = self_launch(args, argv)
        if rc:
            raise SystemExit(rc)
        return
    world = int(world_size_env or "1")
    rank = int(os.environ.get("RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")
    if args.stub_engine:
        if args.backend != "gloo" or model_factory is not None:
            raise SystemExit("--stub-engine belongs to the --backend gloo rehearsal (and replaces main(model_factory=...))")
        model_factory = load_stub_factory(args.stub_engine)
    # The contract is ONE JSON line on stdout.  Native libraries write there too (RCCL prints a version banner to fd 1 when a
    # communicator comes up): everything this process emits goes to stderr at the file-descriptor level, except the line itself.
    sys.stdout.flush()
    real_stdout_fd = os.dup(1)
    os.dup2(2, 1)

    def flush_all():
        # C stdio too: RCCL printf()s its version banner into libc's stdout buffer, which is written out whenever libc next
        # flushes it -- at exit, after fd 1 is the caller's again, unless it is pushed out NOW while fd 1 still is stderr
        sys.stdout.flush()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except (OSError, AttributeError):  # pragma: no cover
            pass

    def emit(line):
        flush_all()
        os.dup2(real_stdout_fd, 1)
        print(line, flush=True)
        os.dup2(2, 1)                   # whatever teardown prints is not part of the line either

    try:
        _run(args, world, rank, world_size_env, model_factory, emit)
    finally:                            # in-process callers (tests) get their fd 1 back, and the duplicate does not leak
        flush_all()
        os.dup2(real_stdout_fd, 1)
        os.close(real_stdout_fd)


def _run(args, world, rank, world_size_env, model_factory, emit):
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    rehearsal = args.backend == "gloo"
    if rehearsal and model_factory is None:
        raise SystemExit("--backend gloo is the tests' host-only rehearsal of the multi-rank control flow (it needs a stub "
                         "engine from main(model_factory=...)); the bench itself runs on MI355X GPUs over RCCL")
    if not rehearsal and model_factory is not None:
        raise SystemExit("a model factory is accepted with --backend gloo only")
    if not rehearsal and not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (ROCm) GPU: torch.cuda.is_available() is False; there is no CPU path")
    import torch.distributed as dist
    if rehearsal:
        dev = torch.device("cpu")
        args.no_profile = args.no_extras = args.no_cpu_baseline = True
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)       # "nccl" is RCCL on ROCm

    from plip_amd import weights as W
    from plip_amd.config import get_config
    from plip_amd.dist import sharded_pair_logits
    from plip_amd.model import PlipModel

    cfg = get_config(args.arch)
    B = args.batch
    sd = W.synthetic_state_dict(cfg, seed=0)                   # same weights on every rank
    model = (model_factory or PlipModel)(cfg, sd, device=dev, dtype=args.dtype, max_batch=B)
    # which ranks are REALLY in the job (the driver's SCALE record can check N ranks on N devices were seen): the group's own
    # world size and every rank's device, gathered over the group -- not an echo of --gpus
    my_device = f"rank {rank}: {getattr(model.engine, 'device_name', 'stub')} (cuda:{local_rank})"
    if world > 1:
        rank_devices = [None] * world
        dist.all_gather_object(rank_devices, my_device)
        rccl_ranks = dist.get_world_size()
    else:
        rank_devices, rccl_ranks = [my_device], 1
    px = torch.from_numpy(W.synthetic_pixels(cfg, B, seed=1000 + rank)).to(dev)
    ids_np, mask_np = W.synthetic_ids(cfg, B, seed=2000 + rank)
    ids, mask = torch.from_numpy(ids_np).to(dev), torch.from_numpy(mask_np).to(dev)

    def step():
        return sharded_pair_logits(model, px, ids, mask, overlap=bool(args.overlap), equal_shards=True)

    dev_sync = (lambda: torch.cuda.synchronize(dev)) if dev.type == "cuda" else (lambda: None)

    def fence():
        dev_sync()
        if world > 1:
            dist.barrier()
        dev_sync()

    def window():
        """exactly --steps steps between two fences; max over ranks"""
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            o = step()
        fence()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, o

    for _ in range(args.warmup):
        step()
    elapsed, out = window()                                    # THE measurement
    more = [window()[0] for _ in range(2)]                     # two more windows of the same length, reported beside it
    logits = out[0]
    assert logits.shape == (B, B * world) and bool(torch.isfinite(logits).all())

    # ---- roofline of the dominant kernel: HIP events recorded by the library on the launch stream, over --steps steps on ONE
    # stream -- the kernel owns the GPU, which is what the rocprofv3 kernel trace of the same command measures
    # (profiles/r03_rocprofv3_*).  With two streams an event bracket also contains the time a kernel spends queued behind the
    # other tower's workgroups: that measurement is kept as a side field only.
    def profile_pass(overlap):
        """-> (rows merged per kernel symbol, rows per (kernel, role)); the engine tags GEMM launches 'name|role'."""
        raw = []
        with model.engine.profile(raw):
            for _ in range(args.steps):
                sharded_pair_logits(model, px, ids, mask, overlap=overlap, equal_shards=True)
        merged = {}
        for r in raw:
            name = r["name"].split("|")[0]
            m = merged.setdefault(name, {"name": name, "calls": 0, "total_ms": 0.0, "flops": 0.0, "bytes": 0.0})
            for k in ("calls", "total_ms", "flops", "bytes"):
                m[k] += r[k]
        rows = sorted(merged.values(), key=lambda r: -r["total_ms"])
        return rows, raw

    def roofline_of(rows, name=None):
        dom = next((r for r in rows if r["flops"] > 0 and r["name"].startswith("gemm") and (name is None or r["name"] == name)), None)
        if not dom:
            return None
        ach = dom["flops"] / (dom["total_ms"] * 1e-3) / 1e12
        peak = PEAK_TFLOPS[args.dtype]
        gem = [r for r in rows if r["name"].startswith("gemm")]
        rf = {"bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
              "traffic": None, "kernel": dom["name"], "avg_launch_us": round(dom["total_ms"] * 1e3 / dom["calls"], 2),
              "launches_per_step": dom["calls"] / args.steps,
              "flops_per_launch": round(dom["flops"] / dom["calls"]),
              "all_gemm_tflops": round(sum(r["flops"] for r in gem) / (sum(r["total_ms"] for r in gem) * 1e9), 1)}
        tr = load_pmc_traffic(dom["name"])
        if tr:
            rf["traffic"] = tr["bytes_per_launch"]
            rf["traffic_note"] = tr["note"]
        return rf

    def by_role(raw, kernel):
        """The dominant kernel symbol serves launches with different ceilings (out-proj: 8 B of residual traffic per
        output element for 2*D FLOP -> HBM-side; fc2: at the ridge).  Split its launches by role and price each against
        the bound that applies (ridge = peak FLOP/s / peak B/s)."""
        peak = PEAK_TFLOPS[args.dtype]
        ridge = peak * 1e12 / (HBM_PEAK_GBS * 1e9)
        rows = []
        for r in raw:
            if r["name"].split("|")[0] != kernel or "|" not in r["name"] or not r["total_ms"]:
                continue
            t = r["total_ms"] * 1e-3
            tf, gbs = r["flops"] / t / 1e12, r["bytes"] / t / 1e9
            intensity = r["flops"] / max(r["bytes"], 1.0)
            rows.append({"role": r["name"].split("|")[1], "calls_per_step": r["calls"] / args.steps,
                         "avg_launch_us": round(r["total_ms"] * 1e3 / r["calls"], 2), "tflops": round(tf, 1),
                         "algorithmic_GBps": round(gbs, 1), "flop_per_byte": round(intensity, 1),
                         "bound": "hbm" if intensity < ridge else "mfma",
                         "frac": round(gbs / HBM_PEAK_GBS, 4) if intensity < ridge else round(tf / peak, 4),
                         "frac_mfma": round(tf / peak, 4), "frac_hbm": round(gbs / HBM_PEAK_GBS, 4)})
        return sorted(rows, key=lambda x: -x["calls_per_step"] * x["avg_launch_us"])

    # N > 1: the step's ONE collective on its own -- the stacked [1, 2, B, P] all-gather, 20 back-to-back, max over ranks -- so a scaling
    # curve can be read against it (value(N) = N * B / (t1 + c(N)))
    collective_us = None
    if world > 1:
        from plip_amd.dist import all_gather_rows
        P = cfg.projection_dim
        buf = torch.zeros((1, 2, B, P), dtype=torch.float32, device=dev)
        for _ in range(3):
            all_gather_rows(buf, None, True)
        fence()
        t0 = time.perf_counter()
        for _ in range(20):
            all_gather_rows(buf, None, True)
        dev_sync()
        tc = torch.tensor([(time.perf_counter() - t0) / 20], dtype=torch.float64, device=dev)
        dist.all_reduce(tc, op=dist.ReduceOp.MAX)
        collective_us = round(float(tc.item()) * 1e6, 1)

    roofline = roofline_2s = roles = None
    kernels = []
    one_stream_ms = None
    if not args.no_profile:
        rows1, raw1 = profile_pass(False)
        total = sum(r["total_ms"] for r in rows1) or 1.0
        one_stream_ms = total / args.steps
        kernels = [{"name": r["name"], "calls_per_step": r["calls"] / args.steps,
                    "ms_per_step": round(r["total_ms"] / args.steps, 4), "share": round(r["total_ms"] / total, 4),
                    "tflops": round(r["flops"] / (r["total_ms"] * 1e9), 1) if r["flops"] else None}
                   for r in rows1[:8]]
        roofline = roofline_of(rows1)
        if roofline:
            roofline["mode"] = ("one HIP stream, HIP events on the launch stream: the kernel owns the GPU, as in the rocprofv3 "
                                "kernel trace of the same command")
            roles = by_role(raw1, roofline["kernel"])
            if args.overlap:
                rows2, _ = profile_pass(True)
                roofline_2s = roofline_of(rows2, roofline["kernel"])
                if roofline_2s:
                    roofline_2s["mode"] = ("two HIP streams (the timed configuration): an event bracket includes queueing behind the "
                                           "other tower's workgroups -- not a kernel-quality number")
                    roofline_2s.pop("traffic", None)
                    roofline_2s.pop("traffic_note", None)

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    pairs = B * world * args.steps
    value = pairs / elapsed
    ex = executed_gflop_per_pair(cfg, args.dtype)
    win = [elapsed / args.steps * 1e3] + [m / args.steps * 1e3 for m in more]
    res = {
        "metric": "image+text pairs embedded/sec at 224px bs=256",
        "value": round(value, 1), "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype,
        # structured beside `dtype`: how many LEADING text blocks of the bf16 engine run on f16 MFMA operands
        "text_f16_lead_blocks": int(getattr(model.engine, "text_f16_layers", 0)),
        "rccl_ranks": rccl_ranks, "rank_devices": rank_devices,
        # how the ranks came to be: WORLD_SIZE as this process found it, and who started the ranks -- "self" (bench.py re-executed
        # itself under torch.distributed.run because --gpus N > 1 arrived without a rank environment), "torchrun" (somebody else's
        # launcher set the environment) or "none" (one plain process)
        "world_size_env": world_size_env, "launcher": args.launcher or ("torchrun" if world_size_env is not None else "none"),
        "data": "synthetic" if not rehearsal else "synthetic; gloo REHEARSAL with a stub engine -- control flow only, not a measurement",
        "config": {"workload": f"full dual encoder (image tower + text tower + L2 normalise + logits_per_image), "
                               f"{args.arch}, bs={B} pairs per GPU, {cfg.image_size}px, {cfg.context_length} tokens, "
                               f"{args.dtype} MFMA / fp32 accumulate (BASELINE.json configs[2])" +
                               (f"; the first {model.engine.text_f16_layers} of the {cfg.t_layers} text blocks on f16 MFMA operands "
                                f"(plipmi_config.text_f16_layers, the engine default)" if getattr(model.engine, "text_f16_layers", 0) else ""),
                   "arch": args.arch, "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"dp{world}",
                   "collective": "none" if world == 1 else "RCCL all-gather of [B,512] fp32 image+text embeddings",
                   "collective_us": collective_us,
                   "streams": 2 if args.overlap else 1, "device": model.engine.device_name},
        "windows": {"ms_per_step": [round(w, 3) for w in win], "median": round(float(np.median(win)), 3), "min": round(min(win), 3),
                    "note": "three back-to-back windows of --steps steps each; `value` / `ms_per_step` are the FIRST (the contract's "
                            "exactly-K-steps measurement), the others show the spread"},
        # FLOPs the kernels actually execute (pooled last block) -- THE TFLOP/s figure of this line; and the same rate priced at
        # SURVEY.md section 8d's dense 14.777 GFLOP per pair (work the engine proves unnecessary and skips), for comparison only
        "executed_tflops": round(value * ex["executed"] * 1e9 / 1e12, 2),
        "dense_equivalent_tflops": round(value * cfg.pair_flops() / 1e12, 2),
        "executed_gflop_per_pair": ex,
        "roofline": roofline,
        "roofline_two_stream_events": roofline_2s,
        "roofline_by_role": {"note": "launches of the dominant kernel symbol (one stream) split by what they compute, each priced against "
                                     "the bound its arithmetic intensity puts it under (ridge 312 FLOP/B at 2.5 PF / 8 TB/s); bytes are "
                                     "algorithmic (operands once, fp32 residual stream read + written as two 16-bit planes)",
                             "roles": roles} if roles else None,
        "one_stream_sum_of_kernel_ms": round(one_stream_ms, 3) if one_stream_ms else None,
        "kernels": kernels,
    }
    try:
        from plip_amd.build import source_digest
        res["csrc_sha16"] = source_digest()
    except Exception:  # pragma: no cover
        pass
    extras = world == 1 and not args.no_extras and args.dtype != "f32"
    if world == 1 and not args.no_cpu_baseline:
        try:
            res["logits_max_abs_err"] = logits_error_vs_hf_golden(model, cfg, sd, px, ids, mask, B, args.arch)
        except Exception as e:  # pragma: no cover
            res["logits_max_abs_err"] = {"error": repr(e)}

    def side_engine(dtype, **kw):
        return PlipModel(cfg, sd, device=dev, dtype=dtype, max_batch=B, **kw)

    if extras:
        # the OTHER 16-bit engine on the same step: bf16 is the headline BASELINE.json names; f16 (IEEE half, the reference's
        # own GPU dtype) has 8x smaller operand rounding at the same matrix-core rate
        other = "f16" if args.dtype == "bf16" else "bf16"
        try:
            mo = side_engine(other)
            dto, _ = timed_steps(lambda: sharded_pair_logits(mo, px, ids, mask, overlap=bool(args.overlap), equal_shards=True),
                                 args.steps, dev, args.warmup)
            res[other] = {"note": f"the same step on the {other} engine (compute_dtype PLIPMI_{other.upper()})",
                          "pairs_per_s": round(B / dto, 1), "ms_per_step": round(dto * 1e3, 3),
                          "logits_max_abs_err": logits_error_vs_hf_golden(mo, cfg, sd, px, ids, mask, B, args.arch)}
            mo.engine.close()
            del mo
        except Exception as e:  # pragma: no cover
            res[other] = {"error": repr(e)}
        # the bf16 engine's precision dial (plipmi_config.text_f16_layers): how many LEADING text blocks run on IEEE-half operands.
        # The headline engine uses the default; 0 = pure bf16, t_layers = the whole text tower (PLIPMI_FLAG_TEXT_TOWER_F16's engine)
        if args.dtype == "bf16":
            dial = {}
            for n in sorted({0, 4, 8, cfg.t_layers} - {model.engine.text_f16_layers}):
                try:
                    mm = side_engine("bf16", text_f16_layers=n)
                    dtm, _ = timed_steps(lambda: sharded_pair_logits(mm, px, ids, mask, overlap=bool(args.overlap), equal_shards=True),
                                         args.steps, dev, args.warmup)
                    dial[str(n)] = {"pairs_per_s": round(B / dtm, 1), "ms_per_step": round(dtm * 1e3, 3),
                                    "logits_max_abs_err": logits_error_vs_hf_golden(mm, cfg, sd, px, ids, mask, B, args.arch)}
                    mm.engine.close()
                    del mm
                except Exception as e:  # pragma: no cover
                    dial[str(n)] = {"error": repr(e)}
            res["text_f16_layers"] = {
                "note": "the same step on bf16 engines with that many leading text blocks on f16 operands (image tower bf16 in all of "
                        f"them); the headline engine runs {model.engine.text_f16_layers}", "engines": dial}
        # A/B: the same step with the last block computed on EVERY token (as HF does), i.e. the dense 14.777 GFLOP per pair
        try:
            md = side_engine(args.dtype, pooled_last_block=False)
            dtd, od = timed_steps(lambda: sharded_pair_logits(md, px, ids, mask, overlap=bool(args.overlap), equal_shards=True),
                                  args.steps, dev, args.warmup)
            res["dense_last_block"] = {
                "note": "same step on an engine built with PLIPMI_FLAG_DENSE_LAST_BLOCK: the last block's out_proj / fc1 / fc2 on all "
                        "tokens instead of the pooled row only (identical embeddings up to fp32 summation order)",
                "pairs_per_s": round(B / dtd, 1), "ms_per_step": round(dtd * 1e3, 3),
                "max_abs_diff_of_logits_vs_pooled_path": float((od[0] - logits).abs().max())}
            md.engine.close()
            del md
        except Exception as e:  # pragma: no cover
            res["dense_last_block"] = {"error": repr(e)}
        # A/B, NOT the headline: the same step with the text tower on the captions' live rows only (plipmi_set_text_packing).
        # The headline above executes every padded position of the 77-token context, as the reference does.
        try:
            model.engine.set_text_packing(True)
            dtp, op = timed_steps(step, args.steps, dev, args.warmup)
            live = float(mask.sum(dim=1).float().mean().item())          # BOS .. EOS inclusive
            res["packed_captions"] = {
                "note": "same step with plipmi_set_text_packing(1): rows behind each caption's EOS token are not computed (causal "
                        "tower + EOS pooling: they cannot reach text_embeds); embeddings bit-identical to the padded step. Opt-in; "
                        "the headline value is the padded computation",
                "pairs_per_s": round(B / dtp, 1), "ms_per_step": round(dtp * 1e3, 3),
                "mean_live_tokens_per_caption": round(live, 1), "context_length": cfg.context_length,
                "max_abs_diff_of_logits_vs_padded": float((op[0] - logits).abs().max())}
        except Exception as e:  # pragma: no cover
            res["packed_captions"] = {"error": repr(e)}
        finally:
            model.engine.set_text_packing(False)
        # larger per-GPU batches of the same step (inputs generated on the device).  The engine runs a call of B >= 2 * pass_batch samples
        # as equal back-to-back passes (plipmi_config.pass_batch, 256 for this model: one pass's activations stay in the Infinity Cache);
        # `one_pass` is the A/B with the splitting off (pass_batch = -1: round 5's behaviour)
        try:
            scal = {}
            for Bb in (512, 1024):
                g = torch.Generator(device=dev).manual_seed(5)
                pxb = torch.randn((Bb, 3, cfg.image_size, cfg.image_size), generator=g, device=dev)
                ib, mbk = W.synthetic_ids(cfg, Bb, seed=2000)
                ib, mbk = torch.from_numpy(ib).to(dev), torch.from_numpy(mbk).to(dev)
                # the two engines interleaved, three rounds each: the first window behind an engine's creation runs slower than its later ones
                arms = {"passes": PlipModel(cfg, sd, device=dev, dtype=args.dtype, max_batch=Bb),
                        "one_pass": PlipModel(cfg, sd, device=dev, dtype=args.dtype, max_batch=Bb, pass_batch=-1)}
                times = {k: [] for k in arms}
                for rnd in range(3):
                    for label, mb in arms.items():
                        dtb, _ = timed_steps(lambda: sharded_pair_logits(mb, pxb, ib, mbk, overlap=bool(args.overlap), equal_shards=True),
                                             max(4, args.steps // 2), dev, 2)
                        times[label].append(dtb)
                row = {k: {"pairs_per_s": round(Bb / float(np.median(v)), 1), "ms_per_step": round(float(np.median(v)) * 1e3, 3),
                           "rounds_ms": [round(t * 1e3, 3) for t in v]} for k, v in times.items()}
                for mb in arms.values():
                    mb.engine.close()
                del arms
                scal[f"bs{Bb}"] = {**row["passes"], "one_pass": row["one_pass"]}
                del pxb
            res["batch_scaling"] = scal
        except Exception as e:  # pragma: no cover
            res["batch_scaling"] = {"error": repr(e)}
        # "report with and without H2D of inputs".  The same step with the inputs starting in PINNED HOST memory: a
        # copy stream moves batch k+1 (pixels + ids + mask) into the second of two device buffers while the towers run on batch k -- what
        # plip_amd/pipeline.py does for PLIP.encode_images.  fp32 pixels (154 MB per step: what the reference hands over, plip.py:49) and
        # native uint8 tiles (38.5 MB; normalisation fused on the GPU).  Never `value`.
        try:
            res["h2d_inclusive"] = h2d_inclusive(model, cfg, px, ids, mask, args.steps, args.warmup, bool(args.overlap))
        except Exception as e:  # pragma: no cover
            res["h2d_inclusive"] = {"error": repr(e)}
        # ONE tower over many batches (the image side of BASELINE.json configs[3]; PLIP.encode_images, plip.py:41-52): a handle runs one
        # batch per tower at a time, so the product's host loops give consecutive batches to the engine and to a plipmi_clone of it (same
        # packed weights, a workspace of its own) on two streams (Engine.lane_loop) -- one batch's launch boundaries, epilogues and pooled
        # tail run under the next batch's GEMMs.  `one_lane` = the plain loop; arms interleaved, median of three rounds of 40 batches.
        try:
            eng = model.engine
            tiles_c = [torch.randint(0, 256, (B, cfg.image_size, cfg.image_size, 3), dtype=torch.uint8, device=dev) for _ in range(2)]

            def corpus(two_lanes, nb=40):
                eng.use_lanes = two_lanes
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                with eng.lane_loop() as run:
                    outs = [run(lambda e, k=k: e.encode_image_u8(tiles_c[k & 1], True)) for k in range(nb)]
                torch.cuda.synchronize(dev)
                return (time.perf_counter() - t0) / nb, outs

            same_bits = all(torch.equal(a, b) for a, b in zip(corpus(False, 4)[1], corpus(True, 4)[1]))
            rounds = {False: [], True: []}
            for _ in range(3):
                for arm in (False, True):
                    rounds[arm].append(corpus(arm)[0])
            eng.use_lanes = True
            t1, t2 = sorted(rounds[False])[1], sorted(rounds[True])[1]
            res["image_corpus_lanes"] = {
                "workload": f"image tower only, uint8 {cfg.image_size}px tiles resident in HBM, 40 consecutive batches of {B}",
                "two_lanes": {"images_per_s": round(B / t2, 1), "ms_per_batch": round(t2 * 1e3, 3), "rounds_ms": [round(t * 1e3, 3) for t in rounds[True]]},
                "one_lane": {"images_per_s": round(B / t1, 1), "ms_per_batch": round(t1 * 1e3, 3), "rounds_ms": [round(t * 1e3, 3) for t in rounds[False]]},
                "bit_identical": same_bits}
        except Exception as e:  # pragma: no cover
            res["image_corpus_lanes"] = {"error": repr(e)}
        # one GPU's share of BASELINE.json configs[4] (ViT-L/14@336, bs=512 over 8 GPUs = 64 pairs per GPU), same dtype
        try:
            cl = get_config("ViT-L/14@336px")
            sdl = W.synthetic_state_dict(cl, seed=3)
            ml = PlipModel(cl, sdl, device=dev, dtype=args.dtype, max_batch=64)
            # rows 0..7 = the inputs of tests/golden/vitl14_336_b8.npz (HF outputs for exactly these pixels / ids), the rest device noise
            g = torch.Generator(device=dev).manual_seed(6)
            pxl = torch.randn((64, 3, cl.image_size, cl.image_size), generator=g, device=dev)
            pxl[:8] = torch.from_numpy(W.synthetic_pixels(cl, 8, seed=6000)).to(dev)
            il, mk = W.synthetic_ids(cl, 64, seed=42)
            i8, m8 = W.synthetic_ids(cl, 8, seed=6001)
            il[:8], mk[:8] = i8, m8
            il, mk = torch.from_numpy(il).to(dev), torch.from_numpy(mk).to(dev)
            dtl, _ = timed_steps(lambda: sharded_pair_logits(ml, pxl, il, mk, overlap=bool(args.overlap), equal_shards=True), 5, dev, 2)
            res["vitl14_336_b64"] = {
                "workload": "ViT-L/14@336 dual encoder, 64 pairs (one GPU's share of configs[4]'s bs=512 on 8 GPUs), same step; parity "
                            "of this architecture: tests/test_gpu_configs.py against HF (tests/golden/vitl14_336_b2.npz)",
                "pairs_per_s": round(64 / dtl, 1), "ms_per_step": round(dtl * 1e3, 3),
                "dense_equivalent_tflops": round(64 * cl.pair_flops() / dtl / 1e12, 1),
                "dense_equivalent_frac_of_mfma_peak": round(64 * cl.pair_flops() / dtl / 1e12 / PEAK_TFLOPS[args.dtype], 4)}
            if not args.no_cpu_baseline:
                # this very batch's first pairs against the CPU oracle (batch-invariant engine: the same bits as inside the 64)
                res["vitl14_336_b64"]["logits_max_abs_err"] = logits_error_vs_hf_golden(ml, cl, sdl, pxl, il, mk, 64, "ViT-L/14@336px",
                                                                                        fixture="vitl14_336_b8")
            ml.engine.close()
            del ml, pxl
        except Exception as e:  # pragma: no cover
            res["vitl14_336_b64"] = {"error": repr(e)}
        # BASELINE.json configs[1]: ViT-B/32 image tower only, bs=256, fp32 (exact-fp32 MFMA engine), same pixels
        try:
            m32 = side_engine("f32")
            n32 = max(3, args.steps // 4)
            dt32, e32 = timed_steps(lambda: m32.engine.encode_image(px, True), n32, dev, 2)
            rows32 = []
            with m32.engine.profile(rows32):
                m32.engine.encode_image(px, True)
            rows32.sort(key=lambda r: -r["total_ms"])
            dom32 = next((r for r in rows32 if r["flops"] > 0), None)
            # the fp32 PAIR rate (both towers + logits on the fp32 engine), same batch
            dtp, _ = timed_steps(lambda: sharded_pair_logits(m32, px, ids, mask, overlap=bool(args.overlap), equal_shards=True), n32, dev, 1)
            e16 = model.engine.encode_image(px, True)
            res["fp32_pairs"] = {"workload": "full dual encoder bs=256 on the exact-fp32 MFMA engine (v_mfma_f32_32x32x2_f32)",
                                 "pairs_per_s": round(B / dtp, 1), "ms_per_step": round(dtp * 1e3, 3),
                                 "dense_tflops": round(B * cfg.pair_flops() / dtp / 1e12, 2),
                                 "frac_of_fp32_mfma_peak": round(B * cfg.pair_flops() / dtp / 1e12 / PEAK_TFLOPS["f32"], 4)}
            res["config1_fp32_image_tower"] = {
                "workload": "ViT-B/32 image tower only, bs=256, fp32 (v_mfma_f32_32x32x2_f32), synthetic 224px tiles",
                "images_per_s": round(B / dt32, 1), "ms_per_step": round(dt32 * 1e3, 3),
                "dense_tflops": round(B * cfg.image_flops() / dt32 / 1e12, 2),
                "roofline": None if dom32 is None else {
                    "bound": "mfma", "kernel": dom32["name"], "peak": PEAK_TFLOPS["f32"], "unit": "TFLOP/s",
                    "achieved": round(dom32["flops"] / (dom32["total_ms"] * 1e9), 2),
                    "frac": round(dom32["flops"] / (dom32["total_ms"] * 1e9) / PEAK_TFLOPS["f32"], 4)},
                f"{args.dtype}_vs_fp32_embedding_max_abs_diff": float((e16 - e32).abs().max())}
            m32.engine.close()
            del m32
        except Exception as e:  # pragma: no cover
            res["config1_fp32_image_tower"] = {"error": repr(e)}
    if extras and res.get("roofline"):
        # `roofline.traffic` as an OBSERVATION of this box (the committed, source-stamped PMC summary is a claim about another
        # one): two rocprofv3 --pmc passes over a short one-stream run of this script, now.  The committed figure stays beside it.
        try:
            live = live_pmc_traffic(res["roofline"]["kernel"])
        except Exception:  # pragma: no cover
            live = None
        if live:
            res["roofline"]["traffic_committed_summary"] = res["roofline"].get("traffic")
            res["roofline"]["traffic"] = live["bytes_per_launch"]
            res["roofline"]["traffic_note"] = live["note"]
            res["roofline"]["traffic_fetch_write"] = [live["fetch_bytes"], live["write_bytes"]]
    if extras:
        # The step's ONE collective on this box's single GPU: a one-rank RCCL group (communicator on this device, the stacked
        # [1, 2, B, P] all_gather_into_tensor enqueued behind the two tower streams every step).  Not a scaling number -- the cost
        # of issuing the collective, which is what an N-rank step adds before any wire time.
        try:
            import socket
            with socket.socket() as so:
                so.bind(("127.0.0.1", 0))
                port = so.getsockname()[1]
            dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
            try:
                run = lambda ac: sharded_pair_logits(model, px, ids, mask, overlap=bool(args.overlap), equal_shards=True,
                                                     always_collective=ac)
                o1 = run(True)
                same = bool(torch.equal(o1[0], run(False)[0]))
                # three interleaved windows per arm, medians: one window each used to put a host hiccup into the difference
                # (a 5.37 ms "without" window against 4.27 with -> "-1108 us")
                w_c, w_p = [], []
                for _ in range(3):
                    w_c.append(timed_steps(lambda: run(True), args.steps, dev, 2)[0])
                    w_p.append(timed_steps(lambda: run(False), args.steps, dev, 2)[0])
                t_c, t_p = sorted(w_c)[1], sorted(w_p)[1]
                res["rccl_one_rank"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                                        "ms_per_step_with_all_gather": round(t_c * 1e3, 3), "ms_per_step_without": round(t_p * 1e3, 3),
                                        "windows_ms": {"with": [round(t * 1e3, 3) for t in w_c], "without": [round(t * 1e3, 3) for t in w_p]},
                                        "all_gather_cost_us": round((t_c - t_p) * 1e6, 1), "logits_bit_identical": same,
                                        "collective": f"all_gather_into_tensor of [1, 2, {B}, {cfg.projection_dim}] fp32 "
                                                      f"({2 * B * cfg.projection_dim * 4 // 1024} KiB per rank)"}
            finally:
                dist.destroy_process_group()
        except Exception as e:  # pragma: no cover
            res["rccl_one_rank"] = {"error": repr(e)}
    res["cpu_baseline"] = cpu_baseline(cfg, sd, args.cpu_seconds) if (world == 1 and not args.no_cpu_baseline) else None
    emit(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
