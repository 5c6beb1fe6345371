"""Build libplipmi.so (the HIP/gfx950 engine) in-tree with hipcc.

    python -m plip_amd.build            # incremental, parallel over translation units
    python -m plip_amd.build --force

hipcc cross-compiles gfx950 code objects without a GPU.  The shared object lands next
to the sources (plip_amd/csrc/libplipmi.so): git-ignored, but shipped to the GPU box.
"""
from __future__ import annotations

import concurrent.futures as cf
import hashlib
import os
import shutil
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
BUILD = os.path.join(CSRC, "build")
LIB = os.path.join(CSRC, "libplipmi.so")
SOURCES = ["engine.hip", "gemm.hip", "gemm_f32.hip", "gemm_bf16.hip", "gemm_f16.hip", "gemm_skinny.hip", "kernels.hip",
           "attention.hip", "attention_mfma.hip", "qkv_attention.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-Wno-unused-result", "-Wno-unused-value"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm; set HIPCC=/path/to/hipcc)")


def _digest(paths) -> str:
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for p in sorted(paths):
        with open(p, "rb") as f:
            h.update(p.encode())
            h.update(f.read())
    return h.hexdigest()


def source_digest() -> str:
    """sha256 (first 16 hex digits) over every kernel source and header of libplipmi.so: the identity of the code a
    measurement was taken on.  bench.py stamps its line with it and refuses profiles/pmc_traffic.json when that file
    was collected on different sources."""
    paths = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hip"))]
    for hdr in ("plipmi.h", "plipmi_test.h"):
        paths.append(os.path.join(os.path.dirname(os.path.dirname(CSRC)), "include", hdr))
    h = hashlib.sha256()
    for p in sorted(paths):
        with open(p, "rb") as f:
            h.update(os.path.basename(p).encode())
            h.update(f.read())
    return h.hexdigest()[:16]


def _compile(hipcc: str, src: str, obj: str) -> str:
    cmd = [hipcc, *FLAGS, "-I", CSRC, "-c", os.path.join(CSRC, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    return r.stderr


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = _hipcc()
    os.makedirs(BUILD, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    for hdr in ("plipmi.h", "plipmi_test.h"):
        headers.append(os.path.join(os.path.dirname(os.path.dirname(CSRC)), "include", hdr))
    hdr_digest = _digest(headers)
    jobs = []
    objs = []
    for src in SOURCES:
        obj = os.path.join(BUILD, src.replace(".hip", ".o"))
        stamp = obj + ".sha"
        want = _digest([os.path.join(CSRC, src)]) + hdr_digest
        objs.append(obj)
        have = open(stamp).read() if os.path.exists(stamp) else ""
        if force or have != want or not os.path.exists(obj):
            jobs.append((src, obj, stamp, want))
    if jobs:
        if verbose:
            print(f"[plip_amd.build] hipcc {len(jobs)} translation unit(s) for gfx950 ...", flush=True)
        with cf.ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            futs = {ex.submit(_compile, hipcc, src, obj): (src, stamp, want) for src, obj, stamp, want in jobs}
            for fut in cf.as_completed(futs):
                src, stamp, want = futs[fut]
                warn = fut.result()
                if verbose and warn.strip():
                    print(warn, file=sys.stderr)
                with open(stamp, "w") as f:
                    f.write(want)
    if jobs or not os.path.exists(LIB):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[plip_amd.build] linked {LIB} ({os.path.getsize(LIB) / 1e6:.1f} MB)", flush=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
