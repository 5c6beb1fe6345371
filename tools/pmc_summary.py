"""Summarise rocprofv3 --pmc passes into per-kernel HBM/fabric traffic (bytes per launch).

    python tools/pmc_summary.py gpurun_out/pmc_bench_FETCH_SIZE gpurun_out/pmc_bench_WRITE_SIZE > profiles/pmc_traffic.json

FETCH_SIZE / WRITE_SIZE are reported in KiB-ish units of 1 KB by rocprofv3; on gfx950 FETCH_SIZE counts 128-byte
requests as 64 bytes for wide coalesced reads, so the read side is doubled (MI355X_MICROARCH.md, HBM section).
Kernel symbols are mapped to the names the engine's own profile uses (gemm_nt<dtype,tile,epilogue>).
"""
import collections
import csv
import json
import re
import sys

EPI = {"0": "bias", "1": "bias_qgelu", "2": "bias_resid", "3": "scale", "4": "patch", "5": "ln_bias", "6": "ln_qgelu",
       "7": "resid_emit", "8": "resid_split"}


def pretty(sym, variants=None):
    """gemm_nt_kernel<T, BM, BN, WM, WN, EPI, SCHED, ADDR, NSTAGE> (mangled) -> the engine's profile name gemm_nt<dtype,tile,epilogue>"""
    m = re.search(r"gemm_nt_kernelI(DF16b|DF16_|Dh|f)Li(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E", sym)
    if not m:
        return sym
    dt = {"DF16b": "bf16", "DF16_": "f16", "Dh": "f16", "f": "f32"}[m.group(1)]
    bm, bn, wm, wn, epi, addr = m.group(2), m.group(3), m.group(4), m.group(5), m.group(6), m.group(8)
    tile = f"{bm}x{bn}_w{wm}x{wn}_" + ("ring3" if m.group(9) == "3" else "bufdma" if addr == "1" else "glds64")
    return f"gemm_nt<{dt},{tile},{EPI.get(epi, epi)}>"


VARIANTS = None   # (the tile name follows from the template arguments since round 3)


def main():
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in sys.argv[1:]:
        with open(d + "/pmc_counter_collection.csv") as f:
            for r in csv.DictReader(f):
                acc[pretty(r["Kernel_Name"], VARIANTS)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for k, c in acc.items():
        if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c or not k.startswith("gemm_nt<"):
            continue
        fetch = 2.0 * 1024.0 * sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"])     # gfx950: x2, KB -> bytes
        write = 1024.0 * sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"])
        out[k] = {"bytes_per_launch": round(fetch + write), "fetch_bytes": round(fetch), "write_bytes": round(write),
                  "launches": len(c["FETCH_SIZE"]),
                  "note": "rocprofv3 --pmc FETCH_SIZE (x2 gfx950 correction) + WRITE_SIZE, separate passes, "
                          "mean over the launches of `bench.py --steps 3` (both stream modes)"}
    # stamp: the kernel sources these counters were collected on (bench.py emits `traffic: null` on a mismatch)
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from plip_amd.build import source_digest
    out["_stamp"] = {"csrc_sha16": source_digest(), "how": "tools/gpu_round.sh pmcbench (rocprofv3 --kernel-trace --pmc, one counter per pass)"}
    json.dump(out, sys.stdout, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
