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


def pretty(sym, variants):
    m = re.search(r"gemm_nt_kernelI(DF16b|f)Li(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb([01])(?:ELi(\d+))?(?:ELi(\d+))?(?:ELi(\d+))?(?:ELi(\d+))?", sym)
    if not m:
        return sym
    dt = "bf16" if m.group(1) == "DF16b" else "f32"
    bm, bn, wm, wn, epi, glds = m.group(2), m.group(3), m.group(4), m.group(5), m.group(6), m.group(7)
    sched, l2pf, nst = m.group(8) or "0", m.group(9) or "0", m.group(10) or "2"
    addr = m.group(11) or "0"
    key = (bm, bn, glds, sched, l2pf, nst)
    tile = variants.get(key, f"{bm}x{bn}_w{wm}x{wn}_g{glds}s{sched}p{l2pf}n{nst}")
    if addr == "1":
        tile = tile.replace("_glds", "_bufdma")
    return f"gemm_nt<{dt},{tile},{EPI.get(epi, epi)}>"


VARIANTS = {("128", "128", "0", "0", "0", "2"): "128x128_w2x2_regstage", ("128", "128", "1", "0", "0", "2"): "128x128_w2x2_glds",
            ("256", "256", "1", "0", "0", "2"): "256x256_w4x2_glds", ("256", "256", "1", "1", "0", "2"): "256x256_w4x2_glds_fragpipe",
            ("128", "128", "1", "1", "0", "2"): "128x128_w2x2_glds_fragpipe", ("256", "128", "1", "1", "0", "2"): "256x128_w4x2_glds_fragpipe",
            ("256", "256", "1", "3", "0", "2"): "256x256_w4x2_glds_spreadfill", ("128", "128", "1", "3", "0", "2"): "128x128_w2x2_glds_spreadfill",
            ("192", "256", "1", "3", "0", "2"): "192x256_w2x4_glds_spreadfill", ("192", "256", "1", "1", "0", "2"): "192x256_w2x4_glds_fragpipe",
            ("320", "256", "1", "3", "0", "2"): "320x256_w2x4_glds_spreadfill", ("320", "256", "1", "0", "0", "2"): "320x256_w2x4_glds",
            ("256", "256", "1", "5", "0", "2"): "256x256_w4x2_glds_fill2", ("320", "256", "1", "5", "0", "2"): "320x256_w2x4_glds_fill2",
            ("192", "256", "1", "5", "0", "2"): "192x256_w2x4_glds_fill2", ("256", "256", "1", "6", "0", "2"): "256x256_w4x2_glds_fill3",
            ("320", "256", "1", "6", "0", "2"): "320x256_w2x4_glds_fill3", ("192", "256", "1", "6", "0", "2"): "192x256_w2x4_glds_fill3"}


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
