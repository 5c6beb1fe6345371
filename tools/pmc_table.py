"""Per-kernel table of hardware counters from several rocprofv3 --kernel-trace --pmc passes over the same command.

    python tools/pmc_table.py gpurun_out/pmcstep_* > profiles/r02_pmc_step_counters.txt

Each pass directory holds pmc_counter_collection.csv (one row per dispatch and counter, with the dispatch's start / end
timestamps).  Counters are averaged per launch and per kernel (engine naming, tools/pmc_summary.py); derived lines follow
MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are in KB and the read side is doubled on gfx950; SQ_* wave counters are
quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES is cycles summed over SIMDs (1024 on the chip); GRBM_GUI_ACTIVE is
summed over the 8 XCDs.
"""
import collections
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import VARIANTS, pretty  # noqa: E402


def short(sym):
    s = pretty(sym, VARIANTS)
    if s != sym:
        return s
    for key in ("qkv_attention_kernel", "gemm_skinny_kernel", "pool_gather_kernel", "attention_mfma_kernel", "attention_flash_kernel", "attention_valu_kernel", "layernorm_emit_kernel",
                "layernorm_fixed_kernel", "layernorm_kernel", "text_embed_emit_kernel", "text_embed_kernel", "unfold_kernel",
                "unfold_u8_kernel", "head_gemm_kernel", "pool_layernorm_kernel", "l2_normalize_kernel", "cls_rows_kernel",
                "logits_kernel", "row_argmax_kernel"):
        if key in sym:
            tail = sym.split(key, 1)[1]
            arg = tail[1:tail.index("E")] if tail.startswith("I") and "E" in tail else ""
            arg = arg.replace("DF16b", "bf16,").replace("DF16_", "f16,").replace("Li", "").replace("E", ",").strip(",")
            return key.replace("_kernel", "") + (f"<{arg}>" if arg else "")
    return None


def main():
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for d in sys.argv[1:]:
        path = os.path.join(d, "pmc_counter_collection.csv")
        if not os.path.exists(path):
            continue
        seen = set()
        with open(path) as f:
            for r in csv.DictReader(f):
                k = short(r["Kernel_Name"])
                if k is None:
                    continue
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                if (d, r["Dispatch_Id"]) not in seen:
                    seen.add((d, r["Dispatch_Id"]))
                    dur[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    order = sorted(acc, key=lambda k: -sum(dur[k]))
    per_step = []
    for k in order:
        c = {n: sum(v) / len(v) for n, v in acc[k].items()}
        t_ns = sum(dur[k]) / len(dur[k])
        n_launch = max(len(v) for v in acc[k].values())
        print(f"{k}   ({n_launch} launches per pass, mean duration under the profiler {t_ns / 1e3:.1f} us)")
        for n in sorted(c):
            print(f"   {n:32s} {c[n]:12.4g}")
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            b = 2.0 * 1024 * c["FETCH_SIZE"] + 1024 * c["WRITE_SIZE"]
            print(f"   => HBM/fabric traffic {b / 1e6:.1f} MB per launch ({2.0 * 1024 * c['FETCH_SIZE'] / 1e6:.1f} read, x2 gfx950 correction; "
                  f"{1024 * c['WRITE_SIZE'] / 1e6:.1f} written) = {b / t_ns / 1e3:.2f} TB/s")
            per_step.append((k, len(acc[k]["FETCH_SIZE"]), b))
        if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"] > 0:
            w = c["SQ_WAVE_CYCLES"]
            print(f"   => wave time: WAIT_ANY {100 * c.get('SQ_WAIT_ANY', 0) / w:.0f} %, WAIT_INST_ANY {100 * c.get('SQ_WAIT_INST_ANY', 0) / w:.0f} %, "
                  f"ACTIVE {100 * c.get('SQ_ACTIVE_INST_ANY', 0) / w:.0f} %; LDS bank-conflict cycles {c.get('SQ_LDS_BANK_CONFLICT', 0):.0f}")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c and c["GRBM_GUI_ACTIVE"] > 0:
            # GRBM_GUI_ACTIVE spans more than the kernel's own timestamps (the round-2 table derived an "effective clock" of
            # 2.4-2.7 GHz from it, above the chip's maximum): it is used as the matrix pipes' time base only; the shader
            # clock under load comes from the in-kernel s_memtime / s_memrealtime stamps (profiles/r03_gemm_kloop.txt)
            clk = c["GRBM_GUI_ACTIVE"] / 8.0
            print(f"   => MFMA pipe busy {100 * c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * clk):.1f} % of the matrix pipes' time (GRBM_GUI_ACTIVE / 8 as the time base)")
        print()
    # bytes per step: the patch GEMM runs once per step, so its launch count is the number of steps the profiled command ran
    steps = [n for k, n, _ in per_step if k.endswith(",patch>")]
    if steps:
        print(f"# ---- bytes per step (launches per step x (FETCH_SIZE x 2 + WRITE_SIZE) per launch; the profiled pass ran {steps[0]} steps) ----")
        rows = sorted(((k, n / steps[0], b) for k, n, b in per_step), key=lambda r: -r[1] * r[2])
        for k, n, b in rows:
            print(f"#   {k:56s} {n:6.1f} x {b / 1e6:7.1f} MB = {n * b / 1e9:6.3f} GB")
        print(f"#   TOTAL {sum(n * b for _, n, b in rows) / 1e9:.2f} GB per step")


if __name__ == "__main__":
    main()
