"""Concurrency summary of a rocprofv3 --kernel-trace CSV: how much of the wall time the GPU runs 0 / 1 / 2+ kernels, the
sum of kernel durations, launches per step, and the gaps between consecutive kernels of one stream.

    python tools/trace_overlap.py gpurun_out/prof/plip_kernel_trace.csv [steps]
"""
import csv
import sys


def main():
    path = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), r["Kernel_Name"]))
    rows.sort()
    # keep the steady part: the last 60 % of the launches (after warm-up / weight packing)
    rows = rows[int(len(rows) * 0.4):]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    ev = []
    for s, e, _, _ in rows:
        ev.append((s, 1)); ev.append((e, -1))
    ev.sort()
    depth, last, hist = 0, t0, {}
    for t, d in ev:
        hist[min(depth, 3)] = hist.get(min(depth, 3), 0) + (t - last)
        depth += d; last = t
    wall = t1 - t0
    tot = sum(e - s for s, e, _, _ in rows)
    print(f"{len(rows)} launches over {wall / 1e6:.3f} ms; sum of kernel durations {tot / 1e6:.3f} ms ({tot / wall:.2f}x the wall time)")
    for k in sorted(hist):
        print(f"   {k}{'+' if k == 3 else ''} kernels in flight: {100 * hist[k] / wall:5.1f} % of the wall time")
    per_q = {}
    for s, e, q, n in rows:
        per_q.setdefault(q, []).append((s, e))
    for q, iv in per_q.items():
        gaps = sorted(b[0] - a[1] for a, b in zip(iv, iv[1:]))
        if gaps:
            print(f"   queue {q}: {len(iv)} launches, busy {sum(e - s for s, e in iv) / 1e6:.3f} ms, gap between consecutive launches "
                  f"median {gaps[len(gaps) // 2] / 1e3:.2f} us, p90 {gaps[int(len(gaps) * 0.9)] / 1e3:.2f} us, total {sum(g for g in gaps if g > 0) / 1e6:.3f} ms")
    if steps:
        print(f"   launches per step: {len(rows) / (0.6 * steps):.1f}")


if __name__ == "__main__":
    main()
