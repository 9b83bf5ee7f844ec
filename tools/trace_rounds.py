"""Per-launch durations and gaps of the row reduction's round kernel from a rocprofv3 kernel trace (developer tool).
usage: trace_rounds.py <kernel_trace.csv> [kernel-name-substring]"""
import csv
import sys

import numpy as np


def main():
    path = sys.argv[1]
    key = sys.argv[2] if len(sys.argv) > 2 else "wide_sc_round"
    rows = [r for r in csv.DictReader(open(path))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    sel = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if key in r["Kernel_Name"]]
    if not sel:
        print("no such kernel")
        return
    # the LAST solve's launches: split at gaps > 1 ms
    runs, cur = [], [sel[0]]
    for a, b in zip(sel, sel[1:]):
        if b[0] - a[1] > 1_000_000:
            runs.append(cur)
            cur = []
        cur.append(b)
    runs.append(cur)
    run = runs[-1]
    d = np.array([e - s for s, e in run]) / 1e3
    g = np.array([b[0] - a[1] for a, b in zip(run, run[1:])]) / 1e3
    print(f"{key}: {len(run)} launches in the last solve; span {(run[-1][1] - run[0][0]) / 1e6:.3f} ms")
    print(f"  duration us: sum {d.sum() / 1e3:.3f} ms  median {np.median(d):.2f}  p10 {np.percentile(d, 10):.2f}  p90 {np.percentile(d, 90):.2f}  max {d.max():.1f}")
    print(f"  gap us:      sum {g.sum() / 1e3:.3f} ms  median {np.median(g):.2f}  p10 {np.percentile(g, 10):.2f}  p90 {np.percentile(g, 90):.2f}  max {g.max():.1f}")
    big = np.argsort(-d)[:10]
    print("  longest launches (index: us):", ", ".join(f"{i}: {d[i]:.1f}" for i in sorted(big)))
    bigg = np.argsort(-g)[:10]
    print("  longest gaps (after index: us):", ", ".join(f"{i}: {g[i]:.1f}" for i in sorted(bigg)))


main()
