"""What a rocprofv3 kernel trace says about a BATCHED run (several streams at once): per kernel name the launches, the sum of their
durations, and the wall time during which at least one launch of that name was running (the union of its intervals); the same for
all kernels together -- a name whose summed duration is many times its union shares the chip with itself on several streams.
usage: trace_overlap.py <kernel_trace.csv>"""
import csv
import re
import sys
from collections import defaultdict


def union(iv):
    iv.sort()
    tot, cs, ce = 0, None, None
    for s, e in iv:
        if cs is None or s > ce:
            if cs is not None:
                tot += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    if cs is not None:
        tot += ce - cs
    return tot


def short(name):
    name = re.sub(r"^void\s+", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("cyto::", "")


def main():
    by = defaultdict(list)
    allv = []
    for r in csv.DictReader(open(sys.argv[1])):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        by[short(r["Kernel_Name"])].append((s, e))
        allv.append((s, e))
    span = max(e for _, e in allv) - min(s for s, _ in allv)
    print(f"all kernels: {len(allv)} launches, summed {sum(e - s for s, e in allv) / 1e6:.1f} ms, busy (union) {union(list(allv)) / 1e6:.1f} ms, first to last {span / 1e6:.1f} ms")
    rows = sorted(by.items(), key=lambda kv: -sum(e - s for s, e in kv[1]))
    for k, iv in rows[:18]:
        d = [e - s for s, e in iv]
        d.sort()
        print(f"{k[:58]:58s} n={len(iv):6d} sum {sum(d) / 1e6:9.2f} ms  union {union(list(iv)) / 1e6:8.2f} ms  median {d[len(d) // 2] / 1e3:8.1f} us  p90 {d[int(len(d) * 0.9)] / 1e3:8.1f}  max {d[-1] / 1e3:9.1f}")


main()
