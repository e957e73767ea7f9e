#!/usr/bin/env python
"""Condense a rocprofv3 --kernel-trace --stats output directory into a small text
summary (top kernels by total time) that fits under profiles/.  Reads the
*_kernel_stats.csv if present, else aggregates *_kernel_trace.csv."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(d, out, top=45):
    top = int(top)
    stats = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    rows = []
    if stats:
        with open(stats[0]) as f:
            for r in csv.DictReader(f):
                rows.append((r["Name"], int(r["Calls"]), float(r["TotalDurationNs"]), float(r["AverageNs"]),
                             float(r["Percentage"])))
    else:
        tr = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
        agg = defaultdict(lambda: [0, 0.0])
        for t in tr:
            with open(t) as f:
                for r in csv.DictReader(f):
                    a = agg[r["Kernel_Name"]]
                    a[0] += 1
                    a[1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        tot = sum(v[1] for v in agg.values()) or 1.0
        rows = [(k, v[0], v[1], v[1] / v[0], 100 * v[1] / tot) for k, v in agg.items()]
    rows.sort(key=lambda r: -r[2])
    tot = sum(r[2] for r in rows)
    with open(out, "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats summary of {d}\n")
        f.write(f"# total kernel time {tot/1e6:.3f} ms over {sum(r[1] for r in rows)} launches, {len(rows)} distinct kernels\n")
        f.write(f"{'calls':>7} {'total_ms':>10} {'avg_us':>10} {'pct':>6}  name\n")
        for name, calls, total, avg, pct in rows[:top]:
            f.write(f"{calls:7d} {total/1e6:10.3f} {avg/1e3:10.2f} {100*total/tot:6.2f}  {name[:150]}\n")
    print(open(out).read())


if __name__ == "__main__":
    main(*sys.argv[1:4])
