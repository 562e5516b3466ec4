#!/usr/bin/env python3
"""Per-launch durations of the 7 resize launches (levels 1..7) from a rocprofv3 --kernel-trace CSV: usage resize_launch_times.py <kernel_trace.csv>"""
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "resize_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per = collections.defaultdict(list)
gaps = collections.defaultdict(list)
for i, r in enumerate(rows):
    per[i % 7].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    if i % 7:
        gaps[i % 7].append((int(r["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"])) / 1e3)
for l in range(7):
    v = sorted(per[l]); g = sorted(gaps[l]) if gaps[l] else [0]
    print("level %d: median %.1f us over %d launches, gap before it %.1f us" % (l + 1, v[len(v) // 2], len(v), g[len(g) // 2]))
print("sum of medians %.1f us" % sum(sorted(per[l])[len(per[l]) // 2] for l in range(7)))
