#!/usr/bin/env python3
"""Launch-to-launch gaps of the single-window LM loop from a rocprofv3 --kernel-trace CSV of scripts/ba_rate.py: usage ba_gap_summary.py <kernel_trace.csv>"""
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "ba_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
busy = gap = 0
n = 0
per = collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    d = int(a["End_Timestamp"]) - int(a["Start_Timestamp"])
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    name = a["Kernel_Name"].split("(")[0].split("::")[-1].strip()
    per[name].append(d)
    if 0 <= g < 50000:          # inside a solve (between solves the host is in the way)
        busy += d; gap += g; n += 1
print("launches %d: kernel time %.1f us, gaps %.1f us (%.1f %% of the GPU timeline inside the solves), mean gap %.2f us" % (n, busy / 1e3, gap / 1e3, 100.0 * gap / (busy + gap), gap / n / 1e3))
for k, v in sorted(per.items()):
    v.sort()
    print("  %-28s n %5d  median %7.1f us" % (k, len(v), v[len(v) // 2] / 1e3))
