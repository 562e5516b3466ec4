#!/usr/bin/env python3
"""Per-kernel means of every counter in a rocprofv3 counter_collection.csv.  usage: pmc_sq_summary.py <csv> [<csv> ...]"""
import collections, csv, re, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        name = re.split(r"[(<]", r["Kernel_Name"].replace("(anonymous namespace)::", ""))[0].strip()
        if name.startswith("void at::") or "rocclr" in name or name.startswith("at::"):
            continue
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
ctrs = sorted({c for k in acc for c in acc[k]})
print("%-26s" % "kernel" + "".join("%18s" % c[:17] for c in ctrs))
for k in sorted(acc):
    print("%-26s" % k[:25] + "".join("%18.4g" % (sum(acc[k][c]) / max(len(acc[k][c]), 1)) for c in ctrs))
