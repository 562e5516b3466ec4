#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes per kernel.

usage: pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> [images_per_launch]
FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE reports half the bytes of a wide coalesced
stream (MI355X_MICROARCH.md §HBM), so the corrected read figure is 2x; other access widths are
uncalibrated — both the raw and the doubled value are printed.
"""
import collections, csv, re, sys

def load(path, tag):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != tag:
            continue
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "")
        name = re.split(r"[(<]", name)[0].strip()
        if name.startswith("void at::") or "rocclr" in name or name.startswith("at::"):
            continue
        acc[name].append(float(r["Counter_Value"]))
    return acc

f = load(sys.argv[1], "FETCH_SIZE"); w = load(sys.argv[2], "WRITE_SIZE")
print("%-28s %8s %14s %14s %14s" % ("kernel", "launches", "FETCH KiB", "FETCHx2 MB", "WRITE MB"))
for k in sorted(f):
    fm = sum(f[k]) / len(f[k]); wm = sum(w.get(k, [0])) / max(len(w.get(k, [0])), 1)
    print("%-28s %8d %14.1f %14.2f %14.2f" % (k, len(f[k]), fm, 2 * fm * 1024 / 1e6, wm * 1024 / 1e6))
