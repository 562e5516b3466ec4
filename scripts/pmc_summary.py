#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes per kernel.

usage: pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> [batch n_features out.json [sq_counter_collection.csv]]
The optional SQ pass (SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES) adds wave-instruction counts per launch, from
which bench.py prices the VALU issue rate of the dominant kernel (one wave64 VALU instruction = 4 cycles of a SIMD).
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

import json
f = load(sys.argv[1], "FETCH_SIZE"); w = load(sys.argv[2], "WRITE_SIZE")
out = dict(batch=int(sys.argv[3]) if len(sys.argv) > 3 else None, n_features=int(sys.argv[4]) if len(sys.argv) > 4 else None,
           note="per launch; hbm_bytes = 2*FETCH_SIZE KiB (gfx950 half-count correction, calibrated for wide "
                "coalesced streams only) + WRITE_SIZE KiB; fetch_raw_bytes is the uncorrected counter", kernels={})
for k in sorted(f):
    fm = sum(f[k]) / len(f[k]); wm = sum(w.get(k, [0])) / max(len(w.get(k, [0])), 1)
    out["kernels"][k] = dict(fetch_raw_bytes=int(fm * 1024), write_bytes=int(wm * 1024), hbm_bytes_per_launch=int((2 * fm + wm) * 1024))
if len(sys.argv) > 6:
    for tag, key in (("SQ_INSTS_VALU", "valu_wave_instr"), ("SQ_INSTS_SALU", "salu_wave_instr"), ("SQ_INSTS_LDS", "lds_wave_instr"),
                     ("SQ_WAVES", "waves")):
        c = load(sys.argv[6], tag)
        for k in c:
            if k in out["kernels"]:
                out["kernels"][k][key + "_per_launch"] = int(sum(c[k]) / len(c[k]))
if len(sys.argv) > 5:
    json.dump(out, open(sys.argv[5], "w"), indent=1)
print("%-28s %8s %14s %14s %14s" % ("kernel", "launches", "FETCH KiB", "FETCHx2 MB", "WRITE MB"))
for k in sorted(f):
    fm = sum(f[k]) / len(f[k]); wm = sum(w.get(k, [0])) / max(len(w.get(k, [0])), 1)
    print("%-28s %8d %14.1f %14.2f %14.2f" % (k, len(f[k]), fm, 2 * fm * 1024 / 1e6, wm * 1024 / 1e6))
