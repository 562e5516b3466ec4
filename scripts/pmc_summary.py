#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes per kernel.

usage: pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> [batch n_features out.json [sq_counter_collection.csv [sq2_counter_collection.csv]]]
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
        if name.startswith("void "):          # template instances carry their return type
            name = name[5:]
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
if len(sys.argv) > 7:
    # second SQ pass.  SQ_ACTIVE_INST_VALU turned out to count issued VALU instructions, not cycles (calibrated with the issue probe:
    # profiles/r02_pmc_counter_calibration.txt), so "valu_instr_x4_over_cycles" = instructions x 4 / SIMD cycles is an instruction rate in
    # units of the 4-cycle opcode class — the busy fraction of a kernel made of 4-cycle opcodes only, and up to 2.0 for 2-cycle opcodes.
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs.  LDS pipe activity and its bank-conflict share beside it
    extra = {t: load(sys.argv[7], t) for t in ("SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT")}
    for k in out["kernels"]:
        if k in extra["SQ_ACTIVE_INST_VALU"] and k in extra["GRBM_GUI_ACTIVE"]:
            act = sum(extra["SQ_ACTIVE_INST_VALU"][k]) / len(extra["SQ_ACTIVE_INST_VALU"][k])
            gui = sum(extra["GRBM_GUI_ACTIVE"][k]) / len(extra["GRBM_GUI_ACTIVE"][k])
            d = out["kernels"][k]
            d["valu_active_quadcycles_per_launch"] = int(act)
            d["gui_active_cycles_per_launch"] = int(gui)
            d["valu_instr_x4_over_cycles"] = round(act * 4 / (gui / 8 * 1024), 4) if gui > 0 else None
            if k in extra["SQ_LDS_IDX_ACTIVE"]:
                la = sum(extra["SQ_LDS_IDX_ACTIVE"][k]) / len(extra["SQ_LDS_IDX_ACTIVE"][k])
                lc = sum(extra["SQ_LDS_BANK_CONFLICT"].get(k, [0])) / max(len(extra["SQ_LDS_BANK_CONFLICT"].get(k, [0])), 1)
                d["lds_busy_frac"] = round(la / (gui / 8 * 256), 4) if gui > 0 else None
                d["lds_bank_conflict_share"] = round(lc / la, 4) if la > 0 else None
import os
if os.environ.get("ORBX_PMC_UNIQUE_PAIRS"):
    out["unique_pairs"] = int(os.environ["ORBX_PMC_UNIQUE_PAIRS"])        # the input the counter passes ran on (refresh_profiles.sh)
    out["input"] = os.environ.get("ORBX_PMC_INPUT", "")
if len(sys.argv) > 5:
    json.dump(out, open(sys.argv[5], "w"), indent=1)
print("%-28s %8s %14s %14s %14s" % ("kernel", "launches", "FETCH KiB", "FETCHx2 MB", "WRITE MB"))
for k in sorted(f):
    fm = sum(f[k]) / len(f[k]); wm = sum(w.get(k, [0])) / max(len(w.get(k, [0])), 1)
    print("%-28s %8d %14.1f %14.2f %14.2f" % (k, len(f[k]), fm, 2 * fm * 1024 / 1e6, wm * 1024 / 1e6))
if len(sys.argv) > 7:
    print()
    print("%-28s %14s %14s %10s %10s %12s" % ("kernel", "VALU instr", "VALU active x4", "instr x4/cyc", "LDS busy", "LDS conflict"))
    for k in sorted(out["kernels"]):
        d = out["kernels"][k]
        if "valu_instr_x4_over_cycles" in d:
            print("%-28s %14.3e %14.3e %9.1f%% %9.1f%% %11.1f%%" % (k, d.get("valu_wave_instr_per_launch", 0), 4.0 * d["valu_active_quadcycles_per_launch"],
                  100 * (d["valu_instr_x4_over_cycles"] or 0), 100 * (d.get("lds_busy_frac") or 0), 100 * (d.get("lds_bank_conflict_share") or 0)))
