#!/usr/bin/env python3
"""MFMA utilisation per kernel from a rocprofv3 --pmc pass with SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64
SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_WAVES: per-launch means (counters are sums over the 8 XCDs), and for kernels that issue MFMAs
  cycles per MFMA      = SQ_VALU_MFMA_BUSY_CYCLES / SQ_INSTS_MFMA
  MFMA-busy fraction   = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8)     (GRBM_GUI_ACTIVE is summed over 8 XCDs)
  executed f64 TFLOP/s = SQ_INSTS_MFMA * 2048 flop / (GRBM_GUI_ACTIVE / 8 / clock)         at the clock given (default 2.1 GHz under the profiler)
usage: pmc_mfma_summary.py <counter_collection.csv> [clock_GHz]"""
import collections, csv, re, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    name = re.split(r"[(<]", r["Kernel_Name"].replace("(anonymous namespace)::", ""))[0].strip()
    if name.startswith("void at::") or "rocclr" in name or name.startswith("at::"):
        continue
    acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
ghz = float(sys.argv[2]) if len(sys.argv) > 2 else 2.1
m = lambda k, c: sum(acc[k][c]) / max(len(acc[k][c]), 1) if c in acc[k] else 0.0
print("%-24s %9s %14s %12s %14s %10s %12s %12s %12s" % ("kernel", "launches", "GUI_ACTIVE/8", "SQ_WAVES", "INSTS_MFMA", "cyc/MFMA", "MFMA busy", "exec TFLOP/s", "of 78.6"))
for k in sorted(acc):
    n = len(next(iter(acc[k].values())))
    act = m(k, "GRBM_GUI_ACTIVE") / 8.0
    nm = m(k, "SQ_INSTS_MFMA")
    busy = m(k, "SQ_VALU_MFMA_BUSY_CYCLES")
    line = "%-24s %9d %14.4g %12.4g %14.4g" % (k[:24], n, act, m(k, "SQ_WAVES"), nm)
    if nm > 0 and act > 0:
        tf = nm * 2048.0 / (act / (ghz * 1e9)) / 1e12
        line += " %10.1f %11.1f%% %12.2f %11.1f%%" % (busy / nm, 100.0 * busy / (1024.0 * act), tf, 100.0 * tf / 78.6)
    print(line)
