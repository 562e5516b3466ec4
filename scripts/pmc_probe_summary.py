#!/usr/bin/env python3
"""What SQ_ACTIVE_INST_VALU counts: the issue probe (one opcode per kernel instance) under
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE -- scripts/valu_issue_probe
usage: pmc_probe_summary.py <counter_collection.csv>"""
import collections, csv, sys
ACT = sys.argv[2] if len(sys.argv) > 2 else "SQ_ACTIVE_INST_VALU"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("%-70s %9s %12s %12s %8s %9s" % ("kernel", "launches", "INSTS_VALU", "ACTIVE_VALU", "act/inst", "act*4/cyc"))
for k in sorted(acc):
    c = acc[k]
    if "SQ_INSTS_VALU" not in c:
        continue
    # launches of one instance differ in occupancy (1, 2, 4, 8 waves per SIMD): print each
    n = len(c["SQ_INSTS_VALU"])
    for i in range(n):
        ins = c["SQ_INSTS_VALU"][i]; act = c[ACT][i]; gui = c["GRBM_GUI_ACTIVE"][i]
        print("%-70s %9d %12.4g %12.4g %8.3f %9.3f" % (k[:70], i, ins, act, act / max(ins, 1), act * 4 / max(gui / 8 * 1024, 1)))
