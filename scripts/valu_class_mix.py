#!/usr/bin/env python3
"""VALU issue-class mix of every kernel in liborbx_hip.so, derived from the code object itself.

  python scripts/valu_class_mix.py [--lib orb-slam3-rust_amd/liborbx_hip.so] [--out profiles/valu_class_mix.json] [--show KERNEL]

What it does, with no GPU:
  1. unbundles the gfx950 code objects of the library (llvm-objdump --offloading on a copy in a temp dir) and disassembles them
     (llvm-objdump -d --symbolize-operands);
  2. classifies every VALU instruction by the issue class MEASURED on this chip: profiles/r02_valu_issue_probe.txt lists, per opcode,
     the chip-wide rate at 4 waves per SIMD — above 800 G wave-instr/s = the 2-cycle class (v_add_u32, v_and/or/xor_b32, v_lshrrev_b32,
     v_mov_b32, v_bitop3_b32, f32 add/mul/fma, v_min_u16), 400..800 = the 4-cycle class (packed i16 / f16, v_perm, v_alignbyte, dot4 /
     dot2, 24-bit multiplies, v_cmp, v_cndmask, min / max, DPP and SDWA forms, shifts left, v_bfe, ...), below 400 = 8 cycles
     (v_max3_i16, v_mad_u64_u32 at one wave).  An opcode the probe did not measure takes the class of its family (rules below) and is
     counted in `unmeasured` so that the share can be bounded;
  3. finds the loops of a kernel (a branch to an earlier label = back edge; its extent = label .. branch; nesting by containment)
     and weights every instruction by the product of the trip counts of the loops around it.  Trip counts are STATED, per kernel, in
     profiles/valu_loop_weights.json with the reason for each (loop order = address order of the loop heads; the script refuses a
     kernel whose loop count differs from the stated list, so a recompile that changes the loop structure is noticed);
  4. writes profiles/valu_class_mix.json: per kernel the static and the weighted counts per class, share_2cycle (what bench.py reads
     for roofline.valu_issue), the weighted VALU instructions per wave (to be compared with SQ_INSTS_VALU / SQ_WAVES of the counter
     pass: profiles/pmc_traffic.json — `model_vs_pmc` is filled in when that file has the kernel).
"""
import argparse
import collections
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def probe_classes(path):
    """opcode -> (class, G/s at 4 waves per SIMD) from the probe table."""
    cls = {}
    for line in open(path):
        if not line.startswith("v_") or "|" not in line:
            continue
        cols = [c.strip() for c in line.split("|")]
        name = cols[0]
        if " " in name and not name.endswith("op_sel"):     # "v_mov_b32 dpp wave_shr:1", "(sgpr-pair mask)": keyed separately below
            base, rest = name.split(" ", 1)
            key = base + ("|dpp" if "dpp" in rest else "|sdwa" if "sdwa" in base else "")
        else:
            key = name.split(" ")[0]
        try:
            gps = float(cols[3].split()[2])                  # 4 waves/SIMD: cyc/w simd G/s res
        except (IndexError, ValueError):
            continue
        c = 2 if gps > 800.0 else (4 if gps > 400.0 else 8)
        if key not in cls or name == key:
            cls[key] = (c, gps)
    return cls


# family rules for opcodes the probe did not measure (each use is counted in `unmeasured`)
def family_class(op):
    if op.startswith(("v_pk_", "v_cmp", "v_cmpx", "v_cndmask", "v_min", "v_max", "v_med3", "v_perm", "v_align", "v_dot", "v_bfe", "v_bfi",
                      "v_cvt", "v_lshl", "v_ashr", "v_mul", "v_mad", "v_sad", "v_msad", "v_mbcnt", "v_bcnt", "v_ffb", "v_lshr_b64",
                      "v_lshlrev_b64", "v_lshrrev_b64", "v_ashrrev", "v_add3", "v_and_or", "v_or3", "v_xad", "v_add_lshl", "v_lshl_add",
                      "v_readlane", "v_readfirstlane", "v_writelane", "v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos",
                      "v_frexp", "v_ldexp", "v_fract", "v_floor", "v_ceil", "v_trunc", "v_rndne", "v_div", "v_fma_f64", "v_add_f64",
                      "v_mul_f64", "v_fma_f16", "v_add_f16", "v_mul_f16", "v_accvgpr")):
        return 4
    if op.startswith(("v_add_co", "v_addc", "v_sub_co", "v_subb", "v_subrev", "v_sub_", "v_add_", "v_not", "v_xnor", "v_and", "v_or_", "v_xor",
                      "v_fma_f32", "v_fmac_f32", "v_fmaak", "v_fmamk", "v_mac_f32", "v_nop", "v_lshrrev_b32", "v_lshrrev_b16")):
        return 2
    return 4


def classify(op, operands, probe):
    """(class, measured?) of one VALU instruction.  DPP / SDWA forms issue in the 4-cycle class whatever the base opcode (probe rows
    'v_mov_b32 dpp', 'v_add_u32_sdwa')."""
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "mfma", True
    base = re.sub(r"_(e32|e64|dpp|sdwa|e64_dpp)$", "", op)
    if op.endswith(("_dpp", "_sdwa")) or " dpp" in operands or "row_" in operands or "quad_perm" in operands or "wave_sh" in operands or "row_bcast" in operands:
        return 4, True
    if base in probe:
        return probe[base][0], True
    return family_class(base), False


def unbundle(lib, tmp):
    cp = os.path.join(tmp, "lib.so")
    shutil.copy(lib, cp)
    subprocess.run([OBJDUMP, "--offloading", cp], cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
    return sorted(os.path.join(tmp, f) for f in os.listdir(tmp) if "gfx950" in f)


def disassemble(co):
    return subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", "--symbolize-operands", co], stdout=subprocess.PIPE, text=True, check=True).stdout


def demangle_short(sym):
    m = re.search(r"_GLOBAL__N_1(\d+)", sym)
    if m:
        n = int(m.group(1)); rest = sym[m.end():]
        name = rest[:n]; tail = rest[n:]
        t = re.match(r"I(L[bi]\d+E)+E", tail)
        if t:
            args = re.findall(r"L([bi])(\d+)E", t.group(0))
            name += "<" + ",".join(("true" if v == "1" else "false") if k == "b" else v for k, v in args) + ">"
        return name
    return sym


def parse_kernels(text):
    """kernel -> list of (kind, payload): ('label', name) | ('ins', (op, operands))."""
    kernels = collections.OrderedDict()
    cur = None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m and re.fullmatch(r"L\d+", m.group(1)):
            if cur is not None:
                cur.append(("label", m.group(1)))
            continue
        if m:
            cur = kernels.setdefault(demangle_short(m.group(1)), [])
            continue
        if cur is None or not line.startswith("\t"):
            continue
        body = line.split("//")[0].strip()
        if not body:
            continue
        parts = body.split(None, 1)
        cur.append(("ins", (parts[0], parts[1] if len(parts) > 1 else "")))
    return kernels


def analyse(items, probe, weights=None):
    pos = {}
    ins = []
    for kind, p in items:
        if kind == "label":
            pos[p] = len(ins)
        else:
            ins.append(p)
    # loops: backward branches
    loops = []
    for i, (op, opnd) in enumerate(ins):
        if op.startswith(("s_cbranch", "s_branch")):
            tgt = opnd.strip().split()[0] if opnd.strip() else ""
            if tgt in pos and pos[tgt] <= i:
                loops.append([pos[tgt], i, tgt])
    # merge back edges that share a head (one loop, several latches): extent = head .. last latch
    byhead = {}
    for a, b, t in loops:
        if a in byhead:
            byhead[a][1] = max(byhead[a][1], b)
        else:
            byhead[a] = [a, b, t]
    loops = sorted(byhead.values())
    w_loop = [1.0] * len(loops)
    stated = None
    if weights is not None:
        stated = weights.get("loops")
        if stated is not None and len(stated) != len(loops):
            raise SystemExit("loop structure changed: %d loops found, %d stated" % (len(loops), len(stated)))
        if stated is not None:
            w_loop = [float(x["trips"]) for x in stated]
    wt = [1.0] * len(ins)
    for (a, b, _), w in zip(loops, w_loop):
        for i in range(a, b + 1):
            wt[i] *= w
    # stated weights of straight-line regions (conditional paths that are rarely or partly taken): [first label, next label) ranges
    # ... {"skipped_to": label, "weight": w}: the instructions a forward conditional branch to `label` jumps over are executed by the
    # fraction w of the waves (s_cbranch_execz around a block only part of the waves enter)
    for reg in (weights or {}).get("regions", []):
        if reg["skipped_to"] not in pos:
            raise SystemExit("region label %s not in the kernel: the code changed, restate the weights" % reg["skipped_to"])
        b = pos[reg["skipped_to"]]
        srcs = [i for i, (op, opnd) in enumerate(ins) if op.startswith("s_cbranch") and opnd.strip().split()[0] == reg["skipped_to"] and i < b]
        if not srcs:
            raise SystemExit("no forward branch to %s" % reg["skipped_to"])
        for i in range(max(srcs) + 1, b):
            wt[i] *= float(reg["weight"])
    # stated weights of spans between two anchors (conditional rounds of an unrolled sequence that later `break`s skip, nested skips to one label):
    # {"from": anchor, "to": anchor, "weight": w}; an anchor is {"label": L} (the label's position) or {"branch_to": L, "nth": k} (the k-th
    # branch to L in address order, forward or backward); the instructions strictly after `from` and before `to` are executed by the
    # fraction w of the waves that reach `from` (spans multiply where they overlap)
    def anchor(a):
        if "label" in a:
            if a["label"] not in pos:
                raise SystemExit("span label %s not in the kernel: the code changed, restate the weights" % a["label"])
            return pos[a["label"]]
        br = [i for i, (op, opnd) in enumerate(ins) if op.startswith(("s_cbranch", "s_branch")) and opnd.strip().split()[0] == a["branch_to"]]
        if a["nth"] >= len(br):
            raise SystemExit("span anchor: branch %d to %s not in the kernel: the code changed, restate the weights" % (a["nth"], a["branch_to"]))
        return br[a["nth"]]
    for sp in (weights or {}).get("spans", []):
        a, b = anchor(sp["from"]), anchor(sp["to"])
        if not a < b:
            raise SystemExit("span %s: anchors out of order" % sp)
        if "instructions" in sp and sp["instructions"] != b - a - 1:
            raise SystemExit("span %s holds %d instructions, %d stated: the code changed (labels are renumbered by any edit), restate the weights" % (sp, b - a - 1, sp["instructions"]))
        for i in range(a + 1, b):
            wt[i] *= float(sp["weight"])
    stat = collections.Counter(); dyn = collections.Counter(); unm_s = 0; unm_d = 0.0
    ops_dyn = collections.Counter()
    salu = 0.0; lds = 0.0; vmem = 0.0
    for (op, opnd), w in zip(ins, wt):
        if op.startswith("v_"):
            c, measured = classify(op, opnd, probe)
            stat[c] += 1; dyn[c] += w
            ops_dyn[(op, c)] += w
            if not measured:
                unm_s += 1; unm_d += w
        elif op.startswith("s_") and not op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_endpgm", "s_sleep")):
            salu += w
        elif op.startswith("ds_"):
            lds += w
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            vmem += w
    def share(c):
        tot = c[2] + c[4] + c[8]
        return (c[2] / tot) if tot else 0.0
    loop_info = []
    for k, ((a, b, t), w) in enumerate(zip(loops, w_loop)):
        lc = collections.Counter()
        for i in range(a, b + 1):
            if ins[i][0].startswith("v_"):
                lc[classify(ins[i][0], ins[i][1], probe)[0]] += 1
        loop_info.append(dict(index=k, head=t, instructions=b - a + 1, valu_2cycle=lc[2], valu_4cycle=lc[4], valu_8cycle=lc[8], mfma=lc["mfma"],
                              trips=w, why=(stated[k].get("why") if stated else None)))
    return dict(instructions=len(ins),
                valu_static={"2cycle": stat[2], "4cycle": stat[4], "8cycle": stat[8], "mfma": stat["mfma"], "unmeasured_opcodes": unm_s},
                share_2cycle_static=round(share(stat), 4),
                valu_weighted_per_wave={"2cycle": round(dyn[2], 1), "4cycle": round(dyn[4], 1), "8cycle": round(dyn[8], 1), "mfma": round(dyn["mfma"], 1),
                                        "unmeasured_opcodes": round(unm_d, 1), "total": round(dyn[2] + dyn[4] + dyn[8] + dyn["mfma"], 1)},
                share_2cycle=round(share(dyn), 4),
                share_2cycle_bounds=[round((dyn[2] - sum(w for (o, c), w in ops_dyn.items() if c == 2 and not classify(o, "", probe)[1])) / max(dyn[2] + dyn[4] + dyn[8], 1e-9), 4),
                                     round((dyn[2] + sum(w for (o, c), w in ops_dyn.items() if c == 4 and not classify(o, "", probe)[1])) / max(dyn[2] + dyn[4] + dyn[8], 1e-9), 4)],
                salu_weighted_per_wave=round(salu, 1), lds_weighted_per_wave=round(lds, 1), vmem_weighted_per_wave=round(vmem, 1),
                weights="stated (profiles/valu_loop_weights.json)" if stated is not None else "static (every loop body once)",
                loops=loop_info,
                top_opcodes=[dict(op=o, cls=c, weighted=round(w, 1)) for (o, c), w in ops_dyn.most_common(12)])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=os.path.join(ROOT, "orb-slam3-rust_amd", "liborbx_hip.so"))
    ap.add_argument("--probe", default=os.path.join(ROOT, "profiles", "r02_valu_issue_probe.txt") + "," + os.path.join(ROOT, "profiles", "r03_valu_issue_probe2.txt"),
                    help="comma-separated probe tables (scripts/valu_issue_probe.hip, scripts/valu_issue_probe2.hip); a missing file is skipped")
    ap.add_argument("--weights", default=os.path.join(ROOT, "profiles", "valu_loop_weights.json"))
    ap.add_argument("--pmc", default=os.path.join(ROOT, "profiles", "pmc_traffic.json"))
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "valu_class_mix.json"))
    ap.add_argument("--show", default=None, help="print the loops of one kernel (to write its weights)")
    ap.add_argument("--skeleton", default=None, help="print the labels and branches of one kernel with their instruction indices (to write spans)")
    ap.add_argument("--kernels", default="fast_kernel<true>,blur_kernel,resize_kernel<6>,describe_kernel,describe_fused_kernel,describe_tile_kernel,harris_select_kernel,rank_select_kernel,"
                                         "stereo_match_kernel,stereo_match_lds_kernel<true>,stereo_bucket_kernel,stereo_compact_kernel")
    args = ap.parse_args()
    probe = {}
    probes_used = []
    for pth in args.probe.split(","):
        if os.path.exists(pth):
            for k, v in probe_classes(pth).items():
                probe.setdefault(k, v)
            probes_used.append(os.path.relpath(pth, ROOT))
    weights = json.load(open(args.weights)) if os.path.exists(args.weights) else {}
    pmc = json.load(open(args.pmc)).get("kernels", {}) if os.path.exists(args.pmc) else {}
    tmp = tempfile.mkdtemp(prefix="orbx_co_")
    try:
        kernels = collections.OrderedDict()
        for co in unbundle(args.lib, tmp):
            kernels.update(parse_kernels(disassemble(co)))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    want = [k for k in args.kernels.split(",") if k]
    out = dict(source="scripts/valu_class_mix.py: llvm-objdump -d of the gfx950 code objects in %s; opcode classes from %s (rate at 4 waves per SIMD: "
                      "> 800 G/s = 2-cycle, 400..800 = 4-cycle, < 400 = 8-cycle); loop trip counts from %s"
                      % (os.path.relpath(args.lib, ROOT), " + ".join(probes_used), os.path.relpath(args.weights, ROOT)),
               probe_opcodes=len(probe), kernels={})
    for name in want:
        if name not in kernels:
            print("kernel %s not in the code objects (have: %s...)" % (name, ", ".join(list(kernels)[:6])), file=sys.stderr)
            continue
        res = analyse(kernels[name], probe, weights.get(name))
        short = name.split("<")[0]
        kd = pmc.get(short, {})
        if kd.get("valu_wave_instr_per_launch") and kd.get("waves_per_launch"):
            meas = kd["valu_wave_instr_per_launch"] / kd["waves_per_launch"]
            res["model_vs_pmc"] = dict(pmc_valu_per_wave=round(meas, 1), model_valu_per_wave=res["valu_weighted_per_wave"]["total"],
                                       ratio=round(res["valu_weighted_per_wave"]["total"] / meas, 3))
        out["kernels"][short] = res
        if args.skeleton and args.skeleton in name:
            n = 0
            for kind, pz in kernels[name]:
                if kind == "label":
                    print("  %5d %s:" % (n, pz))
                else:
                    if pz[0].startswith(("s_cbranch", "s_branch", "s_barrier", "s_endpgm")):
                        print("  %5d     %s %s" % (n, pz[0], pz[1]))
                    n += 1
        if args.show and args.show in name:
            print(name, "static share", res["share_2cycle_static"], "weighted", res["share_2cycle"])
            for l in res["loops"]:
                print("  loop %d head %s: %d instr, valu 2c %d / 4c %d, trips %s" % (l["index"], l["head"], l["instructions"], l["valu_2cycle"], l["valu_4cycle"], l["trips"]))
    json.dump(out, open(args.out, "w"), indent=1)
    for k, v in out["kernels"].items():
        print("%-24s share_2cycle %.3f (static %.3f, bounds %s)  VALU/wave %.0f  %s" % (k, v["share_2cycle"], v["share_2cycle_static"], v["share_2cycle_bounds"],
                                                                                        v["valu_weighted_per_wave"]["total"], v.get("model_vs_pmc", "")))


if __name__ == "__main__":
    main()
