mkdir -p gpurun_out/r04g
echo fused; python scripts/pcie_probe.py 256 2>&1 | tail -2
echo unfused; ORBX_DESC_UNFUSED=1 python scripts/pcie_probe.py 256 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04g/trace -- python3 $GRAFT_REPO_ROOT/scripts/pcie_probe.py 64 > $GRAFT_REPO_ROOT/gpurun_out/r04g/trace_run.txt 2>&1
cd $GRAFT_REPO_ROOT; tail -2 gpurun_out/r04g/trace_run.txt; ls gpurun_out/r04g/trace/*/ | head; python - <<'PY'
import csv, glob
kt = glob.glob('gpurun_out/r04g/trace/*/*kernel_trace.csv'); mc = glob.glob('gpurun_out/r04g/trace/*/*memory_copy_trace.csv')
ev = []
for f in kt:
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-40:]))
for f in mc:
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r.get('Direction', '') + ' ' + r.get('Bytes', '')))
ev.sort()
t0 = ev[-400][0] if len(ev) > 400 else ev[0][0]
for s, e, n in ev[-400:-250]:
    print("%9.1f %8.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, n))
PY
