set -e
mkdir -p gpurun_out/r04d
python -m pytest tests/test_ba_gpu.py -x -q -m gpu -k "batch or pinned or c_abi or two_handles" > gpurun_out/r04d/ba_batch_tests.txt 2>&1 || { tail -30 gpurun_out/r04d/ba_batch_tests.txt; exit 1; }
tail -2 gpurun_out/r04d/ba_batch_tests.txt
g++ -std=c++17 -O2 -I include tests/cpp/ba_batch_driver.cpp -o /tmp/ba_batch_driver -L orb-slam3-rust_amd -lorbx_hip -Wl,-rpath,$PWD/orb-slam3-rust_amd
python - <<'PY'
import sys; sys.path.insert(0, '.')
import orb_slam3_rust_amd as P
wins = [P.synth.ba_window(200 + i, 20, 2000, P.BA_OBS) for i in range(32)]
P.synth.write_ba_batch_file('/tmp/batch32.bin', wins, P.BA_OBS)
PY
for frac in 0.5 0.4 0.33 0.25 0.2; do
  for rep in 1 2; do
    echo -n "split_frac=$frac pinned: "; ORBX_BA_SPLIT_FRAC=$frac /tmp/ba_batch_driver /tmp/batch32.bin /tmp/out.bin 12 pinned | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_call_median'], d['ms_per_call_min'])"
  done
done | tee gpurun_out/r04d/split_sweep.txt
echo -n "no split (one stream): "; ORBX_BA_NO_SPLIT=1 /tmp/ba_batch_driver /tmp/batch32.bin /tmp/out.bin 12 pinned | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_call_median'], d['ms_per_call_min'])" | tee -a gpurun_out/r04d/split_sweep.txt
echo -n "pageable 0.5: "; /tmp/ba_batch_driver /tmp/batch32.bin /tmp/out.bin 12 pageable | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_call_median'], d['ms_per_call_min'])" | tee -a gpurun_out/r04d/split_sweep.txt
ORBX_BA_TIMING=1 /tmp/ba_batch_driver /tmp/batch32.bin /tmp/out.bin 4 pinned 2> gpurun_out/r04d/timing_c.txt > /dev/null; tail -6 gpurun_out/r04d/timing_c.txt
python - <<'PY' | tee gpurun_out/r04d/mirror.txt
import sys, time; sys.path.insert(0, '.')
import orb_slam3_rust_amd as P
cam = P.CameraModel(**P.synth.EUROC_CAMERA); cfg = P.LocalBAConfigLM()
h = P.Handle(cam, 100)
wins = [P.synth.ba_window(200 + i, 20, 2000, P.BA_OBS) for i in range(32)]
b = h.prepare_ba_batch(wins); packed = P.Handle.pack_ba_windows(wins)
for name, f in (("BaBatch.solve", lambda: b.solve(cam, cfg)), ("adhoc pinned", lambda: h.ba_solve_visual_batch(cam, cfg, packed)), ("adhoc pageable", lambda: h.ba_solve_visual_batch(cam, cfg, wins))):
    f(); ts = []
    for _ in range(12):
        t0 = time.perf_counter(); r = f(); ts.append(time.perf_counter() - t0)
    ts.sort(); its = sum(x["iterations"] for x in r)
    print("%-16s median %.3f ms  min %.3f ms  -> %.0f LM it/s" % (name, ts[len(ts)//2]*1e3, ts[0]*1e3, its/ts[len(ts)//2]))
PY
