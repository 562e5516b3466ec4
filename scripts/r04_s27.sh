mkdir -p gpurun_out/r04ac
true

g++ -std=c++17 -O2 -I include tests/cpp/ba_batch_driver.cpp -o /tmp/ba_batch_driver -L orb-slam3-rust_amd -lorbx_hip -Wl,-rpath,$PWD/orb-slam3-rust_amd
python - <<'PY'
import sys; sys.path.insert(0, '.')
import orb_slam3_rust_amd as P
wins = [P.synth.keypoint_precision(P.synth.ba_window(200 + i, 20, 2000, P.BA_OBS)) for i in range(32)]
P.synth.write_ba_batch_file('/tmp/batch32.bin', wins, P.BA_OBS)
PY
for frac in 0.5 0.45 0.4 0.33 0.5 0.4; do
    echo -n "split_frac=$frac pinned32: "; ORBX_BA_SPLIT_FRAC=$frac timeout -k 10 60 /tmp/ba_batch_driver /tmp/batch32.bin /tmp/out.bin 16 pinned32 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_call_median'], d['ms_per_call_min'])"
done | tee gpurun_out/r04ac/split_sweep.txt
echo -n "no split (one stream): "; ORBX_BA_NO_SPLIT=1 timeout -k 10 60 /tmp/ba_batch_driver /tmp/batch32.bin /tmp/out.bin 16 pinned32 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_call_median'], d['ms_per_call_min'])" | tee -a gpurun_out/r04ac/split_sweep.txt
ORBX_BA_TIMING=1 timeout -k 10 60 /tmp/ba_batch_driver /tmp/batch32.bin /tmp/out.bin 4 pinned32 2> gpurun_out/r04ac/timing_c.txt > /dev/null; tail -6 gpurun_out/r04ac/timing_c.txt
