mkdir -p gpurun_out/r04w
timeout -k 10 120 python scripts/bf_debug.py 2>&1 | grep -A1 "backward" | grep -B1 "n=294\|n=192\|n=320\|n=319" | tee gpurun_out/r04w/bf_debug.txt
timeout -k 10 300 python -m pytest tests/test_ba_gpu.py -q -m gpu -x 2>&1 | tail -2
for v in new rows_simd0 new rows_simd0; do
  if [ $v = new ]; then unset ORBX_LIBRARY; else export ORBX_LIBRARY=$PWD/build_ab/$v.so; fi
  echo "== $v"
  timeout -k 10 120 python scripts/ba_profile.py 50 8000 visual-only 2>/dev/null | grep -E "wall|ba_solve"
  timeout -k 10 120 python scripts/ba_profile.py 33 2000 visual-only 2>/dev/null | grep -E "wall|ba_solve"
done 2>&1 | tee gpurun_out/r04w/ab.txt
