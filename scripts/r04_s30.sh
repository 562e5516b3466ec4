mkdir -p gpurun_out/r04af
timeout -k 10 300 python -m pytest tests/test_ba_gpu.py -q -m gpu -x -k "config5 or multi_block or global_cholesky or large_window" 2>&1 | tail -2
for v in new base new base; do
  if [ $v = new ]; then unset ORBX_LIBRARY; else export ORBX_LIBRARY=$PWD/build_ab/$v.so; fi
  echo "== $v"
  timeout -k 10 120 python scripts/ba_profile.py 50 8000 visual-only 2>/dev/null | grep -E "wall|ba_kf_schur|sum of"
done 2>&1 | tee gpurun_out/r04af/ab.txt
