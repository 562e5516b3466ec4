mkdir -p gpurun_out/r04ai
timeout -k 10 500 python -m pytest tests/test_ba_gpu.py tests/test_global_ba.py tests/test_fuzz_gpu.py -q -m gpu -x > gpurun_out/r04ai/ba_tests.txt 2>&1
echo "BA tests rc=$?"; tail -12 gpurun_out/r04ai/ba_tests.txt
for v in 1 0 1 0; do
  echo "== gen_ws $v"
  ORBX_BA_GEN_WS=$v timeout -k 10 120 python scripts/ba_profile.py 50 8000 visual-only 2>/dev/null | grep -E "wall|ba_kf_schur|sum of"
  ORBX_BA_GEN_WS=$v timeout -k 10 120 python scripts/ba_profile.py 33 2000 visual-only 2>/dev/null | grep -E "wall|ba_kf_schur|sum of"
done 2>&1 | tee gpurun_out/r04ai/ab.txt
