mkdir -p gpurun_out/r04ad
timeout -k 10 400 python -m pytest tests/test_ba_gpu.py tests/test_inertial_ba.py tests/test_global_ba.py tests/test_fuzz_gpu.py tests/test_local_mapper_host.py -q -m gpu -x > gpurun_out/r04ad/ba_tests.txt 2>&1
echo "BA tests rc=$?"; tail -4 gpurun_out/r04ad/ba_tests.txt
for v in 1 0 1 0 1 0; do
  echo "== fused $v"
  ORBX_BA_FUSED=$v timeout -k 10 120 python scripts/ba_profile.py 20 2000 visual-only 2>/dev/null | grep -E "wall|sum of"
  ORBX_BA_FUSED=$v timeout -k 10 120 python scripts/ba_profile.py 50 8000 visual-only 2>/dev/null | grep -E "wall|sum of"
  ORBX_BA_FUSED=$v timeout -k 10 120 python scripts/ba_profile.py 8 400 visual-only 2>/dev/null | grep -E "wall|sum of"
done 2>&1 | tee gpurun_out/r04ad/ab2.txt
