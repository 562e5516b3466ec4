mkdir -p gpurun_out/r04ah
timeout -k 10 500 python -m pytest tests/test_ba_gpu.py tests/test_inertial_ba.py tests/test_global_ba.py tests/test_fuzz_gpu.py tests/test_local_mapper_host.py -q -m gpu -x > gpurun_out/r04ah/ba_tests.txt 2>&1
echo "BA tests rc=$?"; tail -12 gpurun_out/r04ah/ba_tests.txt
for v in 1 0 1 0; do
  echo "== fused $v"
  ORBX_BA_FUSED=$v timeout -k 10 200 python scripts/ba_profile.py 20 2000 2>/dev/null | grep -iE "inertial|wall"
done 2>&1 | tee gpurun_out/r04ah/ab.txt
