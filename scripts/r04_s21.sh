mkdir -p gpurun_out/r04u
timeout -k 10 400 python -m pytest tests/test_ba_gpu.py tests/test_inertial_ba.py tests/test_global_ba.py tests/test_fuzz_gpu.py tests/test_local_mapper_host.py -q -m gpu > gpurun_out/r04u/ba_tests.txt 2>&1
echo "BA tests rc=$?"; tail -15 gpurun_out/r04u/ba_tests.txt
for v in new back_sep new back_sep; do
  if [ $v = new ]; then unset ORBX_LIBRARY; else export ORBX_LIBRARY=$PWD/build_ab/$v.so; fi
  echo "== $v"
  timeout -k 10 120 python scripts/ba_profile.py 50 8000 visual-only 2>/dev/null | grep -E "wall|ba_[a-z_]*kernel|sum of"
  timeout -k 10 120 python scripts/ba_profile.py 33 2000 visual-only 2>/dev/null | grep -E "wall|ba_solve|sum of"
done 2>&1 | tee gpurun_out/r04u/ab.txt
