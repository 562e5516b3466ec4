mkdir -p gpurun_out/r04n
python -m pytest tests/test_ba_gpu.py tests/test_inertial_ba.py tests/test_global_ba.py tests/test_fuzz_gpu.py tests/test_local_mapper_host.py -q -m gpu > gpurun_out/r04n/ba_tests.txt 2>&1
echo "fma BA tests rc=$?"; tail -30 gpurun_out/r04n/ba_tests.txt
for v in new old new old; do
  if [ $v = old ]; then export ORBX_LIBRARY=$PWD/build_ab/pre_fma.so; else unset ORBX_LIBRARY; fi
  echo "== $v"
  python scripts/ba_batch_profile.py 32 20 2000 kernels 2>/dev/null | grep -E "pinned|ba_[a-z_]*kernel|device ms"
  python scripts/ba_profile.py 20 2000 visual-only 2>/dev/null | grep -E "wall|ba_[a-z_]*kernel|sum of"
  python scripts/ba_profile.py 50 8000 visual-only 2>/dev/null | grep -E "wall|ba_[a-z_]*kernel|sum of"
done 2>&1 | tee gpurun_out/r04n/ab.txt
