mkdir -p gpurun_out/r04m
python -m pytest tests/test_ba_gpu.py tests/test_inertial_ba.py tests/test_global_ba.py tests/test_fuzz_gpu.py tests/test_local_mapper_host.py -q -m gpu > gpurun_out/r04m/ba_tests.txt 2>&1
echo "rsqrt-M BA tests rc=$?"; tail -15 gpurun_out/r04m/ba_tests.txt
for v in rsq chol rsq chol; do
  if [ $v = chol ]; then export ORBX_LIBRARY=$PWD/build_ab/z_chol.so; else unset ORBX_LIBRARY; fi
  echo "== $v"
  python scripts/ba_batch_profile.py 32 20 2000 kernels 2>/dev/null | grep -E "pinned|ba_schur|ba_build|device ms"
  python scripts/ba_profile.py 20 2000 visual-only 2>/dev/null | grep -E "wall|build|sum of"
  python scripts/ba_profile.py 50 8000 visual-only 2>/dev/null | grep -E "wall|build|sum of"
done 2>&1 | tee gpurun_out/r04m/ab.txt
