mkdir -p gpurun_out/r04k
python -m pytest tests/test_ba_gpu.py tests/test_inertial_ba.py tests/test_global_ba.py tests/test_fuzz_gpu.py tests/test_local_mapper_host.py -q -m gpu > gpurun_out/r04k/ba_tests_z.txt 2>&1
echo "Z-form BA tests rc=$?"; tail -15 gpurun_out/r04k/ba_tests_z.txt
for v in z yw z yw; do
  if [ $v = yw ]; then export ORBX_LIBRARY=$PWD/build_ab/schur_yw.so; else unset ORBX_LIBRARY; fi
  echo "== $v"
  python scripts/ba_batch_profile.py 32 20 2000 kernels 2>/dev/null | grep -E "pinned|ba_schur|ba_build|device ms"
  python scripts/ba_profile.py 20 2000 visual-only 2>/dev/null | grep -E "wall|kf_schur|sum of"
  python scripts/ba_profile.py 50 8000 visual-only 2>/dev/null | grep -E "wall|kf_schur|sum of"
done 2>&1 | tee gpurun_out/r04k/ab.txt
unset ORBX_LIBRARY
echo "== pinned vs pageable, one window"
ORBX_BA_TIMING=1 python scripts/ba_pinned_probe.py 50 8000 2>&1 | grep -E "pageable|pinned|orbx ba" | tail -24 | tee gpurun_out/r04k/pinned_probe.txt
python scripts/ba_pinned_probe.py 20 2000 2>&1 | grep -E "pageable|pinned" | tee -a gpurun_out/r04k/pinned_probe.txt
