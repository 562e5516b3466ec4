mkdir -p gpurun_out/r04s
python -m pytest tests/test_ba_gpu.py tests/test_inertial_ba.py tests/test_global_ba.py tests/test_fuzz_gpu.py tests/test_local_mapper_host.py -q -m gpu > gpurun_out/r04s/ba_tests.txt 2>&1
echo "BA tests rc=$?"; tail -15 gpurun_out/r04s/ba_tests.txt
ORBX_LIBRARY=$PWD/build_ab/schst48.so python scripts/ba_schur_stamps.py 32 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04s/schur_stamps_48.txt
ORBX_LIBRARY=$PWD/build_ab/schst48.so python scripts/ba_schur_stamps.py 1 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04s/schur_stamps_48.txt
for v in new rows24 new rows24; do
  if [ $v = new ]; then unset ORBX_LIBRARY; else export ORBX_LIBRARY=$PWD/build_ab/$v.so; fi
  echo "== $v"
  python scripts/ba_batch_profile.py 32 20 2000 kernels 2>/dev/null | grep -E "pinned|ba_[a-z_]*kernel|device ms"
  python scripts/ba_profile.py 20 2000 visual-only 2>/dev/null | grep -E "wall|ba_kf_schur|sum of"
  python scripts/ba_profile.py 50 8000 visual-only 2>/dev/null | grep -E "wall|ba_kf_schur|sum of"
done 2>&1 | tee gpurun_out/r04s/ab.txt
