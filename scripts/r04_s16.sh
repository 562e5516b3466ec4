mkdir -p gpurun_out/r04q
python -m pytest tests/test_ba_gpu.py -q -m gpu -x > gpurun_out/r04q/ba_tests.txt 2>&1
echo "BA tests rc=$?"; tail -5 gpurun_out/r04q/ba_tests.txt
ORBX_LIBRARY=$PWD/build_ab/schst_d2.so python scripts/ba_schur_stamps.py 32 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04q/schur_stamps_d2.txt
for v in new d1 new d1; do
  if [ $v = new ]; then unset ORBX_LIBRARY; else export ORBX_LIBRARY=$PWD/build_ab/$v.so; fi
  echo "== $v"
  python scripts/ba_batch_profile.py 32 20 2000 kernels 2>/dev/null | grep -E "pinned|ba_schur|device ms"
  python scripts/ba_profile.py 20 2000 visual-only 2>/dev/null | grep -E "wall|ba_kf_schur|sum of"
done 2>&1 | tee gpurun_out/r04q/ab.txt
