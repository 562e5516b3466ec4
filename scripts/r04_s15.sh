mkdir -p gpurun_out/r04p
ORBX_LIBRARY=$PWD/build_ab/schst_p1.so python scripts/ba_schur_stamps.py 32 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04p/schur_stamps_p1.txt
for v in new prio1 prio3 new prio1 prio3; do
  if [ $v = new ]; then unset ORBX_LIBRARY; else export ORBX_LIBRARY=$PWD/build_ab/$v.so; fi
  echo "== $v"
  python scripts/ba_batch_profile.py 32 20 2000 kernels 2>/dev/null | grep -E "pinned|ba_schur|ba_build|device ms"
  python scripts/ba_profile.py 20 2000 visual-only 2>/dev/null | grep -E "wall|ba_kf_schur|sum of"
done 2>&1 | tee gpurun_out/r04p/ab.txt
