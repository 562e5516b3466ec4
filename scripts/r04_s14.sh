mkdir -p gpurun_out/r04o
ORBX_LIBRARY=$PWD/build_ab/schst.so python scripts/ba_schur_stamps.py 32 2>&1 | tee gpurun_out/r04o/schur_stamps.txt
ORBX_LIBRARY=$PWD/build_ab/schst.so python scripts/ba_schur_stamps.py 1 2>&1 | tee -a gpurun_out/r04o/schur_stamps.txt
for v in new mb3 mb4 new mb3 mb4; do
  if [ $v = new ]; then unset ORBX_LIBRARY; else export ORBX_LIBRARY=$PWD/build_ab/$v.so; fi
  echo "== $v"
  python scripts/ba_batch_profile.py 32 20 2000 kernels 2>/dev/null | grep -E "pinned|ba_build|device ms"
  python scripts/ba_profile.py 20 2000 visual-only 2>/dev/null | grep -E "wall|ba_build|sum of"
done 2>&1 | tee gpurun_out/r04o/ab.txt
