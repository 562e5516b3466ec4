mkdir -p gpurun_out/r04r
for v in new nomfma nofill nomfma_p0; do
  if [ $v = new ]; then unset ORBX_LIBRARY; else export ORBX_LIBRARY=$PWD/build_ab/$v.so; fi
  echo "== $v"
  python scripts/ba_batch_profile.py 32 20 2000 kernels 2>/dev/null | grep -E "ba_schur|device ms"
done 2>&1 | tee gpurun_out/r04r/ab.txt
