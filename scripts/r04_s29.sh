mkdir -p gpurun_out/r04ae
timeout -k 10 300 python -m pytest tests/test_ba_gpu.py -q -m gpu -x 2>&1 | tail -2
for v in new base new base; do
  if [ $v = new ]; then unset ORBX_LIBRARY; else export ORBX_LIBRARY=$PWD/build_ab/$v.so; fi
  echo "== $v"
  timeout -k 10 120 python scripts/ba_batch_profile.py 32 20 2000 kernels 2>/dev/null | grep -E "ba_schur|device ms"
done 2>&1 | tee gpurun_out/r04ae/ab.txt
