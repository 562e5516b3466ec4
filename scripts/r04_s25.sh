mkdir -p gpurun_out/r04aa
timeout -k 10 300 python -m pytest tests/test_ba_gpu.py -q -m gpu -x 2>&1 | tail -2
for d in 4 1 2 4; do
  echo "== gather_div $d"
  ORBX_BA_GATHER_DIV=$d timeout -k 10 120 python scripts/ba_batch_profile.py 32 20 2000 kernels 2>/dev/null | grep -E "ba_gather|device ms"
done 2>&1 | tee gpurun_out/r04aa/gather_div2.txt
timeout -k 10 120 python scripts/ba_profile.py 20 2000 visual-only 2>/dev/null | grep -E "wall|ba_gather|sum of"
