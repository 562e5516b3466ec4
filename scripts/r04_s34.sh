mkdir -p gpurun_out/r04aj
for fl in 100000 6144 4500 3500; do
for sl in 192 258 384 768; do
  echo "== floor $fl slots $sl"
  ORBX_BA_GEN_FLOOR=$fl ORBX_BA_GEN_SLOTS=$sl timeout -k 10 120 python scripts/ba_profile.py 50 8000 visual-only 2>/dev/null | grep -E "ba_kf_schur"
done; done 2>&1 | tee gpurun_out/r04aj/sweep2.txt
