mkdir -p gpurun_out/r04ak
for env in "ORBX_BA_FUSED=0" "ORBX_BA_STAGE_OBS=1" "ORBX_BA_NO_SPLIT=1" "ORBX_BA_BIG_STEPS=1"; do
  echo "== $env"
  env $env timeout -k 10 400 python -m pytest tests/test_ba_gpu.py tests/test_inertial_ba.py tests/test_global_ba.py tests/test_fuzz_gpu.py tests/test_local_mapper_host.py -q -m gpu -x 2>&1 | tail -2
done 2>&1 | tee gpurun_out/r04ak/alt_paths.txt
echo "== ORBX_DESC_UNFUSED=1"
ORBX_DESC_UNFUSED=1 timeout -k 10 400 python -m pytest tests/test_extract_gpu.py tests/test_properties_gpu.py tests/test_frozen_golden.py -q -m gpu -x 2>&1 | tail -2 | tee -a gpurun_out/r04ak/alt_paths.txt
