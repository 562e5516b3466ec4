mkdir -p gpurun_out/r04ab
timeout -k 10 400 python -m pytest tests/test_ba_gpu.py tests/test_inertial_ba.py tests/test_global_ba.py tests/test_fuzz_gpu.py tests/test_local_mapper_host.py -q -m gpu > gpurun_out/r04ab/ba_tests.txt 2>&1
echo "BA tests rc=$?"; tail -5 gpurun_out/r04ab/ba_tests.txt
for i in 1 2; do
timeout -k 10 120 python scripts/ba_batch_profile.py 32 20 2000 kernels 2>/dev/null | grep -E "ba_gather|ba_decide|device ms"
timeout -k 10 120 python scripts/ba_profile.py 20 2000 visual-only 2>/dev/null | grep -E "wall|ba_gather|ba_decide|sum of"
timeout -k 10 120 python scripts/ba_profile.py 50 8000 visual-only 2>/dev/null | grep -E "wall|ba_gather|ba_decide|sum of"
done 2>&1 | tee gpurun_out/r04ab/decide.txt
