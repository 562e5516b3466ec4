mkdir -p gpurun_out/r04x
timeout -k 10 400 python -m pytest tests/test_ba_gpu.py tests/test_inertial_ba.py tests/test_global_ba.py tests/test_fuzz_gpu.py tests/test_local_mapper_host.py -q -m gpu > gpurun_out/r04x/ba_tests.txt 2>&1
echo "BA tests rc=$?"; tail -5 gpurun_out/r04x/ba_tests.txt
timeout -k 10 120 python scripts/ba_profile.py 50 8000 visual-only 2>/dev/null | grep -E "wall|ba_[a-z_]*kernel|sum of"
timeout -k 10 120 python scripts/ba_profile.py 20 2000 2>/dev/null | grep -E "wall|sum of" | head -4
