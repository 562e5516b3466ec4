set -e
O=gpurun_out/r03w
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_ba_gpu.py tests/test_inertial_ba.py tests/test_global_ba.py tests/test_fuzz_gpu.py -m gpu -x -q > $O/ba_tests.txt 2>&1 || { tail -40 $O/ba_tests.txt; exit 1; }
tail -2 $O/ba_tests.txt
timeout -k 10 120 python scripts/ba_profile.py 50 8000 visual-only 2>&1 | grep "wall\|_kernel\|chi2"
timeout -k 10 120 python scripts/ba_profile.py 20 2000 2>&1 | grep "wall\|gather\|inertial"
