set -e -o pipefail
O=gpurun_out/r03p
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_ba_gpu.py tests/test_inertial_ba.py tests/test_global_ba.py tests/test_fuzz_gpu.py -m gpu -q -x > $O/ba_tests.txt 2>&1 || { tail -60 $O/ba_tests.txt; exit 1; }
tail -2 $O/ba_tests.txt
timeout -k 10 120 python scripts/ba_profile.py 20 2000 visual-only 2>&1 | grep "wall\|solve"
ORBX_LIBRARY=$PWD/build_ab/bast.so timeout -k 10 120 python scripts/ba_solve_stamps.py 2>&1 | tail -12
