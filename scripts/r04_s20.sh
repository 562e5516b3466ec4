mkdir -p gpurun_out/r04t
python -m pytest tests/test_extract_gpu.py tests/test_matcher_gpu.py tests/test_properties_gpu.py tests/test_frozen_golden.py tests/test_fuzz_gpu.py tests/test_keyframe_gpu.py -q -m gpu -x > gpurun_out/r04t/orb_tests.txt 2>&1
echo "orb tests rc=$?"; tail -5 gpurun_out/r04t/orb_tests.txt
bash scripts/ab_bench.sh r04t build_ab/new.so build_ab/harris_row32.so build_ab/sm_gather.so 2>&1 | tee gpurun_out/r04t/ab.txt
