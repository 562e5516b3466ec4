mkdir -p gpurun_out/r04al
env ORBX_BA_NO_SPLIT=1 timeout -k 10 400 python -m pytest tests/test_ba_gpu.py tests/test_inertial_ba.py tests/test_global_ba.py tests/test_fuzz_gpu.py tests/test_local_mapper_host.py -q -m gpu 2>&1 | tail -25 | tee gpurun_out/r04al/nosplit.txt
