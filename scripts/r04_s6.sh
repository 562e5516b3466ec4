set -e
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r04/gpu_tests.txt 2>&1 || { tail -30 gpurun_out/r04/gpu_tests.txt; exit 1; }
tail -3 gpurun_out/r04/gpu_tests.txt
bash scripts/refresh_profiles.sh r04 > gpurun_out/r04_refresh.log 2>&1 || { tail -20 gpurun_out/r04_refresh.log; exit 1; }
tail -3 gpurun_out/r04_refresh.log
bash scripts/ba_pmc.sh r04 > gpurun_out/r04_bapmc.log 2>&1 || { tail -20 gpurun_out/r04_bapmc.log; exit 1; }
tail -12 gpurun_out/r04_bapmc.log
