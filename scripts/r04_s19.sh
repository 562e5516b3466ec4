set -e
mkdir -p gpurun_out/r04y
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04y/smoke.txt 2>&1 || { tail -20 gpurun_out/r04y/smoke.txt; exit 1; }
cat gpurun_out/r04y/smoke.txt | tail -4
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r04y/gpu_tests.txt 2>&1 || { tail -30 gpurun_out/r04y/gpu_tests.txt; exit 1; }
tail -2 gpurun_out/r04y/gpu_tests.txt
bash scripts/refresh_profiles.sh r04y > gpurun_out/r04y_refresh.log 2>&1 || { tail -20 gpurun_out/r04y_refresh.log; exit 1; }
tail -2 gpurun_out/r04y_refresh.log | cut -c1-400
bash scripts/ba_pmc.sh r04y > gpurun_out/r04y_bapmc.log 2>&1 || { tail -20 gpurun_out/r04y_bapmc.log; exit 1; }; tail -5 gpurun_out/r04y_bapmc.log | cut -c1-300
