set -e -o pipefail
bash scripts/refresh_profiles.sh r03s
bash scripts/ba_pmc.sh r03s > /dev/null
python bench.py --width 1920 --height 1080 --features 4000 --batch 64 --no-ba --no-files --no-extras --no-cpu-baseline > gpurun_out/r03s/bench_1080p_4000.json 2> gpurun_out/r03s/bench_1080p.err
tail -c 300 gpurun_out/r03s/bench_1080p_4000.json
