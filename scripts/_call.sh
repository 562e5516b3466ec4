set -e
O=gpurun_out/r03t
mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1 || { tail -30 $O/gpu_tests.txt; exit 1; }
tail -3 $O/gpu_tests.txt
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1 || { tail -20 $O/smoke.txt; exit 1; }
cat $O/smoke.txt
timeout -k 10 900 scripts/refresh_profiles.sh r03t
timeout -k 10 400 scripts/ba_pmc.sh r03t > $O/ba_pmc.out 2>&1 || echo "ba_pmc failed"
tail -14 $O/ba_pmc.out
timeout -k 10 300 python bench.py --width 1920 --height 1080 --features 4000 --batch 64 --no-ba --no-files --no-extras > $O/bench_1080p_4000.json 2> $O/bench_1080p.err
tail -c 300 $O/bench_1080p_4000.json
