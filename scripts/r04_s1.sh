set -e
mkdir -p gpurun_out/r04a
python -m pytest tests/test_ba_gpu.py tests/test_inertial_ba.py tests/test_global_ba.py tests/test_fuzz_gpu.py tests/test_local_mapper_host.py -x -q -m gpu > gpurun_out/r04a/ba_tests.txt 2>&1 || { tail -30 gpurun_out/r04a/ba_tests.txt; exit 1; }
tail -3 gpurun_out/r04a/ba_tests.txt
ORBX_BA_TIMING=1 ORBX_PROFILE_REPS=6 python scripts/ba_batch_profile.py 32 20 2000 kernels > gpurun_out/r04a/batch_profile.txt 2>&1
cat gpurun_out/r04a/batch_profile.txt | tail -40
python scripts/ba_profile.py > gpurun_out/r04a/ba_profile.txt 2>&1; tail -5 gpurun_out/r04a/ba_profile.txt
ORBX_DESC_UNFUSED=1 python bench.py --no-extras --no-files --no-cpu-baseline --steps 10 > gpurun_out/r04a/bench_ba.json 2> gpurun_out/r04a/bench_ba.err; python -c "
import json; d=json.load(open('gpurun_out/r04a/bench_ba.json')); b=d['local_ba']; print(d['value']); print(json.dumps({k:b[k] for k in ('lm_iters_per_s','ms_per_solve')})); print(json.dumps(b['batched'],indent=0)[:3000]); print(b['config5'].get('lm_iters_per_s'), b['inertial'].get('lm_iters_per_s')); print(json.dumps(b.get('cpu_baseline'))[:600])"
