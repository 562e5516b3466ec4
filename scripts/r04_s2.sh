mkdir -p gpurun_out/r04b
timeout -k 10 400 python -m pytest tests/test_frozen_golden.py tests/test_extract_gpu.py -x -q -m gpu > gpurun_out/r04b/extract_tests.txt 2>&1
rc=$?
tail -25 gpurun_out/r04b/extract_tests.txt
if [ $rc -eq 124 ] || [ $rc -eq 137 ] || grep -q "Memory access fault\|core dumped\|Aborted" gpurun_out/r04b/extract_tests.txt; then echo "extract tests died rc=$rc: stopping"; exit 1; fi
export EXTRACT_RC=$rc
if [ $rc -ne 0 ]; then export ORBX_DESC_UNFUSED=1; echo "fused describe FAILED its tests: the rest runs unfused"; fi
set -e
python -m pytest tests/test_ba_gpu.py tests/test_inertial_ba.py tests/test_global_ba.py tests/test_fuzz_gpu.py tests/test_local_mapper_host.py -x -q -m gpu > gpurun_out/r04b/ba_tests.txt 2>&1 || { tail -30 gpurun_out/r04b/ba_tests.txt; exit 1; }
tail -3 gpurun_out/r04b/ba_tests.txt
ORBX_BA_TIMING=1 ORBX_PROFILE_REPS=6 python scripts/ba_batch_profile.py 32 20 2000 kernels > gpurun_out/r04b/batch_profile.txt 2>&1
tail -45 gpurun_out/r04b/batch_profile.txt
python scripts/ba_profile.py > gpurun_out/r04b/ba_profile.txt 2>&1; tail -5 gpurun_out/r04b/ba_profile.txt
python bench.py --no-files --steps 20 > gpurun_out/r04b/bench.json 2> gpurun_out/r04b/bench.err; python -c "
import json; d=json.load(open('gpurun_out/r04b/bench.json')); b=d['local_ba']; print('value', d['value'], 'unprofiled', d['value_unprofiled']); print(d['roofline']['kernel_ms_per_step']); print(json.dumps({k:b[k] for k in ('lm_iters_per_s','ms_per_solve')})); print(json.dumps(b['batched'],indent=0)[:3000]); print(b['config5'].get('lm_iters_per_s'), b['inertial'].get('lm_iters_per_s')); print(json.dumps(b.get('cpu_baseline'))[:900])"
if [ "$EXTRACT_RC" = "0" ]; then ORBX_DESC_UNFUSED=1 python bench.py --no-files --no-ba --no-extras --no-cpu-baseline --steps 20 > gpurun_out/r04b/bench_unfused.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r04b/bench_unfused.json')); print('UNFUSED value', d['value'], d['roofline']['kernel_ms_per_step'])"; fi
