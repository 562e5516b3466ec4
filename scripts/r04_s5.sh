set -e
mkdir -p gpurun_out/r04e
python -m pytest tests/test_ba_gpu.py -x -q -m gpu -k "batch or pinned or c_abi or two_handles or abort or partition" > gpurun_out/r04e/ba_batch_tests.txt 2>&1 || { tail -30 gpurun_out/r04e/ba_batch_tests.txt; exit 1; }
tail -2 gpurun_out/r04e/ba_batch_tests.txt
python bench.py --no-files --no-extras --no-cpu-baseline --steps 20 > gpurun_out/r04e/bench.json 2> gpurun_out/r04e/bench.err; python -c "
import json; d=json.load(open('gpurun_out/r04e/bench.json')); b=d['local_ba']; print('value', d['value'], 'unprofiled', d['value_unprofiled']); print(d['roofline']['kernel_ms_per_step']); print(json.dumps({k:b[k] for k in ('lm_iters_per_s','ms_per_solve')})); bb=b['batched']; print({k:bb[k] for k in ('lm_iters_per_s','ms_per_call','device_ms_per_call','lm_iters_per_s_device_only','call_vs_device_only')}); print(bb['obs_32_bytes']); print(bb['adhoc_mirror_call']); print(bb['pageable_observations']); print(json.dumps(bb['c_abi'])); print(bb['two_batches_in_flight']); print(bb['kernel_ms_per_iteration'])"
