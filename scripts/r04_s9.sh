mkdir -p gpurun_out/r04i
set -e
timeout -k 10 300 python -m pytest tests/test_frozen_golden.py tests/test_extract_gpu.py -x -q -m gpu > gpurun_out/r04i/extract_default.txt 2>&1 || { tail -20 gpurun_out/r04i/extract_default.txt; exit 1; }
echo "extractor tests, default build (window flipped to signed bytes when staged):"; tail -1 gpurun_out/r04i/extract_default.txt
python bench.py --no-ba --no-files --no-extras --no-cpu-baseline --steps 20 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default', d['value'], d['value_unprofiled'], d['roofline']['kernel_ms_per_step'])"
ORBX_BA_STAGE_OBS=1 python -m pytest tests/test_ba_gpu.py tests/test_inertial_ba.py tests/test_global_ba.py -x -q -m gpu > gpurun_out/r04i/ba_staged.txt 2>&1 || { tail -20 gpurun_out/r04i/ba_staged.txt; exit 1; }
echo "BA tests with every observation array staged (ORBX_BA_STAGE_OBS=1):"; tail -1 gpurun_out/r04i/ba_staged.txt
ORBX_DESC_UNFUSED=1 python -m pytest tests/test_frozen_golden.py tests/test_extract_gpu.py tests/test_properties_gpu.py -x -q -m gpu > gpurun_out/r04i/extract_unfused.txt 2>&1 || { tail -20 gpurun_out/r04i/extract_unfused.txt; exit 1; }
echo "extractor tests, two-kernel form (ORBX_DESC_UNFUSED=1):"; tail -1 gpurun_out/r04i/extract_unfused.txt
python bench.py --width 1920 --height 1080 --features 4000 --batch 64 --no-ba --no-files --no-extras --no-cpu-baseline > gpurun_out/r04i/bench_1080p_4000.json 2> gpurun_out/r04i/bench_1080p.err
python -c "
import json; d=json.loads(open('gpurun_out/r04i/bench_1080p_4000.json').read().strip().splitlines()[-1]); print('1080p', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'])"
ORBX_DIST_REHEARSE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 --batch 64 > gpurun_out/r04i/rehearse2.json 2> gpurun_out/r04i/rehearse2.err || { tail -20 gpurun_out/r04i/rehearse2.err; exit 1; }
python -c "
import json; d=json.loads(open('gpurun_out/r04i/rehearse2.json').read().strip().splitlines()[-1]); print('rehearsal n_gpus', d['n_gpus'], d['value'], d['config']['parallelism']); print(d['local_ba'])"
