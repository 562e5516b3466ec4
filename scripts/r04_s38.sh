mkdir -p gpurun_out/r04an
timeout -k 10 300 python -m pytest tests/test_extract_gpu.py tests/test_abi.py -q -m gpu -x 2>&1 | tail -3
timeout -k 10 500 python bench.py --no-cpu-baseline --no-ba --no-files --no-extras > gpurun_out/r04an/bench_short.json 2> gpurun_out/r04an/bench_short.err; tail -1 gpurun_out/r04an/bench_short.json | python -c "
import sys, json
d = json.loads(sys.stdin.read())
r = d['roofline']
print(d['value'], d['value_unprofiled'], d['ms_per_step'], r['kernel'], r['avg_launch_us'], r['valu_issue']['frac'] if r.get('valu_issue') else None)
print(r['kernel_ms_per_step']); print(r.get('kernel_ms_per_step_source'))
"
