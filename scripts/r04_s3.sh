mkdir -p gpurun_out/r04c
# correctness of the variants that change the arithmetic path (byte stores by inline asm; f16 pattern table)
for v in v2 v4; do
  ORBX_LIBRARY=$PWD/build_ab/$v.so timeout -k 10 300 python -m pytest tests/test_frozen_golden.py -x -q -m gpu > gpurun_out/r04c/frozen_$v.txt 2>&1
  rc=$?; tail -2 gpurun_out/r04c/frozen_$v.txt
  if [ $rc -eq 124 ] || [ $rc -eq 137 ] || grep -q "Memory access fault\|core dumped\|Aborted" gpurun_out/r04c/frozen_$v.txt; then echo "died rc=$rc"; exit 1; fi
done
timeout -k 10 300 python -m pytest tests/test_frozen_golden.py tests/test_extract_gpu.py tests/test_properties_gpu.py -x -q -m gpu > gpurun_out/r04c/extract_default.txt 2>&1; tail -2 gpurun_out/r04c/extract_default.txt
set -e
bash scripts/ab_bench.sh r04c_ab build_ab/v0.so build_ab/v1.so build_ab/v2.so build_ab/v3.so build_ab/v4.so 2>&1 | tee gpurun_out/r04c/ab.txt
ORBX_DESC_UNFUSED=1 python bench.py --no-cpu-baseline --no-ba --no-files --no-extras --steps 20 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('unfused', d['value'], d['value_unprofiled'], d['roofline']['kernel_ms_per_step'])" | tee -a gpurun_out/r04c/ab.txt
