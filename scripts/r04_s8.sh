mkdir -p gpurun_out/r04h
for v in fused unfused; do
  if [ $v = unfused ]; then export ORBX_DESC_UNFUSED=1; fi
  python bench.py --no-ba --no-files --no-cpu-baseline --steps 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['pcie_inclusive'], d['batch_sweep']['256'])"
done
unset ORBX_DESC_UNFUSED
python bench.py --no-ba --no-files --no-cpu-baseline --steps 10 --batch 128 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('b128', d['value'], d['pcie_inclusive'])"
