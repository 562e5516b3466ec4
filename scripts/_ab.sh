python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-ba --no-files 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['value_unprofiled'], d['ms_per_step'], {k: round(v,4) for k,v in d['roofline']['kernel_ms_per_step'].items()})"
