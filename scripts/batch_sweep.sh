#!/bin/bash
# value / value_unprofiled of bench.py at several batch sizes on one box.   usage: scripts/batch_sweep.sh <tag> <batch> [<batch> ...]
set -e
TAG=$1; shift
O=gpurun_out/$TAG
mkdir -p $O
for b in "$@"; do
  python bench.py --no-cpu-baseline --no-ba --no-files --no-extras --batch $b --steps ${STEPS:-20} > $O/batch_$b.json 2> $O/batch_$b.err
  python - "$O/batch_$b.json" "$b" <<'PY' | tee -a $O/sweep.txt
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d["roofline"]["kernel_ms_per_step"]
print("batch %-5s value %9.1f unprofiled %9.1f ms/step %.4f " % (sys.argv[2], d["value"], d["value_unprofiled"], d["ms_per_step"]) + " ".join("%s=%.4f" % (a.replace("_kernel", ""), b) for a, b in k.items()))
PY
done
