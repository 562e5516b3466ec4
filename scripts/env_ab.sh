#!/bin/bash
# Same-box A/B of environment settings of ONE build (boxes differ by a few per cent: compare only within one call).
# usage: scripts/env_ab.sh <tag> "<NAME=VALUE ...|->" ["<NAME=VALUE ...>" ...]    ("-" = no setting)  -> gpurun_out/<tag>/ab.txt, alternating twice
set -e
TAG=$1; shift
O=gpurun_out/$TAG
mkdir -p $O
i=0
for rep in 1 2; do
  for cfg in "$@"; do
    i=$((i+1))
    if [ "$cfg" = "-" ]; then envs=""; else envs="$cfg"; fi
    env $envs python bench.py --no-cpu-baseline --no-ba --no-files --no-extras --steps ${STEPS:-40} > $O/run_$i.json 2> $O/run_$i.err
    python - "$O/run_$i.json" "$cfg" <<'PY' | tee -a $O/ab.txt
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d["roofline"]["kernel_ms_per_step"]
print("%-40s value %9.1f unprofiled %9.1f  " % (sys.argv[2], d["value"], d["value_unprofiled"]) + " ".join("%s=%.4f" % (a.replace("_kernel", ""), b) for a, b in k.items()))
PY
  done
done
