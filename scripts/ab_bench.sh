#!/bin/bash
# Same-box A/B of two builds of the library (boxes differ by a few per cent, so compare only within one call).
# usage: scripts/ab_bench.sh <tag> <variant.so> [more.so ...]   -> gpurun_out/<tag>/<name>.json, alternating twice
set -e
TAG=$1; shift
O=gpurun_out/$TAG
mkdir -p $O
for rep in 1 2; do
  for so in "$@"; do
    n=$(basename $so .so)
    ORBX_LIBRARY=$PWD/$so python bench.py --no-cpu-baseline --no-ba --no-files --no-extras --steps 20 > $O/${n}_$rep.json 2> $O/${n}_$rep.err
    python - "$O/${n}_$rep.json" "$n" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d["roofline"]["kernel_ms_per_step"]
print(sys.argv[2], d["value"], d["value_unprofiled"], " ".join("%s=%.4f" % (a.replace("_kernel", ""), b) for a, b in k.items()))
PY
  done
done
