#!/bin/bash
# SQ counter passes over the bench workload (GPU box, via gpurun): usage scripts/pmc_sq.sh <tag> "<CTR CTR ...>" ["<CTR ...>" ...]
# One rocprofv3 run per counter group (kernel trace only beside --pmc); summary by scripts/pmc_sq_summary.py.
set -e
TAG=$1; shift
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/g$i -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ba --no-files --no-extras ${ORBX_BENCH_INPUT:---small-gen} > /dev/null 2> $O/g$i.err
done
cd $R
python scripts/pmc_sq_summary.py $(ls $O/g*/*/*counter_collection.csv) > $O/summary.txt
cat $O/summary.txt
