#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: bench line, rocprofv3 kernel stats and the PMC passes of the same
# workload, all written under gpurun_out/<tag>/ for copying into profiles/.   usage: scripts/refresh_profiles.sh <tag>
# The counter passes run on the bench's OWN input: the default run saves its 3 x BATCH unique generated pairs to a file and the profiled
# processes load that file (bench.py --load-batches), so torch's scene generator (about 50 000 small launches, under which
# rocprofv3's counter collection crashed in round 2) is not in the profiled process.
set -e
TAG=${1:-rXX}
BATCH=${BATCH:-512}      # bench.py's default pairs per step
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
B=/tmp/orbx_bench_batches.npy
python bench.py --batch $BATCH --save-batches $B > $O/bench_b$BATCH.json 2> $O/bench_b$BATCH.err
cd /tmp && export TMPDIR=/tmp
COMMON="--batch $BATCH --no-cpu-baseline --no-files --no-extras --load-batches $B"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 10 --warmup 2 $COMMON > $O/bench_under_rocprof.json 2> $O/rocprof_stats.err
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "sq:SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "sq2:SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $O/pmc_$name -- python3 $R/bench.py --steps 3 --warmup 1 --no-ba $COMMON > /dev/null 2> $O/pmc_$name.err
done
cd $R
export ORBX_PMC_UNIQUE_PAIRS=$((3 * BATCH)) ORBX_PMC_INPUT="the bench's own 3 x $BATCH generated pairs (bench.py --save-batches / --load-batches)"
python scripts/pmc_summary.py $(ls $O/pmc_fetch/*/*counter_collection.csv | head -1) $(ls $O/pmc_write/*/*counter_collection.csv | head -1) $BATCH 2000 $O/pmc_traffic.json $(ls $O/pmc_sq/*/*counter_collection.csv | head -1) $(ls $O/pmc_sq2/*/*counter_collection.csv | head -1) > $O/pmc_hbm_traffic_b$BATCH.txt
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/bench_b${BATCH}_kernel_stats.csv
rm -rf $O/stats $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_sq2   # raw traces: tens of MiB each, gpurun_out/ merges back at most 64 MiB
rm -f $B
tail -1 $O/bench_b$BATCH.json | cut -c1-600
