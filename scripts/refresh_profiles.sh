#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: bench line, rocprofv3 kernel stats and the two PMC passes of the same
# workload, all written under gpurun_out/<tag>/ for copying into profiles/.   usage: scripts/refresh_profiles.sh <tag>
set -e
TAG=${1:-rXX}
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
python bench.py > $O/bench_b256.json 2> $O/bench_b256.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-files --no-extras > $O/bench_under_rocprof.json 2> $O/rocprof_stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ba --no-files --no-extras --small-gen > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ba --no-files --no-extras --small-gen > /dev/null 2> $O/pmc_write.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $O/pmc_sq -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ba --no-files --no-extras --small-gen > /dev/null 2> $O/pmc_sq.err
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc_sq2 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ba --no-files --no-extras --small-gen > /dev/null 2> $O/pmc_sq2.err
cd $R
python scripts/pmc_summary.py $(ls $O/pmc_fetch/*/*counter_collection.csv | head -1) $(ls $O/pmc_write/*/*counter_collection.csv | head -1) 256 2000 $O/pmc_traffic.json $(ls $O/pmc_sq/*/*counter_collection.csv | head -1) $(ls $O/pmc_sq2/*/*counter_collection.csv | head -1) > $O/pmc_hbm_traffic_b256.txt
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/bench_b256_kernel_stats.csv
rm -rf $O/stats $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_sq2   # raw traces: tens of MiB each, gpurun_out/ merges back at most 64 MiB
tail -1 $O/bench_b256.json | cut -c1-600
