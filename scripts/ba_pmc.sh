#!/bin/bash
# MFMA counter passes of the BA solver (GPU box, via gpurun): usage scripts/ba_pmc.sh <tag>
#   -> gpurun_out/<tag>/pmc_mfma_ba_k50.txt      one window of configs[4] size (50 keyframes / 8000 points), scripts/ba_profile.py 50 8000
#   -> gpurun_out/<tag>/pmc_mfma_ba_batched.txt  32 windows of configs[2] size through orbx_ba_solve_visual_batch, scripts/ba_batch_profile.py
# The profiled program stands directly behind `--` (no shell, no env hop).
set -e
TAG=${1:-rXX}
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CTRS="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_WAVES"
rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $O/k50 -- python3 $R/scripts/ba_profile.py 50 8000 visual-only > $O/k50_run.txt 2> $O/k50.err
# (the profiler serialises the two halves a large batch normally runs at the same time on two streams, so that each would have half the chip to itself:
# the counters are taken on the batch as ONE launch per kernel)
ORBX_BA_NO_SPLIT=1 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $O/bat -- python3 $R/scripts/ba_batch_profile.py 32 20 2000 > $O/bat_run.txt 2> $O/bat.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/k50s -- python3 $R/scripts/ba_profile.py 50 8000 visual-only > $O/k50_stats_run.txt 2> $O/k50s.err
cd $R
{ echo "# rocprofv3 --kernel-trace --pmc $CTRS -- python3 scripts/ba_profile.py 50 8000 visual-only"; cat $O/k50_run.txt; python scripts/pmc_mfma_summary.py $(ls $O/k50/*/*counter_collection.csv | head -1); } > $O/pmc_mfma_ba_k50.txt
{ echo "# ORBX_BA_NO_SPLIT=1 rocprofv3 --kernel-trace --pmc $CTRS -- python3 scripts/ba_batch_profile.py 32 20 2000"; cat $O/bat_run.txt; python scripts/pmc_mfma_summary.py $(ls $O/bat/*/*counter_collection.csv | head -1); } > $O/pmc_mfma_ba_batched.txt
cp $(ls $O/k50s/*/*kernel_stats.csv | head -1) $O/ba_k50_kernel_stats.csv
rm -rf $O/k50 $O/bat $O/k50s
cat $O/pmc_mfma_ba_k50.txt
