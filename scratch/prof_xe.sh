#!/bin/bash
# Kernel trace + PMC passes of the default bench command (XE, B=256, bf16); summaries -> gpurun_out/<tag>/
# usage: bash scratch/prof_xe.sh <tag> [workload args...]
set -u
TAG=${1:-prof}; shift || true
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $O/trace.log 2>&1
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats.csv && python3 $R/scratch/prof_summary.py $O/kernel_stats.csv 5 > $O/kernel_summary.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $O/pmc_$c.log 2>&1
  f=$(find $O/pmc_$c -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python3 $R/scratch/pmc_summary.py $f > $O/pmc_$(echo $c | tr A-Z a-z).csv
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_sq -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $O/pmc_sq.log 2>&1
f=$(find $O/pmc_sq -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python3 $R/scratch/pmc_summary.py $f > $O/pmc_sq.csv
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_l2 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $O/pmc_l2.log 2>&1
f=$(find $O/pmc_l2 -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python3 $R/scratch/pmc_summary.py $f > $O/pmc_l2.csv
rm -rf $O/trace $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_sq $O/pmc_l2
ls -la $O
