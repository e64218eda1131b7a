#!/bin/bash
# Instruction-cache counters per kernel of a bench.py workload (after round 6's finding that the general GEMM epilogue overflowed the cache):
#     gpurun -- "bash scratch/icache_pmc.sh"        -> gpurun_out/r06/r06_icache_<workload>.csv (requests / hits / misses per launch)
# Separate --pmc run with --kernel-trace only.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06; mkdir -p $O
B="python3 bench.py --no-extra-workloads --no-cpu-baseline --steps 2 --warmup 1"
for w in xe scst decode; do
  extra=""; [ $w != xe ] && extra="--workload $w"
  rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES --output-format csv -d $O/raw_ic_$w -o t -- $B $extra > $O/icache_$w.log 2>&1
  python3 scratch/pmc_summary.py $O/raw_ic_$w/t_counter_collection.csv > $O/r06_icache_$w.csv 2>> $O/icache_$w.log
  rm -rf $O/raw_ic_$w
done
tail -3 $O/icache_xe.log; head -30 $O/r06_icache_xe.csv
