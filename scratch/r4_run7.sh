#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O
B="python3 bench.py --no-extra-workloads --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw_xe -o t -- $B --steps 3 --warmup 1 > $O/xe_chain.log 2>&1
cp $O/raw_xe/t_kernel_stats.csv $O/r04_xe_bchain_kernel_stats.csv
rm -rf $O/raw_xe
python3 scratch/prof_summary.py $O/r04_xe_bchain_kernel_stats.csv 6 | head -24
