#!/bin/bash
# round 4, GPU call 3: decode phase breakdown (plain vs column split at 5 120 / 4 096 / 1 536 rows), sparse_xe 98.8 % trace
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O
python3 scratch/split_phases.py 1024 5 stack,stack_split > $O/phases_1024.txt 2>&1
python3 scratch/split_phases.py 819 5 stack,stack_split > $O/phases_819.txt 2>&1
python3 scratch/split_phases.py 307 5 stack_split > $O/phases_307.txt 2>&1
B="python3 bench.py --no-extra-workloads --no-cpu-baseline"
stats() {   # name, extra args
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw_$1 -o t -- $B --steps 3 --warmup 1 $2 > $O/$1.log 2>&1
  cp $O/raw_$1/t_kernel_stats.csv $O/r04_$1_kernel_stats.csv
}
stats sparse_xe_988_kernels "--workload sparse_xe --variant 988_kernels"
stats sparse_xe_988 "--workload sparse_xe --variant 988"
rm -rf $O/raw_*
cat $O/phases_*.txt | grep -v amdgpu
python3 scratch/prof_summary.py $O/r04_sparse_xe_988_kernels_kernel_stats.csv 6 | head -24
python3 scratch/prof_summary.py $O/r04_sparse_xe_988_kernel_stats.csv 6 | head -12
