#!/bin/bash
# Round-4 refresh of the kernel-trace summaries whose kernel mix changed late in the round (sparse training with the chains kept,
# sparse decode with the stream built on the side stream, the train-mode rollout kernel); run from the repo root through gpurun.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
B="python3 bench.py --no-extra-workloads --no-cpu-baseline --steps 3 --warmup 1"
stats() {   # name, extra args
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw_$1 -o t -- $B $2 > $O/$1.log 2>&1
  echo "$1 rc=$?"
  cp $O/raw_$1/t_kernel_stats.csv $O/r04_$1_kernel_stats.csv
}
date +%T
stats sparse_xe_988_kernels "--workload sparse_xe --variant 988_kernels"
date +%T
stats sparse_decode_stack_b1024_beam5 "--workload sparse_decode"
date +%T
stats scst_b256_ns5 "--workload scst"
date +%T
stats xe_b256_bf16 ""
date +%T
stats decode_stack_b1024_beam5 "--workload decode"
date +%T
rm -rf $O/raw_*
ls -la $O | head -30
