#!/bin/bash
# round 4, GPU call 2: new dist tests, sparse 98.8 % lines, SCST kernel traces
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O
python -m pytest tests/test_gpu_dist.py -x -q 2>&1 | tail -15 > $O/pytest_dist.log
B="python3 bench.py --no-extra-workloads --no-cpu-baseline"
for v in "" kernels 988 988_kernels; do
  $B --workload sparse_xe --variant "$v" --steps 20 --warmup 3 > $O/sparse_xe_$v.json 2>> $O/bench.err
done
for v in "" dense_kernels 988 988_dense_kernels; do
  $B --workload sparse_decode --variant "$v" --steps 10 --warmup 3 > $O/sparse_decode_$v.json 2>> $O/bench.err
done
stats() {   # name, extra args
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw_$1 -o t -- $B --steps 3 --warmup 1 $2 > $O/$1.log 2>&1
  cp $O/raw_$1/t_kernel_stats.csv $O/r04_$1_kernel_stats.csv
}
stats scst_b256_ns5 "--workload scst"
stats scst_nodrop_b256_ns5 "--workload scst --variant nodrop"
rm -rf $O/raw_*
tail -5 $O/pytest_dist.log
for f in $O/sparse_*.json; do echo $f; python3 -c "
import json,sys
d=json.load(open('$f')); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline'].get('dominant_kernel',{}).get('avg_us'))"; done
tail -3 $O/bench.err
python3 scratch/prof_summary.py $O/r04_scst_b256_ns5_kernel_stats.csv 6 | head -30
python3 scratch/prof_summary.py $O/r04_scst_nodrop_b256_ns5_kernel_stats.csv 6 | head -24
