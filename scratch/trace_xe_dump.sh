#!/bin/bash
# kernel trace of a few XE steps -> gpurun_out/<tag>/{timeline.txt, dump.txt}
TAG=${1:-trace}; shift || true
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra-workloads "$@" > $O/trace.log 2>&1
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python3 $R/scratch/timeline_xe.py $f 4 > $O/timeline.txt 2>&1 && python3 $R/scratch/timeline_dump.py $f 4 > $O/dump.txt 2>&1
rm -rf $O/trace
head -8 $O/timeline.txt
