#!/bin/bash
# kernel-trace stats of the XE step (quick look): bash scratch/prof_xe_quick.sh <tag> [bench args]
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
T=$1; shift
O=gpurun_out/prof_$T; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw -o t -- python3 bench.py --no-extra-workloads --no-cpu-baseline --steps 5 --warmup 2 "$@" > $O/log.txt 2>&1
cp $O/raw/t_kernel_stats.csv $O/kernel_stats.csv; rm -rf $O/raw
tail -1 $O/log.txt | cut -c1-400
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/kernel_stats.csv")))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms per step", tot/7/1e6)
for r in rows[:32]:
    print(r['Name'][:100].ljust(100), r['Calls'].rjust(5), f"{float(r['TotalDurationNs'])/7/1e6:8.3f} ms/step", f"{float(r['AverageNs'])/1e3:8.1f} us", r['Percentage'])
PY
