#!/bin/bash
# kernel trace of a few SCST steps -> gpurun_out/<tag>/{stats per step}
TAG=${1:-trace_scst}; shift || true
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 $R/bench.py --workload scst --steps 5 --warmup 2 --no-cpu-baseline --no-extra-workloads "$@" > $O/trace.log 2>&1
cp $O/trace/t_kernel_stats.csv $O/kernel_stats.csv
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 - <<PY
import csv, collections
rows=list(csv.DictReader(open("$f")))
ev=sorted((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Queue_Id"],r["Kernel_Name"]) for r in rows)
ends=[i for i,e in enumerate(ev) if "adam_clip" in e[3]]
# steps end with the last adam of a step (2 per step)
a,b=ends[2*4-1]+1, ends[2*5-1]+1
g=ev[a:b]; t0=g[0][0]
print("span ms", (max(e[1] for e in g)-t0)/1e6, "kernels", len(g))
qs=sorted({e[2] for e in g})
for q in qs:
    k=[e for e in g if e[2]==q]
    print("queue",q,"n",len(k),"busy ms %.3f"%(sum(e[1]-e[0] for e in k)/1e6),"first %.3f last %.3f"%((k[0][0]-t0)/1e6,(k[-1][1]-t0)/1e6))
    by=collections.defaultdict(lambda:[0,0])
    for s,e,_,n in k: by[n.replace("(anonymous namespace)::","")[:58]][0]+=1; by[n.replace("(anonymous namespace)::","")[:58]][1]+=e-s
    for n,(c,t) in sorted(by.items(), key=lambda x:-x[1][1])[:10]: print("    %-58s %4d %.3f ms avg %.1f us"%(n,c,t/1e6,t/c/1e3))
# phase boundaries: first rollout kernel, last rollout kernel
ro=[e for e in g if "decoder_stack_tp" in e[3]]
if ro: print("rollout: first %.3f last end %.3f (%d launches)"%((ro[0][0]-t0)/1e6,(ro[-1][1]-t0)/1e6,len(ro)))
PY
rm -rf $O/trace
