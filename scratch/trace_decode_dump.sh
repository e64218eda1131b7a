#!/bin/bash
# kernel trace of a few beam-5 decodes -> per-queue / per-kernel time of one decode
TAG=${1:-trace_decode}; shift || true
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python3 $R/bench.py --workload decode --steps 4 --warmup 2 --no-cpu-baseline --no-extra-workloads "$@" > $O/trace.log 2>&1
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 - <<PY
import csv, collections
rows=list(csv.DictReader(open("$f")))
ev=sorted((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Queue_Id"],r["Kernel_Name"]) for r in rows)
# a decode ends with beam_finalize
ends=[i for i,e in enumerate(ev) if "beam_finalize" in e[3]]
a,b=ends[2]+1, ends[3]+1
g=ev[a:b]; t0=g[0][0]
print("span ms", (max(e[1] for e in g)-t0)/1e6, "kernels", len(g))
for q in sorted({e[2] for e in g}):
    k=[e for e in g if e[2]==q]
    print("queue",q,"n",len(k),"busy ms %.3f"%(sum(e[1]-e[0] for e in k)/1e6),"first %.3f last %.3f"%((k[0][0]-t0)/1e6,(k[-1][1]-t0)/1e6))
    by=collections.defaultdict(lambda:[0,0])
    for s,e,_,n in k: by[n.replace("(anonymous namespace)::","")[:58]][0]+=1; by[n.replace("(anonymous namespace)::","")[:58]][1]+=e-s
    for n,(c,t) in sorted(by.items(), key=lambda x:-x[1][1])[:14]: print("    %-58s %4d %.3f ms avg %.1f us"%(n,c,t/1e6,t/c/1e3))
st=[e for e in g if "decoder_stack_kernel" in e[3]]
print("first stack launch at %.3f ms; per position: "%((st[0][0]-t0)/1e6), [round((st[i+1][0]-st[i][0])/1e3) for i in range(len(st)-1)])
last=g[0][1]; gaps=0
for s,e,q,n in g[1:]:
    if s>last: gaps+=s-last
    last=max(last,e)
print("idle (no kernel) ms %.3f"%(gaps/1e6))
PY
rm -rf $O/trace
