#!/bin/bash
# round 4, GPU call 4: full suite, the default bench line, sparse_xe / scst after the plan-kernel and 32-bit-hash changes
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O
python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > $O/pytest_full2.log
python3 bench.py --steps 20 --warmup 5 > $O/default_line.json 2> $O/default.err
tail -3 $O/pytest_full2.log; wc -c $O/default_line.json; tail -3 $O/default.err
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r4/default_line.json"))
print("xe", d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("whole_step"), d["cpu_baseline"])
for k,v in d["workloads"].items(): print(k, v["ms_per_step"], v["frac"], v.get("kernel",{}).get("avg_us"), v.get("cpu_baseline"))
PY
