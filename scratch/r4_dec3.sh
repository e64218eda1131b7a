#!/bin/bash
mkdir -p gpurun_out/r04i
for i in 1 2 3; do
for w in decode sparse_decode; do
timeout 400 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r04i/w.json 2> gpurun_out/r04i/w.err
python - <<PY
import json
r=json.loads(open("gpurun_out/r04i/w.json").read().strip().splitlines()[-1]); print("$w ms", r["ms_per_step"], "stack", r["roofline"].get("dominant_kernel",{}).get("avg_us"), "chain", r["roofline"].get("chain_kernel",{}).get("avg_us"))
PY
done
done
