#!/bin/bash
mkdir -p gpurun_out/r04d
for n in 0 2 3 4; do
timeout 400 python bench.py --workload decode --decode-streams $n --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r04d/ds$n.json 2> gpurun_out/r04d/ds$n.err
python - <<PY
import json
try:
    r=json.loads(open("gpurun_out/r04d/ds$n.json").read().strip().splitlines()[-1]); print("decode streams $n ms", r["ms_per_step"])
except Exception as e: print("streams $n failed", e, open("gpurun_out/r04d/ds$n.err").read()[-500:])
PY
done
